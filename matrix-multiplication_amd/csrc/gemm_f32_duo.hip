// Dense fp32 product, persistent "two halves in anti-phase" form — what BERT-base attention's six products
// (reference call sites: README.md:62-78 cublasTransbMM.apply(q, k); matmuls.py:81-176 backward products;
// kernels behind custom_mm.cublas_bmm, src/custom_mm.cpp:125-164 / src/baseline_mm.cu:105-155) run on when
// the shape is made of whole tiles.
//
// Why another kernel.  gemm_f32.hip's tiles keep 2–3 independent 4-wave workgroups on a CU and rely on them
// drifting apart so that one workgroup's staging / barrier / epilogue time hides behind another's MFMAs.  They
// do not drift apart: every workgroup runs the same phases, the matrix pipe is shared fairly, and the SQ counters
// show the pipe idle a third of the time (profiles/r02_gemm_sq_counters.json: 66 % / 72 % busy).  Here the
// anti-phase is built in:
//
//   * one 512-thread workgroup per CU, persistent over a contiguous range of output tiles; its two halves
//     (waves 0–3, waves 4–7: one wave of each half on every SIMD) own separate LDS operand buffers and separate
//     output tiles, and run the same program ONE BARRIER APART (half 1 executes one extra barrier first):
//
//         half 0:  stage(0) | MFMA(0)  | stage(1) | MFMA(1)  | …
//         half 1:           | stage(0) | MFMA(0)  | stage(1) | MFMA(1) …
//
//     so in every interval between two workgroup barriers exactly one wave per SIMD feeds the matrix pipe while
//     its partner does everything else: the previous tile's epilogue (global stores straight from the
//     accumulators), registers → LDS for the next k-tile, and the global loads for the one after.  An interval is
//     64 × TN × 2 MFMAs per wave = 4096 × TN × 2 cycles of matrix pipe; everything else fits in it several times over.
//   * k-tile of 64.  An operand that is contiguous along k goes to LDS split into even and odd k
//     ([row][32 even | 32 odd], row stride 68 floats): the lane (row, k parity) of v_mfma_f32_32x32x2_f32 then
//     reads FOUR consecutive k-steps with one conflict-free ds_read_b128 (the [row][k+1] image of gemm_f32.hip
//     needs one ds_read_b32 per step), written with two ds_write_b64 per float4.  An operand contiguous along
//     m / n is staged as it lies ([k][ext], ds_write_b128) and read with ds_read_b32.
//   * the accumulators hold C tiles with lane ↔ column: register r of a 32×32 block is row
//     (r&3) + 8·(r>>2) + 4·(lane>>5), so the epilogue is 16 plain global_store_dword per block, each writing two
//     whole 128-byte row segments — no LDS patch, no transposition, no waits in the epilogue.
//   * global loads for step s+1 are issued right after the registers of step s went to LDS, i.e. a whole
//     stage + MFMA period (≥ 8192 cycles ≈ 3.6 µs) before they are needed, with one register set.
//
// Per output element the products are accumulated over k in increasing order by the MFMA's fmaf chain, exactly
// as in every other kernel here: same bits as oracle_gemm_f32 and as the tiles of gemm_f32.hip
// (tests/test_gpu_parity.py::test_gemm_duo_*).
#include <type_traits>

#include "mi_common.h"

#ifndef MI_DUO_ABL
#define MI_DUO_ABL 0  // developer probes only: 1 no C stores, 2 no MFMAs, 4 no operand loads
#endif
#ifndef MI_DUO_STAGE_PRIO
#define MI_DUO_STAGE_PRIO 2  // wave priority during a stage interval (the multiplying partner runs at 0)
#endif
#ifndef MI_DUO_DYNAMIC_ROLES
#define MI_DUO_DYNAMIC_ROLES 1  // 0: half = wave index / 4 (developer probes)
#endif

#ifdef MI_DUO_TIMING  // developer probes only: cycle stamps of the first events of every wave, kept in LDS (no
                       // vector-memory operations of their own) and copied out at the end
__device__ unsigned long long g_duo_stamps[256][8][64];
__device__ unsigned long long g_duo_real[256][2];
#define DUO_STAMP() do { if (lane == 0 && stamp_i_ < 64) stamp_lds_[(tid >> 6) * 64 + stamp_i_] = __builtin_amdgcn_s_memtime(); if (tid == 0 && stamp_i_ == 43) g_duo_real[blockIdx.x & 255][1] = __builtin_amdgcn_s_memrealtime(); ++stamp_i_; } while (0)
#else
#define DUO_STAMP() do {} while (0)
#endif

namespace {

using mi::f32x4;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int DBK = 64;    // k-tile
constexpr int DBM = 128;   // tile rows
constexpr int KLD = 68;    // LDS row stride (floats) of a k-contiguous operand image: 17 float4 → b128 reads of 16
                           // rows at one k-quad hit 16 different bank quads

// One operand's k-tile (EXT rows or columns × 64 k) of one half (256 threads).
// KC: element (e, k) lives at src[e*ld + k] (A not transposed / B transposed); else at src[k*ld + e].
//
// Everything a stage interval executes is written to cost NO vector-ALU instruction: on gfx950 the fp32 MFMA runs at
// the vector rate, and a partner wave's VALU instruction waits for a gap in the multiplying wave's MFMA stream
// (measured: ≈50 cycles each; a first version with 64 v_mov + 32 address adds + 64 accumulator clears per stage
// took 14 k cycles per stage against 8.4 k of MFMAs).  So: loads are `uniform base + one loop-invariant 32-bit
// lane offset` (the uniform part advances in scalar registers), LDS stores are one loop-invariant lane address +
// immediate offsets, written dword by dword (no register shuffles), and a tile's accumulators start from the MFMA's
// zero C operand instead of being cleared.
template <int EXT, bool KC>
struct DuoOperand {
  static constexpr int VECS = EXT * DBK / 4 / 256;
  static constexpr int FLOATS = KC ? EXT * KLD : DBK * EXT;
  static constexpr int ROWV = KC ? 16 : EXT / 4;   // float4 per source row
  static constexpr int RPV = 256 / ROWV;           // source rows covered by one float4 per thread

  // byte offset of this thread's first float4 inside a tile (loop-invariant)
  static __device__ __forceinline__ unsigned lane_offset(long ld, int th) {
    return (unsigned)((th / ROWV) * (int)ld + (th % ROWV) * 4) * 4u;
  }
  // this thread's first LDS element (loop-invariant)
  static __device__ __forceinline__ float* lane_lds(float* lds, int th) {
    return KC ? lds + (th / 16) * KLD + 2 * (th % 16) : lds + (th / ROWV) * EXT + (th % ROWV) * 4;
  }

  // tile (uniform pointer) → registers: buffer loads, whose address is a scalar descriptor + scalar row step + the
  // lane's 32-bit offset — no vector address arithmetic (a flat global_load costs one v_lshl_add_u64 each here)
  static __device__ __forceinline__ void load(f32x4 (&r)[VECS], const float* tile, unsigned row_step_bytes, unsigned off) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int v = 0; v < VECS; ++v) {
      if (MI_DUO_ABL & 4) {
        r[v] = f32x4{1.f, 2.f, 3.f, (float)off};
      } else {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, (int)(v * row_step_bytes), 0);
        r[v] = __builtin_bit_cast(f32x4, x);
      }
    }
  }

  // registers → LDS, dword by dword for the even / odd split (volatile: hipcc otherwise pairs {x, z} into a
  // ds_write_b64 and pays two v_mov per pair to make the registers adjacent)
  static __device__ __forceinline__ void store(const f32x4 (&r)[VECS], float* p0) {
#pragma unroll
    for (int v = 0; v < VECS; ++v) {
      if (KC) {
        typedef __attribute__((address_space(3))) float lds_float;  // keeps the volatile stores ds_write (not flat)
        volatile lds_float* p = (volatile lds_float*)(p0 + v * RPV * KLD);
        p[0] = r[v].x;   // k = 4s   → even position 2s
        p[1] = r[v].z;   // k = 4s+2 → even position 2s+1
        p[32] = r[v].y;  // k = 4s+1 → odd position 2s
        p[33] = r[v].w;  // k = 4s+3
      } else {
        *reinterpret_cast<f32x4*>(p0 + v * RPV * EXT) = r[v];
      }
    }
  }

  // MFMA operand values of lane (e, k parity) for the k-steps 4g … 4g+3 of the tile (k = 2·step + parity)
  static __device__ __forceinline__ f32x4 read(const float* lds, int e, int parity, int g) {
    if (KC) return *reinterpret_cast<const f32x4*>(lds + e * KLD + parity * 32 + 4 * g);
    const float* p = lds + (8 * g + parity) * EXT + e;
    return f32x4{p[0], p[2 * EXT], p[4 * EXT], p[6 * EXT]};
  }
};

template <int BN, bool AK, bool BKC, bool HAS_BIAS>
__global__ __launch_bounds__(512) void gemm_f32_duo_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int nk, long lda, long ldb, long ldc,
                                                           long strideA, long strideB, long strideC, int tiles_n,
                                                           int tiles_per_item, int total_tiles,
                                                           const float* __restrict__ bias) {
  typedef DuoOperand<DBM, AK> LA;
  typedef DuoOperand<BN, BKC> LB;
  constexpr int TM = 2, TN = BN / 64;
  extern __shared__ __attribute__((aligned(16))) float duo_lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  // Roles.  The anti-phase only pays if every SIMD holds ONE wave of each half, and the hardware does not promise
  // where wave w of a workgroup lands (observed on gfx950: consecutive waves fill a SIMD pair-wise).  So a wave's
  // role is taken from where it actually runs: role wave = its SIMD id (HW_REG_HW_ID bits 5:4), half = its arrival
  // rank among the workgroup's waves on that SIMD (an LDS counter per SIMD).  If a placement other than two waves
  // per SIMD ever shows up, every wave falls back to the static roles (correct either way; speed only).
  int half = tid >> 8, wave = (tid >> 6) & 3;
#if MI_DUO_DYNAMIC_ROLES
  {
    int* cnt = reinterpret_cast<int*>(duo_lds);
    const int simd = (int)((__builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4)) & 3);  // HW_ID[5:4]
    if (tid < 4) cnt[tid] = 0;
    __syncthreads();
    int rank = 0;
    if (lane == 0) rank = atomicAdd(&cnt[simd], 1);
    __syncthreads();
    const bool two_each = cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2;
    __syncthreads();  // the counters sit where the operand buffers are about to go
    if (two_each) half = __builtin_amdgcn_readfirstlane(rank), wave = simd;
  }
#endif
  half = __builtin_amdgcn_readfirstlane(half);
  wave = __builtin_amdgcn_readfirstlane(wave);
  const int th = wave * 64 + lane;  // thread id inside the half
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  float* As = duo_lds + half * (LA::FLOATS + LB::FLOATS);
  float* Bs = As + LA::FLOATS;

  // this workgroup's contiguous tile range; ranges are dealt XCD-contiguously (workgroups b, b+8, … share an XCD's L2)
  const unsigned G = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = G / 8, rem = G % 8, xcd = bid % 8, pos = bid / 8;
  const unsigned w = xcd * q8 + (xcd < rem ? xcd : rem) + pos;
  const unsigned per = (unsigned)total_tiles / G, extra = (unsigned)total_tiles % G;  // the first `extra` ranges own one more
  const int t_begin = (int)(w * per + (w < extra ? w : extra)), t_end = t_begin + (int)per + (w < extra ? 1 : 0);
  const int steps = (t_end - t_begin + 1) / 2;  // tiles per half (the second half may own one fewer)

  f32x16 acc[TM][TN];
  f32x4 ra[LA::VECS], rb[LB::VECS];
  const unsigned a_off = LA::lane_offset(lda, th), b_off = LB::lane_offset(ldb, th);
  float* const a_lds = LA::lane_lds(As, th);
  float* const b_lds = LB::lane_lds(Bs, th);
  const unsigned a_step = (unsigned)(LA::RPV * (int)lda) * 4u, b_step = (unsigned)(LB::RPV * (int)ldb) * 4u;  // uniform
  const unsigned c_off = (unsigned)(4 * lhi * (int)ldc + l31) * 4u;  // this lane inside a 32×32 block of C (bytes)
  const unsigned c_row = (unsigned)ldc * 4u;

  // tile ids are wave-uniform 32-bit values (the launcher checks the count): scalar divisions, no 64-bit ones
  auto issue_loads = [&](int t, int kt) {
    const unsigned item = (unsigned)t / (unsigned)tiles_per_item;
    const unsigned rt = (unsigned)t % (unsigned)tiles_per_item, tm = rt / (unsigned)tiles_n, tn = rt % (unsigned)tiles_n;
    const float* a = A + (long)item * strideA + (AK ? (long)tm * DBM * lda + (long)kt * DBK : (long)kt * DBK * lda + (long)tm * DBM);
    const float* b = B + (long)item * strideB + (BKC ? (long)tn * BN * ldb + (long)kt * DBK : (long)kt * DBK * ldb + (long)tn * BN);
    LA::load(ra, a, a_step, a_off);
    LB::load(rb, b, b_step, b_off);
  };

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto mfma_phase = [&](auto first_) {
    constexpr bool FIRST = decltype(first_)::value;  // a tile's first k-tile starts its chains from the zero C operand
    f32x4 a[2][TM], b[2][TN];
    auto read_group = [&](int g, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[slot][i] = LA::read(As, wm * 64 + i * 32 + l31, lhi, g);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[slot][j] = LB::read(Bs, wn * (BN / 2) + j * 32 + l31, lhi, g);
    };
    read_group(0, 0);
#pragma unroll
    for (int g = 0; g < DBK / 8; ++g) {
      if (g + 1 < DBK / 8) read_group(g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);  // operand reads one group (16 MFMAs) ahead; left alone hipcc sinks them to their use
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (MI_DUO_ABL & 2) {
              if (FIRST && g == 0 && c == 0) acc[i][j] = zero16;
            } else if (FIRST && g == 0 && c == 0) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][c], b[g & 1][j][c], zero16, 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][c], b[g & 1][j][c], acc[i][j], 0, 0, 0);
            }
          }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // 16 buffer_store_dword per 32×32 block, each two whole 128-byte row segments: scalar descriptor of the wave's
  // 64×(BN/2) patch + scalar row offset + the lane's loop-invariant offset — no vector address arithmetic
  auto epilogue = [&](int t) {
    const unsigned item = (unsigned)t / (unsigned)tiles_per_item;
    const unsigned rt = (unsigned)t % (unsigned)tiles_per_item, tm = rt / (unsigned)tiles_n, tn = rt % (unsigned)tiles_n;
    float* c0 = C + (long)item * strideC + ((long)tm * DBM + wm * 64) * ldc + (long)tn * BN + wn * (BN / 2);  // uniform
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(c0 + (long)(i * 32) * ldc, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float bv = 0.f;
        if (HAS_BIAS) bv = bias[(long)tn * BN + wn * (BN / 2) + j * 32 + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r];
          if (HAS_BIAS) v += bv;  // after the chain: one extra rounding, as `output += bias` (and never x + 0: keeps −0)
          const unsigned soff = (unsigned)((r & 3) + 8 * (r >> 2)) * c_row;  // 16 scalar row offsets, shared by every block
          if (MI_DUO_ABL & 1) {
            if (v == 12345.678f) c0[(long)(i * 32) * ldc + soff / 4 + c_off / 4 + j * 32] = v;  // keeps the accumulators live
          } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (int)(c_off + j * 128), (int)soff, 2 /* nt */);
          }
        }
      }
    }
  };

#ifdef MI_DUO_TIMING
  int stamp_i_ = 0;
  unsigned long long* stamp_lds_ = reinterpret_cast<unsigned long long*>(duo_lds + 2 * (LA::FLOATS + LB::FLOATS));
  if (tid == 0) g_duo_real[blockIdx.x & 255][0] = __builtin_amdgcn_s_memrealtime();
#endif
  DUO_STAMP();
  if (t_begin + half < t_end) issue_loads(t_begin + half, 0);
  if (half == 1) __syncthreads();  // the stagger: half 1 runs one barrier behind half 0
  for (int ti = 0; ti < steps; ++ti) {
    const int t = t_begin + 2 * ti + half;
    const bool valid = t < t_end;
    for (int kt = 0; kt < nk; ++kt) {
      // ---- stage interval (the other half multiplies): registers → my LDS buffers, the next step's loads, and LAST
      // the previous tile's stores: vmcnt retires in order, so with the stores in front the wait for the operand
      // registers would also wait for a store issued a moment ago (a full write round trip: measured 11 k cycles
      // per stage); behind the loads they have a whole period to drain before anything waits on them
      __builtin_amdgcn_s_setprio(MI_DUO_STAGE_PRIO);
      if (valid) {
        LA::store(ra, a_lds);
        LB::store(rb, b_lds);
        if (kt + 1 < nk) issue_loads(t, kt + 1);
        else if (t + 2 < t_end) issue_loads(t + 2, 0);
      }
      if (kt == 0 && ti > 0) epilogue(t - 2);  // (t − 2 is always a tile of this range)
      DUO_STAMP();  // staged, loads issued
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      DUO_STAMP();  // MFMA interval begins
      // ---- MFMA interval (the other half stages)
      if (valid) {
        if (kt == 0) mfma_phase(std::true_type{});
        else mfma_phase(std::false_type{});
      }
      DUO_STAMP();  // MFMAs issued
      if (!(half == 1 && ti == steps - 1 && kt == nk - 1)) __syncthreads();  // half 1 entered one barrier late
      DUO_STAMP();  // stage interval begins
    }
  }
  {
    const int t_last = t_begin + 2 * (steps - 1) + half;  // the last tile's stores have nothing left to hide behind
    if (t_last < t_end) epilogue(t_last);  // (a half that owns one tile fewer stored its last one in the empty step's stage)
  }
#ifdef MI_DUO_TIMING
  if (lane == 0) for (int e = 0; e < 64; ++e) g_duo_stamps[blockIdx.x & 255][tid >> 6][e] = stamp_lds_[(tid >> 6) * 64 + e];
#endif

}

template <int BN, bool AK, bool BKC, bool HAS_BIAS>
int launch_duo_b(const float* A, const float* B, float* C, int m, int n, int k, long lda, long ldb, long ldc, long sA,
               long sB, long sC, int batch, const float* bias, hipStream_t s, int cu_count) {
  typedef DuoOperand<DBM, AK> LA;
  typedef DuoOperand<BN, BKC> LB;
#ifdef MI_DUO_TIMING
  constexpr int lds_bytes = 2 * (LA::FLOATS + LB::FLOATS) * (int)sizeof(float) + 8 * 64 * 8;
#else
  constexpr int lds_bytes = 2 * (LA::FLOATS + LB::FLOATS) * (int)sizeof(float);
#endif
  static_assert(lds_bytes <= 160 * 1024, "two halves' operand buffers must fit one CU's LDS");
  auto kern = gemm_f32_duo_kernel<BN, AK, BKC, HAS_BIAS>;
  // (per device and cheap: set on every launch rather than remembered per process)
  MI_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  const int tiles_n = n / BN, tiles_per_item = (m / DBM) * tiles_n;
  const long total = (long)tiles_per_item * batch;
  const long grid = total / 2 < cu_count ? total / 2 : cu_count;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds_bytes, s, A, B, C, k / DBK, lda, ldb, ldc, sA, sB, sC,
                     tiles_n, tiles_per_item, (int)total, bias);
  return mi::check_launch();
}

template <int BN, bool AK, bool BKC>
int launch_duo(const float* A, const float* B, float* C, int m, int n, int k, long lda, long ldb, long ldc, long sA,
               long sB, long sC, int batch, const float* bias, hipStream_t s, int cu_count) {
  return bias ? launch_duo_b<BN, AK, BKC, true>(A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, bias, s, cu_count)
              : launch_duo_b<BN, AK, BKC, false>(A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, bias, s, cu_count);
}

}  // namespace

namespace mi {

// MI_OK: launched.  1: the shape is not made of whole tiles (or too few of them) — the caller takes gemm_f32.hip's tiles.
int launch_gemm_duo(int transa, int transb, int32_t m, int32_t n, int32_t k, const float* A, int64_t lda,
                    int64_t strideA, const float* B, int64_t ldb, int64_t strideB, const float* bias, float* C,
                    int64_t ldc, int64_t strideC, int32_t batch, bool force, hipStream_t s) {
  if (m % DBM != 0 || n % 64 != 0 || k % DBK != 0 || k == 0) return 1;
  if (lda % 4 != 0 || ldb % 4 != 0 || strideA % 4 != 0 || strideB % 4 != 0 || !aligned16(A) || !aligned16(B)) return 1;
  if (lda >= (1 << 21) || ldb >= (1 << 21) || ldc >= (1 << 21)) return 1;  // a tile's byte offsets stay below 2^31
  // the current device's CU count, asked per call and handed down as an argument (a namespace-scope variable written
  // here and read in launch_duo_b raced between host threads and could size one device's grid with another's count)
  int cu_count = 256;
  {
    int dev = 0, cus = 0;
    MI_HIP_TRY(hipGetDevice(&dev));
    MI_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus > 0) cu_count = cus;
  }
  const int bn = (n % 128 == 0) ? 128 : 64;
  const long total = (long)(m / DBM) * (n / bn) * batch;
  if (total > 0x7fffffffL) return 1;
  if (!force && total < 2L * cu_count) return 1;  // a persistent workgroup wants at least one tile per half
  if (total < 2) return 1;
#define MI_DUO(BN_, AK_, BKC_) \
  return launch_duo<BN_, AK_, BKC_>(A, B, C, m, n, k, lda, ldb, ldc, strideA, strideB, strideC, batch, bias, s, cu_count)
  if (bn == 128) {
    if (!transa && !transb) MI_DUO(128, true, false);
    if (!transa && transb) MI_DUO(128, true, true);
    if (transa && !transb) MI_DUO(128, false, false);
    MI_DUO(128, false, true);
  }
  if (!transa && !transb) MI_DUO(64, true, false);
  if (!transa && transb) MI_DUO(64, true, true);
  if (transa && !transb) MI_DUO(64, false, false);
  MI_DUO(64, false, true);
#undef MI_DUO
}

}  // namespace mi
