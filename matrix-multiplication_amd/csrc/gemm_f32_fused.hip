// Two dense fp32 products that share one large operand, in ONE launch that reads it once (gfx950):
//     C1 = A · B1        ([m, k] · [k, n])          C2 = Aᵀ · B2        ([k, m] · [m, n])
// — the backward of the BERT drop-in `scores = cublasTransbMM.apply(q, k)` (reference README.md:69-77,
// matmuls.py:131-152): dQ = dS·K and dK = dSᵀ·Q both stream the same dS (403 MB at B 32, H 12, S 512: more than the
// 256 MiB Infinity Cache), which two launches of gemm_f32.hip read from HBM twice (include/mi_spmm.h:
// mi_gemm_pair_a_at_f32).
//
// Exactness first: every output element of both products is ONE k-ordered fused-multiply-add chain (what
// v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 compute: the oracle's chain, bit for bit) — so a C1 tile must be one
// wave's chain over ALL of k, and a C2 tile one wave's chain over all of m.  The decomposition that keeps both chains
// whole AND all four matrix pipes busy:
//   * a workgroup (8 waves: two per SIMD, 256 registers each) owns one (item, 32-column half of n) and walks A in
//     chunks of 32 rows (32 × k floats in LDS, double-buffered — 132 KB at k = 512; the two chunks after that are in
//     flight in the loader waves' registers);
//   * per chunk the 32 × 32 slice of C1 is finished: four 16×16 tiles; waves 0 and 1 (two different SIMDs) take one
//     16-column strip each = two INDEPENDENT chains of k/4 MFMAs 16x16x4 that share their B1 operand, interleaved —
//     they fill those SIMDs' matrix pipes by themselves.  A's operands come from the LDS chunk one batch of eight
//     k-steps ahead, B1's straight from the L2s (its half, k × 32 floats = 64 KB per unit, is re-read once per chunk: no
//     room in LDS beside 32-row chunks) sixteen k-steps ahead; fully unrolled, so the compiler counts its waits;
//   * the k × 32 slice of C2 lives in the accumulators of waves 2 and 3 on the OTHER two SIMDs (eight 32×32 tiles each
//     at k = 512: 128 registers) for the whole unit; a chunk adds its 32 rows of A: 16 MFMAs 32x32x2 per tile, A read
//     TRANSPOSED out of the same LDS chunk;
//   * waves 4 … 7 (one on every SIMD) move the chunks global → registers → LDS.
// Per chunk and SIMD that is 8192 matrix-pipe cycles on every SIMD at k = 512 (2 × 128 × 32 for a C1 wave, 8 × 16 × 64 for
// a C2 wave): the launch is bound by the matrix pipes, A is read from HBM once (its second reader — the other column
// half of the same item — runs on the same XCD right behind the first and finds it in that L2).
// Waves w and w + 4 share a SIMD (MI355X_MICROARCH.md § LDS: a workgroup's waves go to the SIMDs in a cyclic order), which
// is all the role assignment relies on, for speed only.
// History at the C5 shape (tools/bench_fused_pair.py + MI_FUSED_ABL; the two plain products: 0.243 ms): 16 waves, C2 over
// eight of them, six loader waves on the C1 SIMDs, one chunk of look-ahead, a run-time k loop 0.378 ms (the compiler
// drained the LDS queue before every batch of MFMAs, spilled 25 registers per lane into the chunk loop, the loaders'
// integer divisions sat on the C1 chains' SIMDs); 8 waves, templated k, 16-row chunks with B1 in LDS 0.236 ms (compute
// alone 0.212: a barrier, a chain prologue and a store per 4096 pipe cycles).
#include "mi_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FR = 32;        // rows of A per chunk
constexpr int FN = 32;        // columns of n per unit
constexpr int FPAD = 4;       // floats of padding per chunk row: 16-byte aligned rows, conflict-free C1 operand reads
constexpr int FTHREADS = 512;
constexpr int FLOADERS = 256;
#ifndef MI_FUSED_BAHEAD  // batches of eight k-steps by which the B1 operand loads run ahead of their MFMAs
#define MI_FUSED_BAHEAD 2
#endif
#ifndef MI_FUSED_ABL  // timing-only builds (tools/bench_fused_pair.py): 1 no C1 chains, 2 no C2 tiles, 4 no chunk loads / LDS writes
#define MI_FUSED_ABL 0
#endif

template <int KT>  // k = 64 · KT
__global__ __launch_bounds__(FTHREADS) void gemm_pair_a_at_kernel(const float* __restrict__ A, const float* __restrict__ B1,
                                                                  const float* __restrict__ B2, float* __restrict__ C1,
                                                                  float* __restrict__ C2, int batch, int m, int n) {
  constexpr int k = 64 * KT;
  constexpr int ld = k + FPAD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* as = lds;                       // [2][FR][ld]
  float* b2s = as + 2 * FR * ld;         // [2][FR][FN]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware units: blocks b and b + 8 (same XCD, dispatched one behind the other) take the two column halves of an item
  const unsigned xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const long item = xcd + 8L * (idx >> 1);
  const int half = (int)(idx & 1u);
  if (item >= batch) return;
  const float* Ai = A + item * (long)m * k;
  const float* B1i = B1 + item * (long)k * n + half * FN;
  const float* B2i = B2 + item * (long)m * n + half * FN;
  float* C1i = C1 + item * (long)m * n + half * FN;
  float* C2i = C2 + item * (long)k * n + half * FN;
  const int chunks = m / FR;

  const int lc = lane & 15, lk = lane >> 4;    // 16x16x4 operand lanes
  const int l31 = lane & 31, lhi = lane >> 5;  // 32x32x2 operand lanes

  if (wave >= 4) {
    // ---- loaders: chunk c + 1 goes to LDS while chunk c is multiplied; chunks c + 2 and c + 3 are in flight ----
    const int ltid = tid - 256;
    constexpr int RQ = k / 4;                      // float4 per row of A
    constexpr int RPJ = FLOADERS / RQ;             // rows of A the loaders cover per load step
    constexpr int APT = FR / RPJ;                  // float4 of A per loader thread and chunk (= 2·KT)
    static_assert(FLOADERS % RQ == 0 && FR % RPJ == 0 && FR * (FN / 4) == FLOADERS, "the loaders share a chunk evenly");
    // No vector-ALU instruction per load or LDS write (on gfx950 the fp32 MFMA runs on the vector ALUs: every VALU
    // instruction of a loader wave is taken out of its SIMD's matrix time — with per-load 64-bit address arithmetic the
    // loaders cost the launch 0.04 ms): buffer loads with a loop-invariant lane offset and scalar step / chunk offsets,
    // LDS writes with one lane address per buffer and immediate offsets.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ai), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B2i), 0, 0x7fffffff, 0x00020000);
    const int a_lane = ((ltid / RQ) * k + 4 * (ltid % RQ)) * 4;                 // bytes, loop-invariant
    const int b_lane = ((ltid / (FN / 4)) * n + 4 * (ltid % (FN / 4))) * 4;
    char* const as_lane = reinterpret_cast<char*>(as) + ((ltid / RQ) * ld + 4 * (ltid % RQ)) * 4;
    char* const b2_lane = reinterpret_cast<char*>(b2s) + 16 * ltid;
    f32x4 ra[2][APT], rb[2];
    auto load_chunk = [&](int set, int c) {
      const int cc = c < chunks ? c : chunks - 1;  // past the end: a copy of the last chunk that nobody multiplies
      const int a_chunk = cc * (FR * k * 4), b_chunk = cc * FR * n * 4;  // uniform
#pragma unroll
      for (int j = 0; j < APT; ++j)  // (default cache policy: the unit of the other column half re-reads A from this L2)
        ra[set][j] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_lane, a_chunk + j * (RPJ * k * 4), 0));
      rb[set] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_lane, b_chunk, 0));
    };
    auto store_chunk = [&](int set, int buf) {
      char* const ab = as_lane + buf * (FR * ld * 4);
#pragma unroll
      for (int j = 0; j < APT; ++j) *reinterpret_cast<f32x4*>(ab + j * (RPJ * ld * 4)) = ra[set][j];
      *reinterpret_cast<f32x4*>(b2_lane + buf * (FR * FN * 4)) = rb[set];
    };
    load_chunk(0, 0);
    store_chunk(0, 0);
    load_chunk(1, 1);
    load_chunk(0, 2);
    __syncthreads();
    for (int c = 0; c < chunks; c += 2) {  // (chunks is even: two chunk steps per trip keep the register sets static)
      if (!(MI_FUSED_ABL & 4)) {
        store_chunk(1, 1);  // chunk c + 1
        load_chunk(1, c + 3);
      }
      __syncthreads();
      if (!(MI_FUSED_ABL & 4)) {
        store_chunk(0, 0);  // chunk c + 2
        load_chunk(0, c + 4);
      }
      __syncthreads();
    }
    return;
  }

  if (wave < 2) {
    // ---- C1: the chunk's 32 rows × 16 columns (16·wave …): two chains of k/4 MFMAs, interleaved, sharing B1's operand ----
    __syncthreads();
    // B1 element (k-step s, lane): row 4s + lk, column 16·wave + lc of the unit's half: buffer loads with a loop-invariant
    // lane offset and a compile-time step (n = 64) — no vector-ALU instruction beside the MFMAs
    const __amdgpu_buffer_rsrc_t b1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B1i), 0, 0x7fffffff, 0x00020000);
    const int b1_lane = (lk * (2 * FN) + 16 * wave + lc) * 4;
    constexpr int b1_step = 4 * (2 * FN) * 4;  // bytes per k-step
    // (plain global loads: B1's operands do not depend on the chunk, so the compiler hoists ALL of them out of the chunk
    // loop — the strip's k × 16 values sit in 128 registers for the whole unit and no chunk re-reads them; buffer loads
    // are not hoisted and measured 0.28 ms against 0.225, a uniform base + 32-bit lane offset 0.243)
#ifndef MI_FUSED_B1BUF
#define MI_FUSED_B1BUF 0
#endif
    const float* const b1_ptr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(B1i) + b1_lane);
    const unsigned b1_lane_elems = (unsigned)b1_lane / 4u;
    auto b1_load = [&](int step) {
      if (MI_FUSED_B1BUF == 1) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b1_rsrc, b1_lane, step * b1_step, 0));
      if (MI_FUSED_B1BUF == 2) return (B1i + step * (b1_step / 4))[b1_lane_elems];  // scalar base + lane offset: no VALU per load
      return b1_ptr[step * (b1_step / 4)];
    };
    for (int c = 0; c < chunks; ++c) {
      if (!(MI_FUSED_ABL & 1)) {
        const float* ap = as + ((c & 1) * FR + lc) * ld + lk;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        constexpr int UB = 8, NB = k / 4 / UB;  // batches of eight k-steps
        constexpr int BA = MI_FUSED_BAHEAD;
        float a0[2][UB], a1[2][UB], bv[BA + 1][UB];  // A one batch ahead (LDS), B1 BA batches ahead (L2)
#pragma unroll
        for (int b = 0; b < BA; ++b)
#pragma unroll
          for (int j = 0; j < UB; ++j)
            if (b < NB) bv[b][j] = b1_load(b * UB + j);
#pragma unroll
        for (int j = 0; j < UB; ++j) {
          a0[0][j] = ap[4 * j];
          a1[0][j] = ap[16 * ld + 4 * j];
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          if (t + BA < NB) {
#pragma unroll
            for (int j = 0; j < UB; ++j) bv[(t + BA) % (BA + 1)][j] = b1_load((t + BA) * UB + j);
          }
          if (t + 1 < NB) {
#pragma unroll
            for (int j = 0; j < UB; ++j) {
              a0[(t + 1) & 1][j] = ap[4 * ((t + 1) * UB + j)];
              a1[(t + 1) & 1][j] = ap[16 * ld + 4 * ((t + 1) * UB + j)];
            }
          }
          __builtin_amdgcn_sched_barrier(0);  // loads first: the scheduler otherwise sinks them to just before their use
#pragma unroll
          for (int j = 0; j < UB; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[t % (BA + 1)][j], a0[t & 1][j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[t % (BA + 1)][j], a1[t & 1][j], acc1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // lane (lc, lk) holds C1[row lc (+16)][16·wave + 4·lk … + 3]
        float* cp = C1i + ((long)c * FR + lc) * n + 16 * wave + 4 * lk;
        __builtin_nontemporal_store(acc0, reinterpret_cast<f32x4*>(cp));
        __builtin_nontemporal_store(acc1, reinterpret_cast<f32x4*>(cp + 16L * n));
      }
      __syncthreads();
    }
    return;
  }

  // ---- C2 (waves 2, 3): tiles kt = (wave − 2) + 2u, u < KT, of 32 rows of C2 each; accumulators live for the whole unit ----
  f32x16 acc2[KT];
#pragma unroll
  for (int u = 0; u < KT; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[u][r] = 0.f;
  __syncthreads();
  for (int c = 0; c < chunks; ++c) {
    if (!(MI_FUSED_ABL & 2)) {
      const float* bp = b2s + ((c & 1) * FR + lhi) * FN + l31;
      const float* ap = as + ((c & 1) * FR + lhi) * ld + 32 * (wave - 2) + l31;
      constexpr int KS = FR / 2;  // MFMAs per tile and chunk
      float bv[KS], av[2][KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) bv[s] = bp[2 * s * FN];
#pragma unroll
      for (int s = 0; s < KS; ++s) av[0][s] = ap[2 * s * ld];
#pragma unroll
      for (int u = 0; u < KT; ++u) {
        if (u + 1 < KT) {
#pragma unroll
          for (int s = 0; s < KS; ++s) av[(u + 1) & 1][s] = ap[2 * s * ld + 64 * (u + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc2[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u & 1][s], bv[s], acc2[u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    const int kt = (wave - 2) + 2 * u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      __builtin_nontemporal_store(acc2[u][r], C2i + (long)row * n + l31);
    }
  }
}

}  // namespace

extern "C" {

int mi_gemm_pair_a_at_f32(const float* A, const float* B1, const float* B2, float* C1, float* C2, int32_t batch,
                          int32_t m, int32_t k, int32_t n, mi_stream_t stream) {
  if (batch < 0 || m < 0 || k < 0 || n < 0) return MI_EINVAL;
  if (batch == 0 || m == 0 || k == 0 || n == 0) return 1;  // nothing this form is for: the caller's two plain products
  if (!A || !B1 || !B2 || !C1 || !C2) return MI_EINVAL;
  // the shapes the fused form covers (BERT-base / -large attention and their halves): n = 64, 64 | k ≤ 512, 64 | m
  if (n != 2 * FN || k % 64 != 0 || k > 512 || m % (2 * FR) != 0) return 1;
  // the loaders form 32-bit byte offsets into one item's operands (raw buffer loads, 2 GiB records): beyond that the two
  // plain products (advisor, round 4)
  if ((int64_t)m * k * 4 >= (1LL << 31) || (int64_t)m * n * 4 >= (1LL << 31)) return 1;
  if (!mi::aligned16(A) || !mi::aligned16(B1) || !mi::aligned16(B2) || !mi::aligned16(C1) || !mi::aligned16(C2)) return 1;
  const long blocks = 8L * ((batch + 7) / 8) * 2;
  if (blocks > 0x7fffffffL) return 1;
  const size_t lds = (2 * (size_t)FR * (k + FPAD) + 2 * (size_t)FR * FN) * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MI_PAIR(T_)                                                                                                   \
  case T_: {                                                                                                          \
    auto kern = gemm_pair_a_at_kernel<T_>;                                                                            \
    if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(FTHREADS), lds, s, A, B1, B2, C1, C2, (int)batch, (int)m, (int)n); \
  } break
  switch (k / 64) {  // (k = 64, 128, 256 or 512: the loaders' rows divide evenly)
    MI_PAIR(1);
    MI_PAIR(2);
    MI_PAIR(4);
    MI_PAIR(8);
    default: return 1;
  }
#undef MI_PAIR
  return mi::check_launch();
}

}  // extern "C"
