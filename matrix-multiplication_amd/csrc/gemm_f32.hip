// Dense fp32 (batched, transposed) product for gfx950 — the MI355X counterpart
// of the reference's cuBLAS entry points cublas_mm_wrapper / cublas_bmm_wrapper
// (src/baseline_mm.cu:52-102, :105-155), i.e. what custom_mm.cublas_mmul /
// cublas_bmm and matmuls.cublas*MM.apply (the README's BERT drop-in,
// cublasTransbMM.apply(q, k)) run on.
//
//   C[b] (m×n, row-major) = op(A[b]) · op(B[b]),  alpha = 1, beta = 0
//
// Design: LDS-tiled, exact-fp32 matrix-core kernel.  v_mfma_f32_32x32x2_f32
// (f32 in / f32 accumulate) is bit-for-bit a k-ordered fmaf chain, so the
// result equals the sequential-k oracle (oracle_gemm_f32) exactly.  A block of
// 256 threads (4 waves, 2×2) owns a BM×BN tile of C; each wave holds TM×TN
// 32×32 accumulators.  The k-tile (BK = 32) of each operand goes through LDS in
// the layout that makes both the global read (16 B per lane along the
// operand's contiguous dimension) and the MFMA operand read (32 lanes on 32
// consecutive m or n at one k) bank-conflict free:
//    source contiguous along k  → Xs[mn][BK+1]   (A not transposed, B transposed)
//    source contiguous along mn → Xs[BK][BMN]    (A transposed, B not transposed)
// The accumulators hold C blocks (lane ↔ column, register ↔ row): the epilogue is 16 dword buffer stores per 32×32
// block straight from the accumulator registers, each writing two whole 128-byte row segments — no LDS staging, no
// vector-ALU instruction (round 3; until then Cᵀ blocks went through an LDS patch).  Work ids are dealt XCD-contiguously
// and in 8-tile column groups, so the workgroups an XCD runs together share operand panels in its L2.
//
// Two kernels share loader, layouts and epilogue (same k order per element → same bits), a third covers one regime:
//  * gemm_f32_kernel (k < 256): one LDS buffer, two barriers per k-tile, the next tile's global
//    loads issued before the MFMAs; 3–4 workgroups per CU overlap each other's phases.  BERT's
//    q·kᵀ (k = 64) is bound by writing the 403 MB of scores and runs here.
//  * gemm_f32_pipe_kernel (k ≥ 256): two LDS buffers, ONE barrier per k-tile.  While the MFMAs
//    of tile t run, the registers holding tile t+1 go to the other buffer (dealt 2 LDS writes per
//    MFMA) and the loads of tile t+3 are issued (1 per MFMA; two register sets, so a load has
//    almost two k-tiles to land); operand reads run one 8-MFMA batch ahead.  The issue order is pinned with sched_group_barrier/sched_barrier: left alone, hipcc
//    sinks the loads to the end of the tile, directly in front of the waits on them.
//  * gemm_f32_t16_tn_kernel (Aᵀ·B with few output tiles and a long k — weight gradients): tiles of 16×16 MFMA
//    blocks (v_mfma_f32_16x16x4_f32, the same k-ordered chain) sized to fill the 256 CUs in whole rounds — 96×96
//    with twelve waves, 64×64 with eight, 32×32 with four — both operands streamed through LDS as they lie in
//    memory; its own section below.
//  * gemm_f32_pair_kernel (k = 32 / 64 / 128 on whole 128×128 tiles — q·kᵀ): a chain of 2–4 output tiles per workgroup,
//    each tile's epilogue inside the next tile's MFMAs; its own section below.
// Measured (MI355X, fp32 MFMA peak 157 TFLOP/s; tools/probes/duo_probe.cpp, profiles/r03_*): 8192³ A·Bᵀ 7.20 ms =
// 152.6 TFLOP/s, 4096³ 0.91–0.92 ms (150), BERT-base attention (B 32, H 12, S 512, D 64) q·kᵀ 0.118 ms, probs·V
// 0.115–0.120 ms; DESIGN.md §3.3 has the history and the ablations.
//
// Translation units: the kernels of ONE transposition case are compiled per file (gemm_f32_nn.hip … _tt.hip define
// MI_GEMM_TU_NAME / _TA / _TB and include this file: four parallel compiles instead of one 2½-minute one); this
// file compiled on its own carries only the C-ABI entry points.  Developer probes that include it directly define
// MI_GEMM_SINGLE_TU and get everything in one unit.
#include <atomic>
#include <type_traits>

#include "mi_common.h"

#ifndef MI_GEMM_N16_ALL
#define MI_GEMM_N16_ALL 0  // 1: the 16 × 16-block kernel for every 64 < n ≤ 96 (developer A/B; default: only where it computes fewer columns than the 96-wide tile, n ≤ 80)
#endif
#ifndef MI_GEMM_STORE_AUX
#define MI_GEMM_STORE_AUX 2  // cache policy of the C stores: 2 = non-temporal, 0 = default (developer probes)
#endif
#ifndef MI_GEMM_ABL
#define MI_GEMM_ABL 0  // developer probes only (tools/probes/*): 1 no C stores, 2 no MFMAs, 4 no operand loads; pipelined kernel also
                       // 8 no LDS operand reads, 16 no LDS writes, 32 no barriers (timing only: wrong results)
#endif

#define MI_GEMM_TU_ARGS                                                                                              \
  const float *A, const float *B, float *C, int m, int n, int k, long lda, long ldb, long ldc, long sA, long sB, long sC, \
      int batch, bool vecA, bool vecB, bool vecC, const float *bias, hipStream_t s
namespace mi {
extern std::atomic<unsigned> g_gemm_launch_counter;  // one counter for every launch of every unit (see launch())
int gemm_f32_tu_nn(MI_GEMM_TU_ARGS);
int gemm_f32_tu_nt(MI_GEMM_TU_ARGS);
int gemm_f32_tu_tn(MI_GEMM_TU_ARGS);
int gemm_f32_tu_tt(MI_GEMM_TU_ARGS);
}  // namespace mi

#if defined(MI_GEMM_TU_NAME) || defined(MI_GEMM_SINGLE_TU)
namespace {

using mi::f32x4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

// One operand's k-tile: EXT×BK elements (EXT = BM or BN), 256 threads.
// KCONTIG: element (e, k) lives at src[e*ld + k]; else at src[k*ld + e].
template <int EXT, bool KCONTIG>
struct TileLoader {
  static constexpr int VECS = EXT * BK / 4 / 256;  // float4 per thread
  static constexpr int LDS_LD = KCONTIG ? (BK + 1) : EXT;
  static constexpr int LDS_FLOATS = KCONTIG ? EXT * (BK + 1) : BK * EXT;
  // instruction counts as hipcc emits them (for the issue-order hints of the pipelined kernel):
  // LDS writes per k-tile (ds_write2_b32 pairs / ds_write_b128), LDS reads per 32-row operand
  // block and two k-steps (hipcc pairs them into one ds_read2_b32 in both layouts)
  static constexpr int WRITES = KCONTIG ? 2 * VECS : VECS;
  static constexpr int READS_PER_BATCH = 1;

  // (e, k) of the first element of this thread's v-th float4.
  static __device__ __forceinline__ void coords(int v, int tid, int& e, int& k) {
    const int idx = v * 256 + tid;
    if (KCONTIG) {
      e = idx / (BK / 4);
      k = (idx % (BK / 4)) * 4;
    } else {
      k = idx / (EXT / 4);
      e = (idx % (EXT / 4)) * 4;
    }
  }

  // Global → registers; zero outside [0,ext_lim) × [0,k_lim).
  static __device__ __forceinline__ void load(f32x4 (&r)[VECS], const float* src, long ld, int e0,
                                              int k0, int ext_lim, int k_lim, bool vec_ok, int tid) {
#pragma unroll
    for (int v = 0; v < VECS; ++v) {
      int e, k;
      coords(v, tid, e, k);
      e += e0;
      k += k0;
      f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
      if (KCONTIG) {
        if (e < ext_lim) {
          const float* p = src + (long)e * ld + k;
          if (vec_ok && k + 3 < k_lim) {
            x = *reinterpret_cast<const f32x4*>(p);
          } else {
            if (k + 0 < k_lim) x.x = p[0];
            if (k + 1 < k_lim) x.y = p[1];
            if (k + 2 < k_lim) x.z = p[2];
            if (k + 3 < k_lim) x.w = p[3];
          }
        }
      } else {
        if (k < k_lim) {
          const float* p = src + (long)k * ld + e;
          if (vec_ok && e + 3 < ext_lim) {
            x = *reinterpret_cast<const f32x4*>(p);
          } else {
            if (e + 0 < ext_lim) x.x = p[0];
            if (e + 1 < ext_lim) x.y = p[1];
            if (e + 2 < ext_lim) x.z = p[2];
            if (e + 3 < ext_lim) x.w = p[3];
          }
        }
      }
      r[v] = x;
    }
  }

  // Interior tile (every element in range, 16-B aligned rows): unconditional buffer loads whose address is a scalar
  // descriptor (the tile's origin: uniform) + a scalar row step + ONE loop-invariant lane offset — no vector address
  // arithmetic in the k-loop.  On gfx950 the fp32 MFMA runs at the vector-ALU rate and a co-resident wave's VALU
  // instruction waits for a gap in the multiplying wave's stream (≈50 cycles each, tools/probes/duo_probe.cpp), so
  // the ≈3.5 address instructions per flat global_load of the first version were not free.
  static constexpr int ROWV = KCONTIG ? BK / 4 : EXT / 4;  // float4 per source row
  static constexpr int RPV = 256 / ROWV;                    // source rows per float4-per-thread pass
  static __device__ __forceinline__ unsigned lane_offset(long ld, int tid) {  // bytes, loop-invariant
    return (unsigned)((tid / ROWV) * (int)ld + (tid % ROWV) * 4) * 4u;
  }
  static __device__ __forceinline__ const float* origin(const float* src, long ld, int e0, int k0) {  // uniform
    return KCONTIG ? src + (long)e0 * ld + k0 : src + (long)k0 * ld + e0;
  }
  static __device__ __forceinline__ void load_fast(f32x4 (&r)[VECS], const float* tile, long ld, unsigned lane_off) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), 0, 0x7fffffff, 0x00020000);
    const unsigned step = (unsigned)(RPV * (int)ld) * 4u;
#pragma unroll
    for (int v = 0; v < VECS; ++v)
      r[v] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)(v * step), 0));
  }

  // The same loads on a tile that may reach beyond the operand (an edge tile of C): one loop-invariant byte offset
  // per float4 instead of one per lane + a scalar step, each CLAMPED to the operand — a row of the tile beyond the
  // last row re-reads the last row, a float4 beyond the end of a k-row (operands contiguous along ext) re-reads the
  // row's last float4.  The duplicates only reach rows / columns of C that are never stored; nothing outside the
  // operand is touched (but the ≤ 3 floats of row padding behind a ragged extent: ld ≥ ext rounded up to 4).  Still
  // no vector-ALU instruction in the k-loop.  `left` = extent of the operand from the tile's origin (≥ 1).  The k
  // extent must be whole tiles.  (Clamping through the descriptor's range check instead — num_records up to the
  // operand's end — cost the 4096³ product 2 %: ten scalar instructions per tile and operand.)
  static __device__ __forceinline__ void lane_offsets(unsigned (&off)[VECS], long ld, int tid, int left) {
#pragma unroll
    for (int v = 0; v < VECS; ++v) {
      const int idx = v * 256 + tid;
      if (KCONTIG) {
        int e = idx / (BK / 4);
        e = e < left ? e : left - 1;
        off[v] = (unsigned)(e * (int)ld + (idx % (BK / 4)) * 4) * 4u;
      } else {
        const int quads = (left + 3) / 4;
        int q = idx % (EXT / 4);
        q = q < quads ? q : quads - 1;
        off[v] = (unsigned)((idx / (EXT / 4)) * (int)ld + q * 4) * 4u;
      }
    }
  }
  static __device__ __forceinline__ void load_clamped(f32x4 (&r)[VECS], const float* tile, const unsigned (&off)[VECS]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int v = 0; v < VECS; ++v)
      r[v] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off[v], 0, 0));
  }

  // Operands whose rows are NOT 16-byte aligned (ld % 4 ≠ 0, a base or batch stride off 16 bytes — ViT's 197 / 577 tokens:
  // probs is 197 × 197) take the same unconditional buffer loads: along k (KCONTIG) a float4 of a whole k-tile lies inside
  // its row whatever the row's alignment, so load_clamped serves as it is (a dword-aligned 16-byte load); along ext a row's
  // last quad would reach into the next row — or past the operand's end — so every ELEMENT is clamped to the extent and
  // loaded as a dword (round 5; until then such products took the bounds-checked loader for every tile:
  // 384 × (197 × 197)·(197 × 64) 0.040 ms against 0.029 at 200 tokens, torch 0.036).
  static constexpr int DWV = KCONTIG ? 1 : VECS;
  static __device__ __forceinline__ void lane_offsets_dw(unsigned (&off)[DWV][4], long ld, int tid, int left) {
    if constexpr (!KCONTIG) {
#pragma unroll
      for (int v = 0; v < VECS; ++v) {
        const int idx = v * 256 + tid;
        const int row = idx / (EXT / 4), e = (idx % (EXT / 4)) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) off[v][c] = (unsigned)(row * (int)ld + (e + c < left ? e + c : left - 1)) * 4u;
      }
    }
  }
  static __device__ __forceinline__ void load_clamped_dw(f32x4 (&r)[VECS], const float* tile, const unsigned (&off)[DWV][4]) {
    if constexpr (!KCONTIG) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int v = 0; v < VECS; ++v) {
        r[v].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off[v][0], 0, 0));
        r[v].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off[v][1], 0, 0));
        r[v].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off[v][2], 0, 0));
        r[v].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off[v][3], 0, 0));
      }
    }
  }

  // Registers → LDS.
  static __device__ __forceinline__ void store(const f32x4 (&r)[VECS], float* lds, int tid) {
#pragma unroll
    for (int v = 0; v < VECS; ++v) {
      int e, k;
      coords(v, tid, e, k);
      if (KCONTIG) {
        float* p = lds + e * LDS_LD + k;  // odd row stride: 4 scalar stores, conflict-free
        p[0] = r[v].x;
        p[1] = r[v].y;
        p[2] = r[v].z;
        p[3] = r[v].w;
      } else {
        *reinterpret_cast<f32x4*>(lds + k * LDS_LD + e) = r[v];  // 16-B aligned rows
      }
    }
  }

  // one of the thread's float4 (the pair kernel spreads a stage's LDS writes over the MFMA steps of the previous one)
  template <int V>
  static __device__ __forceinline__ void store_one(const f32x4 (&r)[VECS], float* lds, int tid) {
    if constexpr (V < VECS) {
      int e, k;
      coords(V, tid, e, k);
      if (KCONTIG) {
        float* p = lds + e * LDS_LD + k;
        p[0] = r[V].x;
        p[1] = r[V].y;
        p[2] = r[V].z;
        p[3] = r[V].w;
      } else {
        *reinterpret_cast<f32x4*>(lds + k * LDS_LD + e) = r[V];
      }
    }
  }

  // MFMA operand element (e, k) from LDS.
  static __device__ __forceinline__ float at(const float* lds, int e, int k) {
    return KCONTIG ? lds[e * LDS_LD + k] : lds[k * LDS_LD + e];
  }
};

// Tile order inside one product: column groups of 8 tiles, walked row by row, so the ≈64
// workgroups an XCD runs at a time form an 8×8 patch (8 A panels + 8 B panels stream through its
// L2) instead of one tile row (1 A panel + every B panel).  Bijective; speed only.
__device__ __forceinline__ void tile_coords(int tile, int tiles_n, int tiles_m, int& tile_m, int& tile_n) {
  constexpr int G = 8;
  const int per_group = G * tiles_m;
  const int g = tile / per_group, r = tile % per_group;
  const int width = tiles_n - g * G < G ? tiles_n - g * G : G;
  tile_m = r / width;
  tile_n = g * G + r % width;
}

// Epilogue shared by the one-tile kernels.  The accumulators hold C blocks with lane ↔ column: register r of a
// 32×32 block is row (r&3) + 8·(r>>2) + 4·(lane>>5), column lane&31, so one store instruction of one register
// writes two whole 128-byte row segments.  16 dword stores per block, no LDS staging, no waits; on whole tiles
// (ALIGNED) they are buffer stores — scalar descriptor of the block + scalar row offset + one loop-invariant lane
// offset — so the epilogue costs no vector-ALU instruction either (+ bias[n], fused, when given).  The first
// version kept Cᵀ in the accumulators and staged every block through an LDS patch to store 16 B per lane: the
// streaming-store ceiling is the same for dword and dwordx4 stores (tools/probes/store_bw_probe.cpp: 5.1–5.5 TB/s
// either way), and the patch cost 8 LDS instructions, two waits and ≈20 address instructions per block.
// MODE 0: whole tile.  MODE 1: edge tile of a product whose rows are 16-byte aligned (ldc < 2²⁴): the same buffer
// stores, rows beyond m dropped by the descriptor's range check (records end with row m − 1), columns beyond n by the
// lane mask.  MODE 2: anything else — bounds-checked scalar stores with 64-bit addresses.
template <int BM, int BN, int TM, int TN, int MODE, bool HAS_BIAS>
__device__ __forceinline__ void gemm_epilogue_b(f32x16 (&acc)[TM][TN], float* __restrict__ C, int m, int n, long ldc,
                                                int m0, int n0, int wm, int wn, int lane, const float* __restrict__ bias) {
  const int l31 = lane & 31, lhi = lane >> 5;
  const unsigned c_lane = (unsigned)(4 * lhi * (int)ldc + l31) * 4u, c_row = (unsigned)ldc * 4u;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row0 = m0 + wm * (TM * 32) + i * 32, col0 = n0 + wn * (TN * 32) + j * 32;  // uniform (a wave owns TM × TN blocks)
      const int col = col0 + l31;
      float bv = 0.f;
      if (HAS_BIAS && (MODE == 0 || col < n)) bv = bias[col];
      if (MODE < 2) {
        int recs = 0x7fffffff;
        if (MODE == 1) {
          const long bytes = ((long)(m - row0 - 1) * ldc + (n - col0)) * 4;  // up to the last stored element of the block
          recs = (row0 >= m || col0 >= n) ? 0 : (bytes > 0x7fffffffL ? 0x7fffffff : (int)bytes);
        }
        if ((MI_GEMM_ABL & 1) && MODE == 0) recs = 0;  // timing only: every store dropped by the range check
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(C + (long)row0 * ldc + col0, 0, recs, 0x00020000);
        if (MODE == 0 || col < n) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][j][r];
            if (HAS_BIAS) v += bv;  // after the chain: one extra rounding, like `output += bias` (never x + 0: keeps −0)
            const unsigned soff = (unsigned)((r & 3) + 8 * (r >> 2)) * c_row;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (int)c_lane, (int)soff, MI_GEMM_STORE_AUX /* nt */);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          float v = acc[i][j][r];
          if (HAS_BIAS) v += bv;
          if (row < m && col < n) __builtin_nontemporal_store(v, C + (long)row * ldc + col);
        }
      }
    }
}

template <int BM, int BN, int TM, int TN, int MODE>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[TM][TN], float* __restrict__ C, int m, int n, long ldc,
                                              int m0, int n0, int wm, int wn, int lane, const float* __restrict__ bias) {
  if (bias) gemm_epilogue_b<BM, BN, TM, TN, MODE, true>(acc, C, m, n, ldc, m0, n0, wm, wn, lane, bias);
  else gemm_epilogue_b<BM, BN, TM, TN, MODE, false>(acc, C, m, n, ldc, m0, n0, wm, wn, lane, bias);
}

#ifdef MI_GEMM_TIMING
__device__ unsigned long long g_gemm_phase[16];
#define GEMM_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g_gemm_phase[k], now_ - stamp_); stamp_ = now_; } while (0)
#else
#define GEMM_STAMP(k) do {} while (0)
#endif

#ifndef MI_GEMM_SHORTK_WAVES
#define MI_GEMM_SHORTK_WAVES 3
#endif
template <int BM, int BN, bool TA, bool TB, bool ALIGNED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MI_GEMM_SHORTK_WAVES, MI_GEMM_SHORTK_WAVES))) void gemm_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int m, int n,
    int k, long lda, long ldb, long ldc, long strideA, long strideB, long strideC, int tiles_n,
    int tiles_per_item, bool vecA, bool vecB, bool vecC, const float* __restrict__ bias, int reverse) {
  constexpr int TM = BM / 64;  // 32×32 tiles per wave along m (2×2 waves)
  constexpr int TN = BN / 64;
  typedef TileLoader<BM, !TA> LA;  // A not transposed → contiguous along k
  typedef TileLoader<BN, TB> LB;   // B transposed     → contiguous along k
  constexpr int kOperandFloats = LA::LDS_FLOATS + LB::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[kOperandFloats];
  float* As = lds;
  float* Bs = lds + LA::LDS_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform: the epilogue's descriptors stay scalar
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware work order: workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD
  // b % 8, each with a private L2), so give every XCD a CONTIGUOUS range of work ids: the tiles
  // that share an A panel (same item, same tile_m) or a B panel then hit in one L2 instead of
  // being fetched by up to 8 of them.  Bijective for any grid size; speed only.
  const unsigned total = gridDim.x, bid = blockIdx.x;
  const unsigned q = total / 8, rem = total % 8, xcd = bid % 8, pos = bid / 8;
  const unsigned work0 = xcd * q + (xcd < rem ? xcd : rem) + pos;
  const unsigned work = reverse ? total - 1u - work0 : work0;  // see launch(): alternate launches walk the items backwards
  const long item = work / tiles_per_item;
  const int tile = work % tiles_per_item;
  int tile_m, tile_n;
  tile_coords(tile, tiles_n, tiles_per_item / tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  A += item * strideA;
  B += item * strideB;
  C += item * strideC;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef MI_GEMM_TIMING
  unsigned long long stamp_ = __builtin_readcyclecounter();
#endif
  f32x4 ra[LA::VECS], rb[LB::VECS];
  const int l31 = lane & 31, lhi = lane >> 5;
  // Launch-uniform: operands whose rows are 16-byte aligned take unconditional buffer loads — on edge tiles too,
  // with lane offsets clamped to the operand (TileLoader::lane_offsets) — for every WHOLE k-tile; only a ragged last
  // k-tile goes through the bounds-checked loader (≈50 scalar / branch instructions per float4), as the prefetch of
  // the last whole one.  Until round 3 every edge tile and every product with a ragged k took that loader for all
  // of its k-tiles (ViT's 197-token attention: 3 of 4 output tiles).
  // ALIGNED: the launcher saw that EVERY tile is whole, and the edge code is compiled out (it is what sets the
  // kernel's register count otherwise)
  // (round 5) rows off 16-byte boundaries take them too — dword-aligned 16-byte loads along k, clamped dword loads along
  // ext (TileLoader::load_clamped_dw) — as long as the 32-bit offsets hold (what vecA / vecB also stand for)
#ifdef MI_GEMM_NO_UNALIGNED_FAST
  const bool fast = ALIGNED || (vecA && vecB);
#else
  const bool fast = ALIGNED || (lda < (1 << 21) && ldb < (1 << 21) && ldc < (1 << 24));
#endif
  const bool whole = ALIGNED || (m0 + BM <= m && n0 + BN <= n);
#ifndef MI_GEMM_KK_UNROLL
#define MI_GEMM_KK_UNROLL 16
#endif
  auto mfma_tile = [&]() {
#pragma unroll MI_GEMM_KK_UNROLL
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = LA::at(As, wm * (BM / 2) + i * 32 + l31, kk + lhi);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = LB::at(Bs, wn * (BN / 2) + j * 32 + l31, kk + lhi);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          // accumulator block = C block: lane ↔ column n, registers ↔ rows (see gemm_epilogue)
          if (!(MI_GEMM_ABL & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
  if (fast) {
    const unsigned a_lane = LA::lane_offset(lda, tid), b_lane = LB::lane_offset(ldb, tid);
    unsigned a_off[LA::VECS], b_off[LB::VECS];
    unsigned a_dw[LA::DWV][4], b_dw[LB::DWV][4];
    // an operand that is contiguous along ext with rows off 16-byte boundaries: element-wise clamped dword loads (uniform)
    const bool a_by_dword = !ALIGNED && TA && !vecA, b_by_dword = !ALIGNED && !TB && !vecB;
    if (!ALIGNED) {
      if (a_by_dword) LA::lane_offsets_dw(a_dw, lda, tid, m - m0);
      else LA::lane_offsets(a_off, lda, tid, m - m0);
      if (b_by_dword) LB::lane_offsets_dw(b_dw, ldb, tid, n - n0);
      else LB::lane_offsets(b_off, ldb, tid, n - n0);
    }
    auto load_tile = [&](int k0) {  // k-tile starting at k0 < k
      if (MI_GEMM_ABL & 4) return;
      if (ALIGNED) {
        LA::load_fast(ra, LA::origin(A, lda, m0, k0), lda, a_lane);
        LB::load_fast(rb, LB::origin(B, ldb, n0, k0), ldb, b_lane);
      } else if (k0 + BK <= k) {
        if (a_by_dword) LA::load_clamped_dw(ra, LA::origin(A, lda, m0, k0), a_dw);
        else LA::load_clamped(ra, LA::origin(A, lda, m0, k0), a_off);
        if (b_by_dword) LB::load_clamped_dw(rb, LB::origin(B, ldb, n0, k0), b_dw);
        else LB::load_clamped(rb, LB::origin(B, ldb, n0, k0), b_off);
      } else {
        LA::load(ra, A, lda, m0, k0, m, k, vecA, tid);
        LB::load(rb, B, ldb, n0, k0, n, k, vecB, tid);
      }
    };
    load_tile(0);
    for (int k0 = 0; k0 < k; k0 += BK) {
      LA::store(ra, As, tid);
      LB::store(rb, Bs, tid);
      GEMM_STAMP(0);  // operand tile landed and written to LDS
      __syncthreads();
      GEMM_STAMP(1);
      if (k0 + BK < k) load_tile(k0 + BK);
      // the loads stay in front of the MFMAs: hipcc otherwise sinks them (they feed nothing until the next
      // iteration) to the end of the tile, right in front of the waits on them — no prefetch left
      __builtin_amdgcn_sched_barrier(0);
      mfma_tile();
      GEMM_STAMP(2);  // MFMAs of the tile issued
      __syncthreads();
      GEMM_STAMP(3);
    }
  } else if (!ALIGNED) {
    LA::load(ra, A, lda, m0, 0, m, k, vecA, tid);
    LB::load(rb, B, ldb, n0, 0, n, k, vecB, tid);
    for (int k0 = 0; k0 < k; k0 += BK) {
      LA::store(ra, As, tid);
      LB::store(rb, Bs, tid);
      __syncthreads();
      if (k0 + BK < k) {
        LA::load(ra, A, lda, m0, k0 + BK, m, k, vecA, tid);
        LB::load(rb, B, ldb, n0, k0 + BK, n, k, vecB, tid);
      }
      mfma_tile();
      __syncthreads();
    }
  }

  if (ALIGNED || (fast && whole)) gemm_epilogue<BM, BN, TM, TN, 0>(acc, C, m, n, ldc, m0, n0, wm, wn, lane, bias);
  else if (fast) gemm_epilogue<BM, BN, TM, TN, 1>(acc, C, m, n, ldc, m0, n0, wm, wn, lane, bias);
  else gemm_epilogue<BM, BN, TM, TN, 2>(acc, C, m, n, ldc, m0, n0, wm, wn, lane, bias);
  GEMM_STAMP(4);  // epilogue issued
}

// Pipelined form: two LDS buffers, ONE barrier per k-tile.  While the MFMAs of tile t run out of
// buffer t&1, the registers holding tile t+1 (loaded during tile t-1) are written to the other
// buffer and the loads of tile t+2 are issued, so a global load has a whole MFMA phase to land and
// a wave that is alone on its SIMD keeps the matrix pipe busy.
// WN: waves along n — 2 (the 2 × 2 layout of every tile so far) or 1 (4 × 1: a wave owns 32 rows × the whole tile width,
// which lets the width be any multiple of 32: the 96-column tile of head sizes 80 / 96, round 5).
template <int BM, int BN, bool TA, bool TB, int WN = 2>
__global__ __launch_bounds__(256) void gemm_f32_pipe_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int m, int n,
    int k, long lda, long ldb, long ldc, long strideA, long strideB, long strideC, int tiles_n,
    int tiles_per_item, bool vecA, bool vecB, bool vecC, const float* __restrict__ bias, int reverse) {
  constexpr int WM = 4 / WN;
  constexpr int TM = BM / (32 * WM);
  constexpr int TN = BN / (32 * WN);
  static_assert(TM * 32 * WM == BM && TN * 32 * WN == BN, "whole 32 x 32 blocks per wave");
  typedef TileLoader<BM, !TA> LA;
  typedef TileLoader<BN, TB> LB;
  constexpr int kStage = LA::LDS_FLOATS + LB::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[2 * kStage];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform: the epilogue's descriptors stay scalar
  const int wm = wave / WN, wn = wave % WN;
  const unsigned total = gridDim.x, bid = blockIdx.x;
  const unsigned q = total / 8, rem = total % 8, xcd = bid % 8, pos = bid / 8;
  const unsigned work0 = xcd * q + (xcd < rem ? xcd : rem) + pos;
  const unsigned work = reverse ? total - 1u - work0 : work0;  // see launch(): alternate launches walk the items backwards
  const long item = work / tiles_per_item;
  const int tile = work % tiles_per_item;
  int tile_m, tile_n;
  tile_coords(tile, tiles_n, tiles_per_item / tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  A += item * strideA;
  B += item * strideB;
  C += item * strideC;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lhi = lane >> 5;
  // launch-uniform: operands with 16-byte aligned rows and whole k-tiles take the pipelined loop below — edge tiles
  // too (lane offsets clamped to the operands: TileLoader::lane_offsets); everything else the simple loop with the
  // bounds-checked loader.  (A ragged last k-tile as one more, bounds-checked, stage of the pipeline was tried: the
  // branch inside the stage costs the straight-line loop 6 % at 4096³.)
  const bool fast = vecA && vecB && k % BK == 0;
  const bool whole = m0 + BM <= m && n0 + BN <= n;
  if (fast) {
    // Two register sets: tile t+1 and tile t+3 travel in one, tile t+2 in the other, so a tile's
    // global loads are issued almost two k-tiles (not one) before they are written to LDS — the
    // MFMA phase of a 128×64 tile is only ≈0.85 µs, shorter than an HBM round trip.
    f32x4 ra0[LA::VECS], rb0[LB::VECS], ra1[LA::VECS], rb1[LB::VECS];
    const int k_last = k - BK;  // start of the last tile: later "prefetches" re-read it (never used)
    unsigned a_off[LA::VECS], b_off[LB::VECS];  // loop-invariant, clamped to the operands: edge tiles need no other care
    LA::lane_offsets(a_off, lda, tid, m - m0);
    LB::lane_offsets(b_off, ldb, tid, n - n0);
    auto load_tile = [&](f32x4 (&ra)[LA::VECS], f32x4 (&rb)[LB::VECS], int k0) {
      const int kk = k0 < k_last ? k0 : k_last;
      if (MI_GEMM_ABL & 4) return;
      // default cache policy: nt / sc loads measured 3-5 % slower here
      LA::load_clamped(ra, LA::origin(A, lda, m0, kk), a_off);
      LB::load_clamped(rb, LB::origin(B, ldb, n0, kk), b_off);
    };
    load_tile(ra0, rb0, 0);
    LA::store(ra0, lds, tid);
    LB::store(rb0, lds + LA::LDS_FLOATS, tid);
    load_tile(ra1, rb1, BK);       // odd tiles travel in set 1
    load_tile(ra0, rb0, 2 * BK);   // even tiles in set 0
    __syncthreads();

    // One k-tile: MFMAs out of LDS buffer `buf`; the set holding tile t+1 goes to the other buffer
    // and is refilled with tile t+3.
    auto k_tile = [&](int k0, int buf, f32x4 (&ra)[LA::VECS], f32x4 (&rb)[LB::VECS]) {
      const float* As = lds + buf * kStage;
      const float* Bs = As + LA::LDS_FLOATS;
      float* An = lds + (buf ^ 1) * kStage;
      float* Bn = An + LA::LDS_FLOATS;
      // operand registers for two k-steps (one "batch" = 2·TM·TN MFMAs), double-buffered: the
      // reads of batch s+1 are issued before the MFMAs of batch s
      float a[2][2][TM], b[2][2][TN];
      auto read_batch = [&](int s, int slot) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[slot][h][i] = (MI_GEMM_ABL & 8) ? (float)(s + i) : LA::at(As, wm * (TM * 32) + i * 32 + l31, 4 * s + 2 * h + lhi);
#pragma unroll
          for (int j = 0; j < TN; ++j) b[slot][h][j] = (MI_GEMM_ABL & 8) ? (float)(h + j) : LB::at(Bs, wn * (TN * 32) + j * 32 + l31, 4 * s + 2 * h + lhi);
        }
      };
      read_batch(0, 0);
#pragma unroll
      for (int s = 0; s < BK / 4; ++s) {
        if (s + 1 < BK / 4) read_batch(s + 1, (s + 1) & 1);
        if (s == 0) {
          // tile t+1: registers → the other buffer, whose last readers passed the previous
          // barrier.  After the last tile this writes a buffer nobody reads.
          if (!(MI_GEMM_ABL & 16)) {
            LA::store(ra, An, tid);
            LB::store(rb, Bn, tid);
          } else {
            asm volatile("" :: "v"(ra[0]), "v"(rb[0]));  // keeps the loads live
          }
        }
        if (s == 1) load_tile(ra, rb, k0 + 3 * BK);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              if (!(MI_GEMM_ABL & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][h][i], b[s & 1][h][j], acc[i][j], 0, 0, 0);
        // Fence the scheduler per batch: hipcc otherwise sinks the operand reads next to their
        // use and the global loads to the end of the tile (right in front of the waits on them).
        if (s == 0) {
          __builtin_amdgcn_sched_group_barrier(0x100, LA::READS_PER_BATCH * TM + LB::READS_PER_BATCH * TN, 0);
#pragma unroll
          for (int q = 0; q < 2 * TM * TN; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, (LA::WRITES + LB::WRITES + 2 * TM * TN - 1) / (2 * TM * TN), 0);
          }
        }
        if (s == 1) {
          __builtin_amdgcn_sched_group_barrier(0x100, LA::READS_PER_BATCH * TM + LB::READS_PER_BATCH * TN, 0);
#pragma unroll
          for (int q = 0; q < 2 * TM * TN; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, (LA::VECS + LB::VECS + 2 * TM * TN - 1) / (2 * TM * TN), 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(MI_GEMM_ABL & 32)) __syncthreads();
    };
    for (int k0 = 0; k0 < k; k0 += 2 * BK) {
      k_tile(k0, 0, ra1, rb1);                        // even tile: tile t+1 is odd → set 1
      if (k0 + BK < k) k_tile(k0 + BK, 1, ra0, rb0);  // odd tile: tile t+1 is even → set 0
    }
  } else {
    f32x4 ra[LA::VECS], rb[LB::VECS];
    LA::load(ra, A, lda, m0, 0, m, k, vecA, tid);
    LB::load(rb, B, ldb, n0, 0, n, k, vecB, tid);
    for (int k0 = 0; k0 < k; k0 += BK) {
      float* As = lds;
      float* Bs = lds + LA::LDS_FLOATS;
      LA::store(ra, As, tid);
      LB::store(rb, Bs, tid);
      __syncthreads();
      if (k0 + BK < k) {
        LA::load(ra, A, lda, m0, k0 + BK, m, k, vecA, tid);
        LB::load(rb, B, ldb, n0, k0 + BK, n, k, vecB, tid);
      }
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = LA::at(As, wm * (TM * 32) + i * 32 + l31, kk + lhi);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = LB::at(Bs, wn * (TN * 32) + j * 32 + l31, kk + lhi);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  if (fast && whole) gemm_epilogue<BM, BN, TM, TN, 0>(acc, C, m, n, ldc, m0, n0, wm, wn, lane, bias);
  else if (fast) gemm_epilogue<BM, BN, TM, TN, 1>(acc, C, m, n, ldc, m0, n0, wm, wn, lane, bias);
  else gemm_epilogue<BM, BN, TM, TN, 2>(acc, C, m, n, ldc, m0, n0, wm, wn, lane, bias);
}

// ---------------------------------------------------------------------------------------------
// Short k, whole tiles: a CHAIN of two or four output tiles of one tile row per workgroup, (tile_m, CHAIN·g …
// CHAIN·g + CHAIN − 1), so that a tile's epilogue — 16 dword buffer stores per 32×32 block, straight from the
// accumulator registers — is issued between the MFMAs of the NEXT tile instead of after its own.  With k = 64 a
// wave has only 128 MFMAs per tile; in the one-tile kernel their 0.082 ms of matrix-pipe time and the staging +
// epilogue skeleton simply add up (tools/probes/gemm_probe.cpp), because co-resident workgroups run in step.
// Here the skeleton of tile T rides inside tile T + 1's MFMA stream of the SAME wave (two accumulator sets), and
// the chain's stages form one prefetch chain through two LDS buffers (the next tile's first operands are
// requested during this tile's last MFMAs).  Same k order per element as every other kernel here → same bits.
// NK = k / 32 is a template parameter so that the epilogue phases land at fixed places of the unrolled MFMA stream.
// ---------------------------------------------------------------------------------------------
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int BM, int BN, bool TA, bool TB, int NK, int CHAIN>
__global__ __launch_bounds__(256) void gemm_f32_pair_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int m, int n, int k, long lda,
    long ldb, long ldc, long strideA, long strideB, long strideC, int groups_n, int groups_per_item, int reverse) {
  constexpr int TM = BM / 64, TN = BN / 64, NT = TM * TN;
  constexpr int PHASES = 2 * NT;            // per tile: each 32×32 block leaves in two phases of 8 dword stores
  static_assert(PHASES % NK == 0 && (BK / 2) % (PHASES / NK) == 0, "phases must divide the MFMA steps of a stage");
  constexpr int PPS = PHASES / NK;          // phases per stage
  constexpr int GAP = (BK / 2) / PPS;       // MFMA k-steps between two phases
  typedef TileLoader<BM, !TA> LA;
  typedef TileLoader<BN, TB> LB;
  constexpr int kOperandFloats = LA::LDS_FLOATS + LB::LDS_FLOATS;
  constexpr int STAGES = CHAIN * NK;        // stage G = k-tile G % NK of chain tile G / NK, in LDS buffer G % 2
  __shared__ __attribute__((aligned(16))) float lds[2 * kOperandFloats];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const unsigned total = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = total / 8, rem = total % 8, xcd = bid % 8, pos = bid / 8;
  const unsigned work0 = xcd * q8 + (xcd < rem ? xcd : rem) + pos;
  const unsigned work = reverse ? total - 1u - work0 : work0;
  const long item = work / groups_per_item;
  const int gr = work % groups_per_item;
  int tile_m, group_n;  // the workgroup owns tiles (tile_m, CHAIN·group_n … CHAIN·group_n + CHAIN − 1)
  tile_coords(gr, groups_n, groups_per_item / groups_n, tile_m, group_n);
  const int m0 = tile_m * BM;
  A += item * strideA;
  B += item * strideB;
  C += item * strideC;

  f32x16 acc[2][TM][TN];  // tile T of the chain accumulates in set T % 2
  const unsigned c_lane = (unsigned)(4 * lhi * (int)ldc + l31) * 4u, c_row = (unsigned)ldc * 4u;

  // one epilogue phase of chain tile T: phase 2q + h stores registers 8h … 8h+7 of 32×32 block q of its accumulators
  // (lane ↔ column, register ↔ row: each store instruction writes two whole 128-byte row segments) as buffer
  // stores — scalar descriptor of the block + scalar row offset + the lane's loop-invariant offset: no LDS, no
  // waits, no vector-ALU instruction (products with a bias take the one-tile kernel).  Spread over the next tile's MFMAs, one store per two MFMAs: the
  // 403 MB of q·kᵀ scores leave as a steady stream instead of in bursts.
  auto phase = [&](auto T_, auto P_) {
    constexpr int T = decltype(T_)::value, P = decltype(P_)::value;
    constexpr int q = P / 2, h = P % 2, i = q / TN, j = q % TN, SET = T % 2;
    const int col0 = (CHAIN * group_n + T) * BN + wn * (BN / 2) + j * 32;  // uniform
    const int row0 = m0 + wm * (BM / 2) + i * 32;
    // (MI_GEMM_ABL & 1, timing only: a descriptor of zero records — the range check drops every store, the instruction
    // stream stays as it is)
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(C + (long)row0 * ldc + col0, 0, (MI_GEMM_ABL & 1) ? 0 : 0x7fffffff, 0x00020000);
#pragma unroll
    for (int r = 8 * h; r < 8 * h + 8; ++r) {
      const float v = acc[SET][i][j][r];
      const unsigned soff = (unsigned)((r & 3) + 8 * (r >> 2)) * c_row;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, (int)c_lane, (int)soff, MI_GEMM_STORE_AUX /* nt */);
    }
  };

  f32x4 ra[LA::VECS], rb[LB::VECS];
  const unsigned a_lane = LA::lane_offset(lda, tid), b_lane = LB::lane_offset(ldb, tid);
  auto load_stage = [&](auto G_) {  // global → registers: the operands of stage G
    constexpr int T = decltype(G_)::value / NK, KT = decltype(G_)::value % NK;
    LA::load_fast(ra, LA::origin(A, lda, m0, KT * BK), lda, a_lane);
    LB::load_fast(rb, LB::origin(B, ldb, (CHAIN * group_n + T) * BN, KT * BK), ldb, b_lane);
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int WV = LA::VECS > LB::VECS ? LA::VECS : LB::VECS;  // MFMA steps that carry one float4 of each operand to LDS
  static_assert(WV + 2 <= BK / 2, "a stage's LDS writes and the next loads must fit its MFMA steps");

  // Two LDS buffers, ONE barrier per stage: stage G's MFMAs read buffer G % 2 while the registers holding stage G + 1
  // (requested one stage earlier) are written to the other buffer, one float4 per operand per MFMA step, and stage
  // G + 2's loads are issued into the freed registers right after.  (The first round-3 build had one buffer and two
  // barriers per stage: every wave stood still while the workgroup's LDS writes drained — with the epilogue's LDS
  // patch gone, 2 × 33 KB per workgroup fit twice per CU beside the registers' own limit of two.)
  load_stage(std::integral_constant<int, 0>{});
  LA::store(ra, lds, tid);
  LB::store(rb, lds + LA::LDS_FLOATS, tid);
  if constexpr (STAGES > 1) load_stage(std::integral_constant<int, 1>{});
  __syncthreads();

  static_for<STAGES>([&](auto G_) {
    constexpr int G = decltype(G_)::value, T = G / NK, KT = G % NK, SET = T % 2;
    const float* As = lds + (G % 2) * kOperandFloats;
    const float* Bs = As + LA::LDS_FLOATS;
    float* An = lds + ((G + 1) % 2) * kOperandFloats;
    float* Bn = An + LA::LDS_FLOATS;
    // operand reads run one k-step ahead of the MFMAs (two register slots): left to itself the compiler reads
    // each step's operands immediately before its MFMAs and the wave waits out the LDS latency every four MFMAs
    float a[2][TM], b[2][TN];
    auto read_step = [&](int step, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[slot][i] = LA::at(As, wm * (BM / 2) + i * 32 + l31, 2 * step + lhi);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[slot][j] = LB::at(Bs, wn * (BN / 2) + j * 32 + l31, 2 * step + lhi);
    };
    read_step(0, 0);
    static_for<BK / 2>([&](auto S_) {
      constexpr int S = decltype(S_)::value;  // MFMA k-step of this stage
      if constexpr (S + 1 < BK / 2) read_step(S + 1, (S + 1) & 1);
      if constexpr (G + 1 < STAGES && S >= 1 && S <= WV) {
        LA::template store_one<S - 1>(ra, An, tid);
        LB::template store_one<S - 1>(rb, Bn, tid);
      }
      if constexpr (G + 2 < STAGES && S == WV + 1) load_stage(std::integral_constant<int, G + 2>{});
      // the previous tile's epilogue rides here (its accumulators are the other set)
      if constexpr (T > 0 && S % GAP == 0) phase(std::integral_constant<int, T - 1>{}, std::integral_constant<int, KT * PPS + S / GAP>{});
      // (the issue order above is the order wanted: hipcc otherwise sinks the loads to the end of the stage, right
      // in front of the waits on them — the ISA of the first round-3 build showed exactly that)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (MI_GEMM_ABL & 2) {
            if (KT == 0 && S == 0) acc[SET][i][j] = zero16;
          } else if constexpr (KT == 0 && S == 0) {
            // a tile's first product starts its accumulators (set reused from tile T − 2) from zero
            acc[SET][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S & 1][i], b[S & 1][j], zero16, 0, 0, 0);
          } else {
            acc[SET][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S & 1][i], b[S & 1][j], acc[SET][i][j], 0, 0, 0);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (G + 1 < STAGES) __syncthreads();
  });
  // the last tile's own epilogue (nothing left to hide it behind)
  static_for<PHASES>([&](auto P_) { phase(std::integral_constant<int, CHAIN - 1>{}, P_); });
}

// ---------------------------------------------------------------------------------------------
// Aᵀ·B with few output tiles and a long k (the weight gradient of an FC layer, dYᵀ·x: m = 3072, n = 768,
// k = 16384 at BERT-base): every output element is ONE k-ordered chain (that is what makes the result
// bit-identical to the oracle), so k cannot be split, and 128×128 tiles give 144 workgroups for 256 CUs —
// 64×64 tiles 576, 2.25 per CU.  96×96 tiles give exactly 256.  A 96×96 tile does not divide into four
// waves' worth of 32×32 MFMA blocks, but into 36 blocks of 16×16: v_mfma_f32_16x16x4_f32 is the same
// k-ordered fmaf chain (tools/probes/mfma16_probe.cpp: 0 of 2²⁰ results differ) at the same flop rate,
// nine blocks (48×48) per wave.  Both operands are contiguous along m / n ([k][m], [k][n]): k-chunks of 32
// rows go through LDS as they lie in memory (row stride 112 floats: the four k-rows an operand read
// touches sit 16 banks apart), double-buffered, one barrier per chunk; the next chunk's six float4 per
// thread are loaded before the chunk's 72 MFMAs and stored after them.  The MFMA operands are swapped
// (D = Bᵀ-block × A-block), so a lane's four accumulator registers are four consecutive n: 16-byte stores.
// Whole tiles only (m, n multiples of 96, k of 64, 16-byte aligned rows): pick_tile() checks.
// ---------------------------------------------------------------------------------------------
#ifndef MI_GEMM_T16_KC
#define MI_GEMM_T16_KC 32
#endif
constexpr int T16_KC = MI_GEMM_T16_KC;
#ifndef MI_GEMM_T16_WAVES
#define MI_GEMM_T16_WAVES 12  // waves per workgroup of the 96×96 tile (4: one wave per SIMD, 0.75 ms at the BERT-base shape)
#endif

// TILE × TILE outputs per workgroup, 16×16 MFMA blocks dealt to WAVES waves so that every SIMD holds 2–4 waves:
//   TILE 128, 16 waves: 32×32 (2×2 blocks) per wave — level with the 32×32-block kernels at 4096 × 1024 / 2048²
//                       outputs (1.12 vs 1.16 ms, 0.55 vs 0.54): instantiable, not dispatched
//   TILE  96, 12 waves: 16×48 (1×3)          per wave — 256 workgroups at 3072 × 768 (BERT-base FFN)
//   TILE  96,  4 waves: 48×48 (3×3), one wave per SIMD: every non-MFMA instruction the wave issues is a gap in its
//                       own MFMA stream (a 16×16×4 MFMA is 32 cycles) — kept for comparison
//   TILE  64,  8 waves: 16×32 (1×2)          per wave — 256 workgroups at 1024² or 2048 × 512 (0.155 vs 0.164 ms)
//   TILE  32,  4 waves: 16×16 (one block)    per wave — tiny outputs (256² with k = 65536: 64 workgroups instead of the
//                       16 of 64×64 tiles; every wave is one dependent MFMA chain, ≈40 cycles per 4 k)
template <int TILE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_f32_t16_tn_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int k, long lda, long ldb,
    long ldc, long strideA, long strideB, long strideC, int tiles_n, int tiles_per_item, int reverse) {
  static_assert((TILE == 128 && WAVES == 16) || (TILE == 96 && (WAVES == 12 || WAVES == 4)) || (TILE == 64 && WAVES == 8) ||
                    (TILE == 32 && WAVES == 4),
                "see the table above");
  constexpr int THREADS = WAVES * 64;
  constexpr int LD = TILE + 16;                 // LDS row stride: the four k-rows an operand read touches sit 16 banks apart
  constexpr int ROWV = TILE / 4;                // float4 per operand row
  constexpr int MB = TILE == 128 ? 2 : ((TILE == 96 && WAVES == 4) ? 3 : 1);  // m blocks per wave
  constexpr int NB = TILE == 96 ? 3 : (TILE == 32 ? 1 : 2);   // n blocks per wave
  constexpr int WM = TILE / 16 / MB;                           // waves along m
  constexpr int VECS = T16_KC * ROWV / THREADS;                // float4 per thread, operand and chunk
  static_assert(VECS * THREADS == T16_KC * ROWV && WM * (TILE / 16 / NB) == WAVES, "tile / wave layout");
  extern __shared__ __attribute__((aligned(16))) float t16_lds[];
  float (*lds)[2][T16_KC * LD] = reinterpret_cast<float (*)[2][T16_KC * LD]>(t16_lds);  // [buffer][operand][k][m or n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m_w = (wave % WM) * (MB * 16), n_w = (wave / WM) * (NB * 16);  // first m / n column of the wave's sub-tile
  // XCD-contiguous work ids, 8-tile column groups: as gemm_f32_kernel
  const unsigned total = gridDim.x, bid = blockIdx.x;
  const unsigned q = total / 8, rem = total % 8, xcd = bid % 8, pos = bid / 8;
  const unsigned work0 = xcd * q + (xcd < rem ? xcd : rem) + pos;
  const unsigned work = reverse ? total - 1u - work0 : work0;
  const long item = work / tiles_per_item;
  const int tile = work % tiles_per_item;
  int tile_m, tile_n;
  tile_coords(tile, tiles_n, tiles_per_item / tiles_n, tile_m, tile_n);
  const float* Ap = A + item * strideA + (long)tile_m * TILE;  // A stored [k][m]
  const float* Bp = B + item * strideB + (long)tile_n * TILE;  // B stored [k][n]

  f32x4 acc[NB][MB];  // [n block][m block]: register r ↔ n = 16·nb + 4·(lane>>4) + r, m = 16·mb + (lane&15)
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // a chunk of an operand: T16_KC rows × ROWV float4; thread t takes float4 t, t + THREADS, ….  Two register
  // sets: chunk c+2 is loaded while chunk c is multiplied and goes to LDS after chunk c+1's MFMAs, so a
  // load has two chunks' time (≈2 µs) to land; with one set (one chunk ahead) the kernel waited on memory
  // a third of the time.  The loads carry no predicate (past the end: the last chunk again), so the
  // compiler's wait before a store to LDS counts the younger loads instead of being vmcnt(0).
  f32x4 ra[2][VECS], rb[2][VECS];
  const int chunks = k / T16_KC;
  // buffer loads: scalar descriptor of the chunk + scalar row step (THREADS / ROWV rows per pass) + one loop-invariant
  // lane offset per operand — no vector address arithmetic between the MFMAs (see TileLoader::load_fast)
  static_assert(THREADS % ROWV == 0, "a pass of float4-per-thread loads covers whole rows");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned a_lane = (unsigned)((tid / ROWV) * (int)lda + 4 * (tid % ROWV)) * 4u;
  const unsigned b_lane = (unsigned)((tid / ROWV) * (int)ldb + 4 * (tid % ROWV)) * 4u;
  const unsigned a_step = (unsigned)((THREADS / ROWV) * (int)lda) * 4u, b_step = (unsigned)((THREADS / ROWV) * (int)ldb) * 4u;
  auto load_chunk = [&](int set, int c) {
    const int k0 = (c < chunks ? c : chunks - 1) * T16_KC;
    const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ap + (long)k0 * lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bp + (long)k0 * ldb), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < VECS; ++i) {
      if (MI_GEMM_ABL & 4) {
        ra[set][i] = rb[set][i] = f32x4{1.f, 2.f, 3.f, (float)k0};
        continue;
      }
      ra[set][i] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(ra_src, (int)a_lane, (int)(i * a_step), 0));
      rb[set][i] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rb_src, (int)b_lane, (int)(i * b_step), 0));
    }
  };
  auto store_chunk = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < VECS; ++i) {
      const int j = tid + THREADS * i, row = j / ROWV, c4 = j % ROWV;
      *reinterpret_cast<f32x4*>(&lds[buf][0][row * LD + 4 * c4]) = ra[set][i];
      *reinterpret_cast<f32x4*>(&lds[buf][1][row * LD + 4 * c4]) = rb[set][i];
    }
  };
  const int lk = lane >> 4, lc = lane & 15;
  // operand reads run one k-step (4 k) ahead of the MFMAs, ACROSS the chunk boundary: the last step's
  // MFMAs of a chunk are issued after the barrier and after the next chunk's first operand reads, so the
  // matrix pipe has work while those reads are in flight
  float a[2][MB], b[2][NB];
  auto read_ops = [&](int set, int buf, int kk) {
    const float* as = &lds[buf][0][(kk * 4 + lk) * LD + m_w + lc];
    const float* bs = &lds[buf][1][(kk * 4 + lk) * LD + n_w + lc];
    if (MI_GEMM_ABL & 8) {  // timing only: no LDS operand reads
#pragma unroll
      for (int i = 0; i < MB; ++i) a[set][i] = (float)(kk + i);
#pragma unroll
      for (int i = 0; i < NB; ++i) b[set][i] = (float)(set + i);
      return;
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) a[set][i] = as[16 * i];
#pragma unroll
    for (int i = 0; i < NB; ++i) b[set][i] = bs[16 * i];
  };
  auto mfmas = [&](int set) {
    if (MI_GEMM_ABL & 2) return;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[set][nb], a[set][mb], acc[nb][mb], 0, 0, 0);
  };
  // one chunk: steps 0 … 6 from LDS buffer `buf`, then the hand-over to the other buffer, then step 7
  auto half = [&](int buf, int next_chunk_to_load, int set_load, int set_store) {
    load_chunk(set_load, next_chunk_to_load);
    __builtin_amdgcn_sched_barrier(0);  // left alone, hipcc sinks the loads behind the MFMAs, in front of the waits on them
#pragma unroll
    for (int kk = 0; kk < T16_KC / 4 - 1; ++kk) {
      read_ops((kk + 1) & 1, buf, kk + 1);
      __builtin_amdgcn_sched_barrier(0);  // reads first: the scheduler otherwise moves them down to just before
      mfmas(kk & 1);                      // their use, and the wave then waits out the LDS latency every step
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    store_chunk(set_store, buf ^ 1);  // the next chunk (past the end: a copy of the last one that nobody multiplies)
    __syncthreads();                  // it is in LDS; every wave is past its reads of this buffer
    read_ops(0, buf ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);  // step 7 of this chunk
    __builtin_amdgcn_sched_barrier(0);
  };

  load_chunk(0, 0);
  store_chunk(0, 0);
  load_chunk(1, 1);
  __syncthreads();
  read_ops(0, 0, 0);
  for (int c = 0; c < chunks; c += 2) {  // an even number of chunks: one straight-line body (an exit between the
    half(0, c + 2, 0, 1);                // halves lets the compiler sink the first half's loads past it)
    half(1, c + 3, 1, 0);
  }
  if (MI_GEMM_ABL & 1) return;
  float* Cp = C + item * strideC + ((long)tile_m * TILE + m_w + lc) * ldc + (long)tile_n * TILE + n_w + 4 * lk;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      __builtin_nontemporal_store(acc[nb][mb], reinterpret_cast<f32x4*>(Cp + (long)(16 * mb) * ldc + 16 * nb));
}

// ---------------------------------------------------------------------------------------------
// Head sizes that are not multiples of 32 (round 5): probs·V (NN) and Pᵀ·dC (TN) with 64 < n ≤ 96 — BERT / ViT variants
// with 72, 80, 88, 96 columns per head.  32 × 32 blocks can only cover them with a 96- or 128-column tile (17 – 37 % of
// the MFMA work thrown away: 1.09 – 1.41 × torch's time, profiles/r03 / r05_attention_shapes.log); 16 × 16 blocks
// (v_mfma_f32_16x16x4_f32: the same k-ordered fmaf chain, the same rate per flop) tile 80 and 96 exactly.
// One workgroup = 128 rows × 16·NB columns (NB = 5 or 6), 8 waves, a wave = 16 rows × all NB blocks, so a k-step of 4
// costs a wave one A read and NB B reads from LDS for NB MFMAs.  Both operands go through LDS as [k][·] images
// (B as it lies in memory; A as it lies for Aᵀ·B, transposed on the way for A·B: a [m][KC + 4] image, whose 36-float
// row stride keeps the 16 rows × 4 k of an operand read on 64 distinct banks).  Two register sets of global loads, one
// barrier per chunk of 32 k, operand reads one k-step ahead — the structure of gemm_f32_t16_tn_kernel.
// Whole 128-row tiles, k % 64 == 0, n % 4 == 0, 16-byte aligned rows: pick_tile() checks.  Columns beyond n (n = 72 on the
// 80-column tile, 88 on 96) read clamped addresses and are never stored.
// ---------------------------------------------------------------------------------------------
template <int NB, bool A_KCONTIG>
__global__ __launch_bounds__(512) void gemm_f32_n16_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int n, int k, long lda, long ldb,
    long ldc, long strideA, long strideB, long strideC, int tiles_per_item, int reverse) {
  constexpr int KC = 32, TM_ = 128, TN_ = 16 * NB, THREADS = 512;
  constexpr int LDA = A_KCONTIG ? KC + 4 : TM_ + 16;  // floats per LDS row of the A image ([m][k] or [k][m])
  constexpr int A_FLOATS = A_KCONTIG ? TM_ * LDA : KC * LDA;
  constexpr int LDB = TN_ + 16;
  constexpr int B_FLOATS = KC * LDB;
  constexpr int VA = TM_ * KC / 4 / THREADS;                      // float4 of A per thread and chunk (2)
  constexpr int BQ = KC * (TN_ / 4);                              // float4 of B per chunk (640 / 768)
  constexpr int VB = (BQ + THREADS - 1) / THREADS;                // per thread (2; the last ones clamped duplicates)
  extern __shared__ __attribute__((aligned(16))) float n16_lds[];
  float* As[2] = {n16_lds, n16_lds + A_FLOATS + B_FLOATS};
  float* Bs[2] = {n16_lds + A_FLOATS, n16_lds + 2 * A_FLOATS + B_FLOATS};
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned total = gridDim.x, bid = blockIdx.x;
  const unsigned q = total / 8, rem = total % 8, xcd = bid % 8, pos = bid / 8;
  const unsigned work0 = xcd * q + (xcd < rem ? xcd : rem) + pos;
  const unsigned work = reverse ? total - 1u - work0 : work0;
  const long item = work / tiles_per_item;
  const int tile_m = work % tiles_per_item;
  const float* Ap = A + item * strideA + (A_KCONTIG ? (long)tile_m * TM_ * lda : (long)tile_m * TM_);
  const float* Bp = B + item * strideB;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  f32x4 acc[NB];  // register r of block nb ↔ row 4·(lane >> 4) + r of the wave's 16 rows … with the operands swapped (below):
                  // r ↔ n = 16·nb + 4·(lane >> 4) + r, m = lane & 15 — four consecutive n per lane: 16-byte stores
#pragma unroll
  for (int i = 0; i < NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // loop-invariant lane offsets (bytes); B's float4 index is clamped to the tile's last one that lies inside the row
  unsigned a_off[VA], b_off[VB];
  const int nq = n >> 2;  // float4 per row of B that exist
#pragma unroll
  for (int i = 0; i < VA; ++i) {
    const int j = tid + THREADS * i;
    if (A_KCONTIG) a_off[i] = (unsigned)((j / (KC / 4)) * (int)lda + 4 * (j % (KC / 4))) * 4u;   // row m, quad along k
    else a_off[i] = (unsigned)((j / (TM_ / 4)) * (int)lda + 4 * (j % (TM_ / 4))) * 4u;           // row k, quad along m
  }
#pragma unroll
  for (int i = 0; i < VB; ++i) {
    int j = tid + THREADS * i;
    j = j < BQ ? j : BQ - 1;
    int c4 = j % (TN_ / 4);
    c4 = c4 < nq ? c4 : nq - 1;
    b_off[i] = (unsigned)((j / (TN_ / 4)) * (int)ldb + 4 * c4) * 4u;
  }
  f32x4 ra[2][VA], rb[2][VB];
  const int chunks = k / KC;
  auto load_chunk = [&](int set, int c) {
    const int k0 = (c < chunks ? c : chunks - 1) * KC;
    const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A_KCONTIG ? Ap + k0 : Ap + (long)k0 * lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bp + (long)k0 * ldb), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < VA; ++i)
      ra[set][i] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(a_src, (int)a_off[i], 0, 0));
#pragma unroll
    for (int i = 0; i < VB; ++i)
      rb[set][i] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(b_src, (int)b_off[i], 0, 0));
  };
  auto store_chunk = [&](int set, int buf) {
#pragma unroll
    for (int i = 0; i < VA; ++i) {
      const int j = tid + THREADS * i;
      if (A_KCONTIG) *reinterpret_cast<f32x4*>(As[buf] + (j / (KC / 4)) * LDA + 4 * (j % (KC / 4))) = ra[set][i];
      else *reinterpret_cast<f32x4*>(As[buf] + (j / (TM_ / 4)) * LDA + 4 * (j % (TM_ / 4))) = ra[set][i];
    }
#pragma unroll
    for (int i = 0; i < VB; ++i) {
      int j = tid + THREADS * i;
      j = j < BQ ? j : BQ - 1;  // (a duplicate of the last float4: same value to the same place)
      *reinterpret_cast<f32x4*>(Bs[buf] + (j / (TN_ / 4)) * LDB + 4 * (j % (TN_ / 4))) = rb[set][i];
    }
  };
  const int lk = lane >> 4, lc = lane & 15;
  const int m_w = wave * 16;
  float a[2], b[2][NB];
  auto read_ops = [&](int set, int buf, int kk) {
    a[set] = A_KCONTIG ? As[buf][(m_w + lc) * LDA + kk * 4 + lk] : As[buf][(kk * 4 + lk) * LDA + m_w + lc];
    const float* bs = Bs[buf] + (kk * 4 + lk) * LDB + lc;
#pragma unroll
    for (int i = 0; i < NB; ++i) b[set][i] = bs[16 * i];
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[set][nb], a[set], acc[nb], 0, 0, 0);
  };
  auto half = [&](int buf, int next_chunk_to_load, int set_load, int set_store) {
    load_chunk(set_load, next_chunk_to_load);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KC / 4 - 1; ++kk) {
      read_ops((kk + 1) & 1, buf, kk + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(kk & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_chunk(set_store, buf ^ 1);
    __syncthreads();
    read_ops(0, buf ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
  };
  load_chunk(0, 0);
  store_chunk(0, 0);
  load_chunk(1, 1);
  __syncthreads();
  read_ops(0, 0, 0);
  for (int c = 0; c < chunks; c += 2) {
    half(0, c + 2, 0, 1);
    half(1, c + 3, 1, 0);
  }
  float* Cp = C + item * strideC + ((long)tile_m * TM_ + m_w + lc) * ldc + 4 * lk;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    if (16 * nb + 4 * lk < n) __builtin_nontemporal_store(acc[nb], reinterpret_cast<f32x4*>(Cp + 16 * nb));
}

template <bool TA, bool TB>
int launch_n16(const float* A, const float* B, float* C, int m, int n, int k, long lda, long ldb, long ldc, long sA, long sB,
               long sC, int batch, hipStream_t s) {
  static_assert(!TB, "B as it lies in memory: [k][n]");
  const long tiles_m = m / 128;
  const long blocks = tiles_m * batch;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  const int rev = (int)(mi::g_gemm_launch_counter.fetch_add(1, std::memory_order_relaxed) & 1u);
  const int nb = n <= 64 ? 4 : (n <= 80 ? 5 : 6);
  const size_t a_floats = TA ? 32 * (128 + 16) : 128 * (32 + 4);
  const size_t lds = 2 * (a_floats + 32 * (16 * (size_t)nb + 16)) * sizeof(float);
#define MI_N16(NB_)                                                                                                        \
  do {                                                                                                                     \
    auto kern = gemm_f32_n16_kernel<NB_, !TA>;                                                                             \
    if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, s, A, B, C, n, k, lda, ldb, ldc, sA, sB, sC, (int)tiles_m, rev); \
  } while (0)
  if (nb == 4) MI_N16(4);
  else if (nb == 5) MI_N16(5);
#if MI_GEMM_N16_ALL
  else MI_N16(6);
#else
  else return MI_EINVAL;  // (88 / 96 columns run the 96-wide tile of 32 × 32 blocks: pick_tile)
#endif
#undef MI_N16
  return mi::check_launch();
}

template <int TILE, int WAVES>
int launch_t16(const float* A, const float* B, float* C, int m, int n, int k, long lda, long ldb, long ldc, long sA,
               long sB, long sC, int batch, int rev, hipStream_t s) {
  constexpr int lds_bytes = 2 * 2 * T16_KC * (TILE + 16) * (int)sizeof(float);
  auto kern = gemm_f32_t16_tn_kernel<TILE, WAVES>;
  if (lds_bytes > 64 * 1024)
    MI_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  const long tiles = (long)(m / TILE) * (n / TILE);
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * batch)), dim3(WAVES * 64), lds_bytes, s, A, B, C, k, lda, ldb, ldc,
                     sA, sB, sC, n / TILE, (int)tiles, rev);
  return mi::check_launch();
}


template <int BM, int BN, bool TA, bool TB>
int launch(const float* A, const float* B, float* C, int m, int n, int k, long lda, long ldb,
           long ldc, long sA, long sB, long sC, int batch, bool vecA, bool vecB, bool vecC,
           const float* bias, hipStream_t s) {
  const long tiles_m = (m + BM - 1) / BM, tiles_n = (n + BN - 1) / BN;
  const long blocks = tiles_m * tiles_n * batch;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  // Boustrophedon over launches: every other product walks its work list backwards.  Two consecutive
  // products that stream the same large operand (the two gradients of one matmul both read the
  // upstream gradient: 403 MB of dScores at BERT-base attention, more than the 256 MiB Infinity Cache)
  // then meet its most recently read half still in the cache instead of evicting it in lock step.
  // Tile order never affects values.
#ifdef MI_GEMM_NO_REVERSE
  const int rev = 0;
#else
  const int rev = (int)(mi::g_gemm_launch_counter.fetch_add(1, std::memory_order_relaxed) & 1u);
#endif
  // long k: the pipelined kernel (its prologue and double LDS buffer pay off from ≈8 k-tiles);
  // short k (BERT q·kᵀ, k = 64): the single-buffer kernel, which keeps 3–4 workgroups per CU
#ifndef MI_GEMM_PIPE_MIN_TILES
#define MI_GEMM_PIPE_MIN_TILES 8  // developer probes may override
#endif
  // (the pipelined kernel's fast loop wants whole k-tiles and 16-byte aligned rows; a long k that is ragged or unaligned —
  // 577 tokens — is better off in the single-buffer kernel, whose whole k-tiles take unconditional loads whatever the
  // alignment and only the ragged last one the bounds-checked loader, than in the pipelined kernel's bounds-checked loop)
#ifdef MI_GEMM_NO_RAGGED_SHORTK
  const bool pipe_ok = true;
#else
  const bool pipe_ok = (vecA && vecB && k % BK == 0) || !(lda < (1 << 21) && ldb < (1 << 21) && ldc < (1 << 24));
#endif
  if (k >= MI_GEMM_PIPE_MIN_TILES * BK && pipe_ok)
    hipLaunchKernelGGL((gemm_f32_pipe_kernel<BM, BN, TA, TB>), dim3((unsigned)blocks), dim3(256), 0, s, A, B, C, m,
                       n, k, lda, ldb, ldc, sA, sB, sC, (int)tiles_n, (int)(tiles_m * tiles_n), vecA, vecB, vecC,
                       bias, rev);
  else if (vecA && vecB && vecC && m % BM == 0 && n % BN == 0 && k % BK == 0) {
#ifndef MI_GEMM_NO_PAIR
#ifndef MI_GEMM_CHAIN_MAX
#define MI_GEMM_CHAIN_MAX 4
#endif
    if (BM == 128 && BN == 128 && (tiles_n % 2 == 0 || tiles_n % 3 == 0) && (k == BK || k == 2 * BK || k == 4 * BK) &&
        bias == nullptr) {
      // CHAIN output tiles of one tile row per workgroup, each tile's epilogue inside the next one's MFMAs
      // (three for tile rows of 3, 9, 15 … tiles: 384 tokens, BERT's SQuAD length)
      const int chain = (tiles_n % 4 == 0 && MI_GEMM_CHAIN_MAX >= 4) ? 4 : (tiles_n % 2 == 0 ? 2 : 3);
      const unsigned gblocks = (unsigned)(blocks / chain);
      const int gn = (int)(tiles_n / chain), gpi = (int)(tiles_m * tiles_n / chain);
#define MI_PAIR(NK_, CH_)                                                                                         \
  hipLaunchKernelGGL((gemm_f32_pair_kernel<128, 128, TA, TB, NK_, CH_>), dim3(gblocks), dim3(256), 0, s, A, B, C, m, n, \
                     k, lda, ldb, ldc, sA, sB, sC, gn, gpi, rev)
      if (chain == 4) {
        if (k == BK) MI_PAIR(1, 4);
        else if (k == 2 * BK) MI_PAIR(2, 4);
        else MI_PAIR(4, 4);
      } else if (chain == 3) {
        if (k == BK) MI_PAIR(1, 3);
        else if (k == 2 * BK) MI_PAIR(2, 3);
        else MI_PAIR(4, 3);
      } else {
        if (k == BK) MI_PAIR(1, 2);
        else if (k == 2 * BK) MI_PAIR(2, 2);
        else MI_PAIR(4, 2);
      }
#undef MI_PAIR
      return mi::check_launch();
    }
#endif
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, true>), dim3((unsigned)blocks), dim3(256), 0, s, A, B, C, m, n, k,
                       lda, ldb, ldc, sA, sB, sC, (int)tiles_n, (int)(tiles_m * tiles_n), vecA, vecB, vecC, bias, rev);
  }
  else
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, false>), dim3((unsigned)blocks), dim3(256), 0, s, A, B, C, m, n, k,
                       lda, ldb, ldc, sA, sB, sC, (int)tiles_n, (int)(tiles_m * tiles_n), vecA, vecB, vecC, bias, rev);
  return mi::check_launch();
}

// 64 < n ≤ 96 with a long k (attention's probs·V and Pᵀ·dC at head sizes 80 / 96): ONE 96-column tile per 128 rows
// instead of a 128-column one — the matrix pipe computes 96 columns for the 80 / 96 wanted, not 128 (round 4 measured
// 1.17–1.41× torch's time on these products, profiles/r03_attention_shapes.log).  Same k order per element: same bits.
template <bool TA, bool TB, int BN_ = 96>
int launch_n96(const float* A, const float* B, float* C, int m, int n, int k, long lda, long ldb, long ldc, long sA, long sB,
               long sC, int batch, bool vecA, bool vecB, bool vecC, const float* bias, hipStream_t s) {
  const long tiles_m = (m + 127) / 128;
  const long blocks = tiles_m * batch;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  const int rev = (int)(mi::g_gemm_launch_counter.fetch_add(1, std::memory_order_relaxed) & 1u);
  hipLaunchKernelGGL((gemm_f32_pipe_kernel<128, BN_, TA, TB, 1>), dim3((unsigned)blocks), dim3(256), 0, s, A, B, C, m, n, k, lda,
                     ldb, ldc, sA, sB, sC, 1, (int)tiles_m, vecA, vecB, vecC, bias, rev);
  return mi::check_launch();
}

template <bool TA, bool TB>
int pick_tile(const float* A, const float* B, float* C, int m, int n, int k, long lda, long ldb,
              long ldc, long sA, long sB, long sC, int batch, bool vecA, bool vecB, bool vecC,
              const float* bias, hipStream_t s) {
  // Largest tile that still gives the chip ≥ 2 workgroups per CU; the 64-wide n tile covers
  // BERT's head dim (probs·V, n = 64) without wasting half the MFMAs, and long-k products with
  // few output tiles (weight gradients: m = 3072, n = 768, k = 16384) drop to 64×64 tiles rather
  // than leave CUs idle — the k order per element is the same for every tile shape.
  auto blocks_for = [&](int bm, int bn) { return (long)((m + bm - 1) / bm) * ((n + bn - 1) / bn) * batch; };
  const long want = 2L * 256;
#ifndef MI_GEMM_NO_T16
  if (TA && !TB && bias == nullptr && vecA && vecB && vecC && k % (2 * T16_KC) == 0 && k >= 16 * T16_KC &&
      blocks_for(128, 128) < want) {
    // Aᵀ·B with few output tiles and a long k (weight gradients): tiles of 16×16 MFMA blocks, one workgroup per CU
    // with 2–4 waves per SIMD, in the tile size that fills the CUs' rounds best (the larger tile on a tie: fewer
    // operand bytes per flop) — see gemm_f32_t16_tn_kernel
    int best = 0;
    double best_eff = 0.0;
    for (int tile : {96, 64}) {  // 128×128 (16 waves of 2×2 blocks) measured level with the 32×32-block kernels: not taken
      if (m % tile != 0 || n % tile != 0) continue;
      const long t = blocks_for(tile, tile);
      if (t > 0x7fffffffL) continue;
      const double eff = (double)t / (double)(((t + 255) / 256) * 256);
      if (eff > best_eff + 1e-9) best_eff = eff, best = tile;
    }
    if (m % 32 == 0 && n % 32 == 0 && blocks_for(64, 64) <= 64) best = 32, best_eff = 1.0;  // a tiny output: most workgroups win
#ifndef MI_GEMM_T16_MIN_EFF
#define MI_GEMM_T16_MIN_EFF 0.75
#endif
    // neither 96- nor 64-tiles fill the CUs' rounds (768², 1536 × 768, 1152² outputs: 0.56–0.63): 32×32 tiles give every
    // SIMD two to five one-block waves instead (round 3, tools/probes/duo_probe.cpp gw*: 768² × 16384 0.287 → 0.261 ms,
    // 1536 × 768 × 8192 0.264 → 0.216, 1152² × 8192 0.267 → 0.249; where 64-tiles give exactly one workgroup per CU they
    // stay ahead: 2048 × 512 × 8192 0.164 vs 0.188)
    if (best_eff < MI_GEMM_T16_MIN_EFF && m % 32 == 0 && n % 32 == 0 && blocks_for(32, 32) >= 256 && blocks_for(64, 64) <= 512)
      best = 32, best_eff = 1.0;
    if (best_eff >= MI_GEMM_T16_MIN_EFF) {
      const int rev = (int)(mi::g_gemm_launch_counter.fetch_add(1, std::memory_order_relaxed) & 1u);
      if (best == 96) return launch_t16<96, MI_GEMM_T16_WAVES>(A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, rev, s);
      if (best == 32) return launch_t16<32, 4>(A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, rev, s);
      return launch_t16<64, 8>(A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, rev, s);
    }
  }
#endif
#define MI_TILE(BM_, BN_) \
  return launch<BM_, BN_, TA, TB>(A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, vecA, vecB, vecC, bias, s)
#ifdef MI_GEMM_FORCE_TILE  // developer probes only
  if (MI_GEMM_FORCE_TILE == 1) MI_TILE(128, 128);
  if (MI_GEMM_FORCE_TILE == 2) MI_TILE(128, 64);
  if (MI_GEMM_FORCE_TILE == 3) MI_TILE(64, 128);
  if (MI_GEMM_FORCE_TILE == 4) MI_TILE(64, 64);
#endif
#ifndef MI_GEMM_NO_N16
  // n = 64 with a LONG k (probs·V / Pᵀ·dC over 2048 tokens): the 16 × 16-block kernel's 128 × 64 tile — 8 waves of 16 rows,
  // two workgroups per CU — is ahead of the 2 × 2-wave tile of 32 × 32 blocks there (48 × 2048² × 64: 0.2375 → 0.219 ms,
  // torch 0.2305; at k = 512 / 1024 the 32 × 32 tiles stay ahead or level: tools/probes/gemm_n64_ab.py)
  if constexpr (!TB) {
    if (n == 64 && k >= 2048 && m % 128 == 0 && k % 64 == 0 && vecA && vecB && vecC && bias == nullptr && blocks_for(128, 64) >= 256)
      return launch_n16<TA, TB>(A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, s);
  }
#endif
#ifndef MI_GEMM_NO_N96
  if (n > 64 && n <= 96 && m > 64 && k >= MI_GEMM_PIPE_MIN_TILES * BK && vecA && vecB && k % BK == 0 && blocks_for(128, 96) >= 256) {
#ifndef MI_GEMM_NO_N16
    // exact 80- / 96-column tiles of 16 × 16 blocks where the shape is made of whole 128-row tiles (probs·V, Pᵀ·dC)
    if constexpr (!TB) {
      // (n ≤ 80: 384 × 512 × 80 probs·V 0.151 → 0.135 ms, × 72 0.159 → 0.142; at 88 / 96 columns the 96-wide tile of 32 × 32 blocks
      // is ahead: 0.153 vs 0.162 ms)
      if (bias == nullptr && vecC && m % 128 == 0 && k % 64 == 0 && n % 4 == 0 && (n <= 80 || MI_GEMM_N16_ALL))
        return launch_n16<TA, TB>(A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, s);
    }
#endif
    return launch_n96<TA, TB>(A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, vecA, vecB, vecC, bias, s);
  }
#endif
  if (m > 64 && n > 64 && blocks_for(128, 128) >= want) MI_TILE(128, 128);
  if (m > 64 && blocks_for(128, 64) >= want) MI_TILE(128, 64);
  if (n > 64 && blocks_for(64, 128) >= want) MI_TILE(64, 128);
  if (blocks_for(64, 64) >= want || (m <= 64 && n <= 64)) MI_TILE(64, 64);
  // small problems: the widest tile the shape fills
  if (m > 64 && n > 64 && blocks_for(64, 64) < 64) MI_TILE(128, 128);
  MI_TILE(64, 64);
#undef MI_TILE
}

// k == 0: C = 0 (beta = 0 semantics), or the bias row.
}  // namespace

#define MI_GEMM_TU_PASS A, B, C, m, n, k, lda, ldb, ldc, sA, sB, sC, batch, vecA, vecB, vecC, bias, s
#ifdef MI_GEMM_TU_NAME
int mi::MI_GEMM_TU_NAME(MI_GEMM_TU_ARGS) { return pick_tile<MI_GEMM_TU_TA, MI_GEMM_TU_TB>(MI_GEMM_TU_PASS); }
#else
int mi::gemm_f32_tu_nn(MI_GEMM_TU_ARGS) { return pick_tile<false, false>(MI_GEMM_TU_PASS); }
int mi::gemm_f32_tu_nt(MI_GEMM_TU_ARGS) { return pick_tile<false, true>(MI_GEMM_TU_PASS); }
int mi::gemm_f32_tu_tn(MI_GEMM_TU_ARGS) { return pick_tile<true, false>(MI_GEMM_TU_PASS); }
int mi::gemm_f32_tu_tt(MI_GEMM_TU_ARGS) { return pick_tile<true, true>(MI_GEMM_TU_PASS); }
#endif
#endif  // MI_GEMM_TU_NAME || MI_GEMM_SINGLE_TU

#ifndef MI_GEMM_TU_NAME  // the entry points: this file on its own, or a probe's single unit
std::atomic<unsigned> mi::g_gemm_launch_counter{0};

namespace {

std::atomic<int> g_gemm_plan{MI_GEMM_PLAN_AUTO};

__global__ void zero_rows_kernel(float* C, int m, int n, long ldc, long strideC, const float* bias) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (long)m * n) C[blockIdx.y * strideC + (idx / n) * ldc + (idx % n)] = bias ? bias[idx % n] : 0.f;
}

}  // namespace

extern "C" int mi_gemm_bias_f32(int transa, int transb, int32_t m, int32_t n, int32_t k,
                                const float* A, int64_t lda, int64_t strideA, const float* B,
                                int64_t ldb, int64_t strideB, const float* bias, float* C,
                                int64_t ldc, int64_t strideC, int32_t batch, mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (m < 0 || n < 0 || k < 0 || batch < 0) return MI_EINVAL;
  if (m == 0 || n == 0 || batch == 0) return MI_OK;
  if (!C || ldc < n) return MI_EINVAL;
  if (k == 0) {
    if (batch > 65535) return MI_ERANGE;  // grid.y of the fill kernel
    const long total = (long)m * n;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((total + 255) / 256), (unsigned)batch),
                       dim3(256), 0, s, C, m, n, ldc, strideC, bias);
    return mi::check_launch();
  }
  if (!A || !B) return MI_EINVAL;
  if (lda < (transa ? m : k) || ldb < (transb ? k : n)) return MI_EINVAL;
  if (strideA < 0 || strideB < 0 || strideC < 0) return MI_EINVAL;
  // (the interior loaders address a tile with 32-bit byte offsets: leading dimensions below 2^21 elements)
  const bool vecA = (lda % 4 == 0) && (strideA % 4 == 0) && mi::aligned16(A) && lda < (1 << 21) && ldc < (1 << 24);
  const bool vecB = (ldb % 4 == 0) && (strideB % 4 == 0) && mi::aligned16(B) && ldb < (1 << 21);
  const bool vecC = (ldc % 4 == 0) && (strideC % 4 == 0) && mi::aligned16(C);
  // The persistent two-halves kernel (gemm_f32_duo.hip; same bits) only when pinned: with their k-loops free of
  // vector-ALU work the tile kernels below are level with it or ahead on every shape measured
  // (tools/probes/duo_probe.cpp; profiles/r03_gemm_duo_vs_tiles.log), so AUTO = TILES.
  const int plan = g_gemm_plan.load(std::memory_order_relaxed);
  if (plan == MI_GEMM_PLAN_DUO) {
    const int st = mi::launch_gemm_duo(transa, transb, m, n, k, A, lda, strideA, B, ldb, strideB, bias, C, ldc, strideC,
                                       batch, plan == MI_GEMM_PLAN_DUO, s);
    if (st != 1) return st;
    if (plan == MI_GEMM_PLAN_DUO) return MI_EINVAL;  // pinned, but the shape is not made of whole tiles
  }
#define MI_GEMM(F_) return mi::F_(A, B, C, m, n, k, lda, ldb, ldc, strideA, strideB, strideC, batch, vecA, vecB, vecC, bias, s)
  if (!transa && !transb) MI_GEMM(gemm_f32_tu_nn);
  if (!transa && transb) MI_GEMM(gemm_f32_tu_nt);
  if (transa && !transb) MI_GEMM(gemm_f32_tu_tn);
  MI_GEMM(gemm_f32_tu_tt);
#undef MI_GEMM
}

extern "C" int mi_gemm_set_plan(int plan) {
  if (plan != MI_GEMM_PLAN_AUTO && plan != MI_GEMM_PLAN_TILES && plan != MI_GEMM_PLAN_DUO) return MI_EINVAL;
  g_gemm_plan.store(plan, std::memory_order_relaxed);
  return MI_OK;
}

extern "C" int mi_gemm_f32(int transa, int transb, int32_t m, int32_t n, int32_t k, const float* A,
                           int64_t lda, int64_t strideA, const float* B, int64_t ldb,
                           int64_t strideB, float* C, int64_t ldc, int64_t strideC, int32_t batch,
                           mi_stream_t stream) {
  return mi_gemm_bias_f32(transa, transb, m, n, k, A, lda, strideA, B, ldb, strideB, nullptr, C, ldc,
                          strideC, batch, stream);
}
#endif  // !MI_GEMM_TU_NAME
