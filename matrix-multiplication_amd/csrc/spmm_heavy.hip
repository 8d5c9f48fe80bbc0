// The heavy rows of a row schedule (spmm_sched.hip): rows of hundreds to thousands of entries in a product that is over in a
// fraction of a millisecond (an adjacency matrix with power-law degrees: 1 % of the rows hold a sixth of the entries).
//
// A row's sum is ONE fmaf chain per output element in CSR order — that is the contract (include/mi_spmm.h; reference
// src/naive_sparse_mm.cu:60-92) and it is what makes every plan, every schedule and every GPU count give the same bits.
// Left to one wave, the chain advances at (gathers in flight) / (memory latency): 8 rows of B per ≈ 1.4 µs.  The chain
// itself is cheap; what a long row lacks is memory-level parallelism.  So here a whole 8-wave workgroup works on ONE row:
//   * the loader waves (8 − ⌈N / 64⌉ of them) gather the B rows of the next `E` entries (96 at N = 128) into one of two LDS
//     images — registers first, two chunks ahead, every load of a chunk in flight at once, then ds_write;
//   * the first ⌈N / 64⌉ waves (one lane per output element) walk the previous image in entry order: one LDS read + one FMA per
//     entry, the row's chain — while the other waves' loads of the next chunks are in the air.  One barrier per chunk.
// The row advances at about the CU's gather rate instead of one wave's, and the arithmetic is exactly the one-wave
// kernel's.  Measured (tools/probes/skew_trace.py, 170 K rows × 128, Pareto lengths clipped at 8 000: 93 heavy rows, the
// longest 7 806 entries): 138 µs alone on the chip = 1.7 µs per chunk of 96 entries (49 KB: 29 GB/s for that CU; the guide's
// figure for a CU gathering from the Infinity Cache is 33) — one wave per row takes ≈ 1 ms for the same row; beside the
// ordinary launch, whose waves share the CU's memory pipeline, 203 µs (raising the waves' priority with s_setprio changes
// nothing: the contention is not for issue slots).  First form of the round (every wave gathered AND the first waves chained;
// 191 µs): the phases of a step added up — chain 1.15, LDS stores 0.67, exposed loads 0.5 µs per 128 entries.  What was tried and
// dropped: a float4 per chain lane (N/4 lanes: 1.7 µs per chunk for the chain alone).  A branch around each load costs a factor
// of 1.6: hipcc then waits `vmcnt(0)` before every one of them (8 dependent trips to memory per chunk) — every slot is made
// valid instead.  New relative to the reference (one warp per row and 32 columns, whatever the row).
#include "spmm_device.h"
#include "spmm_internal.h"

namespace {

using mi::LongArg;

constexpr int kHeavyThreads = 512;
constexpr int kHeavyWaves = kHeavyThreads / 64;
constexpr int kHeavyFloats = 16384;  // floats per LDS image (64 KB); two images
constexpr int kHeavyMaxE = 128;      // entries per chunk at most
constexpr int kHeavyKQ = 8;          // float4 loads per loader thread and chunk: E · N/4 ≤ (loader threads) · 8

// Roles (round 6, second form): the first CW = ⌈N / 64⌉ waves only walk the chain (one output element per lane), the other
// 8 − CW waves only gather.  With every wave doing both, the phases of a step added up (chain 1.15 + LDS stores 0.67 + exposed
// loads 0.5 µs per 128 entries): the chain's waves also carried their share of the gather.  Now the chain of chunk i runs
// beside the loaders' issue of chunk i + 2 and their LDS stores of chunk i + 1; one barrier per chunk as before.
__global__ __launch_bounds__(kHeavyThreads) void spmm_heavy_rows_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ B, float* __restrict__ C, int N, long ldb, long ldc, const float* __restrict__ bias, int E,
    LongArg la) {
  // two images of kHeavyFloats floats, then the two chunks' values — addressed by integer offsets into the one array (a
  // pointer picked by the buffer index would become a table of generic pointers to LDS, which the back end refuses)
  extern __shared__ __attribute__((aligned(16))) float heavy_lds[];
  const int tid = threadIdx.x;
  const int slot = blockIdx.x;
  if (slot >= la.nslots) return;
  const int row = la.order[slot];
  const int start = rowptr[row], end = rowptr[row + 1];
  if (end - start > la.thresh) {  // left to spmm_long_rows_kernel
    if (tid == 0) long_list_append(la, row, end - start);
    return;
  }
  const int NQ = N >> 2;                               // float4 columns of a row
  const int CW = (N + 63) >> 6;                        // chain waves
  const int lt = tid - CW * 64;                        // loader thread index (< 0: a chain lane)
  const int LT = (kHeavyWaves - CW) * 64;              // loader threads
#define MI_HEAVY_CHUNK(P0_) (end - (P0_) < E ? (end - (P0_) > 0 ? end - (P0_) : 0) : E)

  if (lt < 0) {
    // ---- chain role: this lane's output element over the images in turn; CSR order, one fmaf per entry.  16 entries per
    // batch, two register sets: batch i + 1 is read from LDS while batch i's FMAs run.
    float acc1 = 0.f;
    const bool mine = tid < N;
#define MI_HEAVY_READ16(XX_, V4_, E_)                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) V4_[i] = *reinterpret_cast<const f32x4*>(&heavy_lds[vals + (E_) + 4 * i]); \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) XX_[i] = heavy_lds[src + ((E_) + i) * N];
#define MI_HEAVY_FMA16(XX_, V4_)                               \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {              \
    acc1 = __builtin_fmaf(V4_[i].x, XX_[4 * i], acc1);         \
    acc1 = __builtin_fmaf(V4_[i].y, XX_[4 * i + 1], acc1);     \
    acc1 = __builtin_fmaf(V4_[i].z, XX_[4 * i + 2], acc1);     \
    acc1 = __builtin_fmaf(V4_[i].w, XX_[4 * i + 3], acc1);     \
  }
    int p = start, b = 0;
    for (int count = MI_HEAVY_CHUNK(p); count > 0; count = MI_HEAVY_CHUNK(p)) {
      __syncthreads();  // image b is whole
      if (mine) {
        const int src = b * kHeavyFloats + tid, vals = 2 * kHeavyFloats + b * kHeavyMaxE;
        const int full = count & ~15;
        f32x4 va[4], vb[4];
        float ya[16], yb[16];
        int e = 0;
        if (full > 0) { MI_HEAVY_READ16(ya, va, 0) }
        for (; e + 32 <= full; e += 32) {
          MI_HEAVY_READ16(yb, vb, e + 16)
          MI_HEAVY_FMA16(ya, va)
          if (e + 32 < full) { MI_HEAVY_READ16(ya, va, e + 32) }
          MI_HEAVY_FMA16(yb, vb)
        }
        if (e < full) {
          MI_HEAVY_FMA16(ya, va)
          e += 16;
        }
        for (; e < count; ++e) acc1 = __builtin_fmaf(heavy_lds[vals + e], heavy_lds[src + e * N], acc1);
      }
      p += count;
      b ^= 1;
    }
#undef MI_HEAVY_READ16
#undef MI_HEAVY_FMA16
    if (mine) {
      if (bias) acc1 += bias[tid];
      __builtin_nontemporal_store(acc1, C + (long)row * ldc + tid);
    }
    return;
  }

  // ---- loader role: this thread's float4s of a chunk — quad q = lt + LT·k is entry q / NQ, float4 column q % NQ, the same for
  // every chunk.  EVERY slot is made valid (a quad beyond the chunk's E·NQ repeats the last one; an entry beyond a short last
  // chunk repeats its last entry): the loads and stores below are then straight-line code.  That matters — with a branch around
  // each load hipcc cannot count what is in flight and puts `s_waitcnt vmcnt(0)` in front of every one of them: eight
  // dependent trips to memory per chunk instead of one (measured: × 1.6).
  int qe[kHeavyKQ], qc[kHeavyKQ];
#pragma unroll
  for (int k = 0; k < kHeavyKQ; ++k) {
    int q = lt + LT * k;
    q = q < E * NQ ? q : E * NQ - 1;
    qe[k] = q / NQ;
    qc[k] = q - qe[k] * NQ;
  }
  // B rows travel two chunks ahead of the chain (register sets xa / xb, taking turns), col / val three chunks ahead
  f32x4 xa[kHeavyKQ], xb[kHeavyKQ];
  float xva[kHeavyKQ], xvb[kHeavyKQ];
  int cn[kHeavyKQ];
  float vn[kHeavyKQ];
#define MI_HEAVY_FETCH(P0_, COUNT_)                                   \
  _Pragma("unroll") for (int k = 0; k < kHeavyKQ; ++k) {              \
    const int i_ = (P0_) + (qe[k] < (COUNT_) ? qe[k] : (COUNT_) - 1); \
    cn[k] = col[i_];                                                  \
    vn[k] = val[i_];                                                  \
  }
#define MI_HEAVY_ISSUE(X_, XV_)                                                      \
  _Pragma("unroll") for (int k = 0; k < kHeavyKQ; ++k) {                             \
    X_[k] = *reinterpret_cast<const f32x4*>(B + (long)cn[k] * ldb + 4 * qc[k]);      \
    XV_[k] = vn[k];                                                                  \
  }
  // (rows of the image beyond a short chunk's count receive copies of its last entry: the chain never reads them)
#define MI_HEAVY_LAND(X_, XV_, IMG_, VALS_)                                          \
  _Pragma("unroll") for (int k = 0; k < kHeavyKQ; ++k) {                             \
    *reinterpret_cast<f32x4*>(&heavy_lds[(IMG_) + qe[k] * N + 4 * qc[k]]) = X_[k];   \
    if (qc[k] == 0) heavy_lds[(VALS_) + qe[k]] = XV_[k]; /* one lane per entry: 32 lanes storing to ONE address serialise */ \
  }
  // One step: the chunk at p lands from register set X_ in image I_; the barrier; the chunk two ahead takes the freed
  // registers, col / val of the chunk three ahead.  c1 / c2: entries of the chunks one / two ahead.
#define MI_HEAVY_STEP(X_, XV_, I_)                                                          \
  {                                                                                         \
    MI_HEAVY_LAND(X_, XV_, (I_) * kHeavyFloats, 2 * kHeavyFloats + (I_) * kHeavyMaxE)       \
    __syncthreads(); /* image I_ is whole; the chain is done with the other image */        \
    if (c2 > 0) {                                                                           \
      MI_HEAVY_ISSUE(X_, XV_)                                                               \
      const int p3 = p + count + c1 + c2, c3 = MI_HEAVY_CHUNK(p3);                          \
      if (c3 > 0) MI_HEAVY_FETCH(p3, c3)                                                    \
    }                                                                                       \
    p += count;                                                                             \
    count = c1;                                                                             \
    c1 = c2;                                                                                \
    c2 = MI_HEAVY_CHUNK(p + count + c1);                                                    \
  }
  int p = start;
  int count = MI_HEAVY_CHUNK(p);
  int c1 = MI_HEAVY_CHUNK(p + count);
  int c2 = MI_HEAVY_CHUNK(p + count + c1);
  if (count > 0) {
    MI_HEAVY_FETCH(p, count)
    MI_HEAVY_ISSUE(xa, xva)
    if (c1 > 0) {
      MI_HEAVY_FETCH(p + count, c1)
      MI_HEAVY_ISSUE(xb, xvb)
      if (c2 > 0) MI_HEAVY_FETCH(p + count + c1, c2)
    }
  }
  while (count > 0) {
    MI_HEAVY_STEP(xa, xva, 0)
    if (count <= 0) break;
    MI_HEAVY_STEP(xb, xvb, 1)
  }
#undef MI_HEAVY_FETCH
#undef MI_HEAVY_ISSUE
#undef MI_HEAVY_LAND
#undef MI_HEAVY_STEP
#undef MI_HEAVY_CHUNK
}

}  // namespace

namespace mi {

int launch_heavy_rows(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t N, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, LongArg la, hipStream_t s) {
  (void)M;
  if (la.order == nullptr || la.nslots <= 0) return MI_OK;
  if (N < 4 || N % 4 != 0 || N > 256) return MI_EINVAL;  // (chain waves ⌈N / 64⌉ ≤ 4; the dispatcher sends float4 shapes of ≤ 256 columns)
  // entries per chunk: an image holds them (E·N ≤ 16384 floats), the loader threads carry them in 8 float4s each, a multiple of
  // 16 (the chain's batches); N = 128: 96, N = 256: 32, N = 64: 128
  const int loaders = (kHeavyWaves - (N + 63) / 64) * 64;
  int E = kHeavyMaxE;
  if ((long)E * N > kHeavyFloats) E = kHeavyFloats / N;
  if ((long)E * (N / 4) > (long)loaders * kHeavyKQ) E = (int)((long)loaders * kHeavyKQ / (N / 4));
  E = E >= 16 ? E / 16 * 16 : (E >= 1 ? E : 1);
  const size_t lds = (size_t)(2 * kHeavyFloats + 2 * kHeavyMaxE) * sizeof(float);
  MI_HIP_TRY(hipFuncSetAttribute((const void*)spmm_heavy_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(spmm_heavy_rows_kernel, dim3((unsigned)la.nslots), dim3(kHeavyThreads), lds, s, rowptr, col, val, B, C, N,
                     (long)ldb, (long)ldc, bias, E, la);
  return check_launch();
}

}  // namespace mi
