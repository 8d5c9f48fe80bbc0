// The heavy rows of a row schedule (spmm_sched.hip): rows of hundreds to thousands of entries in a product that is over in a
// fraction of a millisecond (an adjacency matrix with power-law degrees: 1 % of the rows hold a sixth of the entries).
//
// A row's sum is ONE fmaf chain per output element in CSR order — that is the contract (include/mi_spmm.h; reference
// src/naive_sparse_mm.cu:60-92) and it is what makes every plan, every schedule and every GPU count give the same bits.
// Left to one wave, the chain advances at (gathers in flight) / (memory latency): 8 rows of B per ≈ 1.4 µs.  The chain
// itself is cheap; what a long row lacks is memory-level parallelism.  So here a whole 8-wave workgroup works on ONE row:
//   * all 512 threads gather the B rows of the next `E` entries (E·N ≤ 16384 floats: 128 entries at N = 128) into one of
//     two LDS images — registers first, every load of the chunk in flight at once (64 KB per CU), then ds_write;
//   * N threads (one per output element) walk the previous image in entry order: one LDS read + one FMA per entry, the
//     row's chain — while the next chunk's loads are in the air.  One barrier per chunk.
// The row advances at about the CU's gather rate instead of one wave's, and the arithmetic is exactly the one-wave
// kernel's.  Measured (tools/probes/skew_trace.py, 170 K rows × 128, Pareto lengths clipped at 8 000: 93 heavy rows, the
// longest 7 806 entries = 61 chunks): 191 µs alone on the chip = 3.1 µs per chunk of 128 entries (64 KB: 21 GB/s per CU;
// the guide's figure for a CU gathering from the Infinity Cache is 33), of which the chain 1.15, the LDS stores 0.67, the
// loads' exposed part 0.5 — the phases of a step mostly add up because the chain's two waves also carry their share of the
// gather.  One wave per row takes ≈ 1 ms for the same row.  What was tried and dropped: a float4 per chain lane (N/4 lanes:
// 1.7 µs per chunk for the chain alone), a second register set of B rows two chunks ahead (no faster), LDS reads of batch
// i + 1 under the FMAs of batch i (spills at 256 VGPRs).  A branch around each load costs a factor of 1.6: hipcc then waits
// `vmcnt(0)` before every one of them (8 dependent trips to memory per chunk) — every slot is made valid instead.
// New relative to the reference (one warp per row and 32 columns, whatever the row).
#include "spmm_device.h"
#include "spmm_internal.h"

namespace {

using mi::LongArg;

constexpr int kHeavyThreads = 512;
constexpr int kHeavyFloats = 16384;  // floats per LDS image (64 KB); two images
constexpr int kHeavyMaxE = 128;      // entries per chunk at most
constexpr int kHeavyKQ = 8;          // float4 loads per thread and chunk: E · N/4 ≤ 512 · 8

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
  const int NQ = N >> 2;  // float4 columns of a row
  // this thread's float4s of a chunk: quad q = tid + 512·k is entry q / NQ, float4 column q % NQ — the same for every chunk.
  // EVERY slot is made valid (a quad beyond the chunk's E·NQ repeats the last one; an entry beyond a short last chunk
  // repeats its last entry): the loads and stores below are then straight-line code.  That matters — with a branch around
  // each load hipcc cannot count what is in flight and put `s_waitcnt vmcnt(0)` in front of every one of them: eight
  // dependent trips to memory per chunk instead of one (measured: 6.1 µs per chunk of 128 entries).
  int qe[kHeavyKQ], qc[kHeavyKQ];
#pragma unroll
  for (int k = 0; k < kHeavyKQ; ++k) {
    int q = tid + kHeavyThreads * k;
    q = q < E * NQ ? q : E * NQ - 1;
    qe[k] = q / NQ;
    qc[k] = q - qe[k] * NQ;
  }
  // B rows travel one chunk ahead of the chain, col / val two chunks ahead (a second register set of B rows, two chunks
  // ahead, was measured: no faster, and with the chain's two read sets it spills)
  f32x4 xa[kHeavyKQ];
  float xva[kHeavyKQ];
  int cn[kHeavyKQ];
  float vn[kHeavyKQ];
  float acc1 = 0.f;  // this lane's output element (lanes < N)

  // col / val of the chunk that starts at entry P0_ (COUNT_ ≥ 1 entries)
#define MI_HEAVY_FETCH(P0_, COUNT_)                                   \
  _Pragma("unroll") for (int k = 0; k < kHeavyKQ; ++k) {              \
    const int i_ = (P0_) + (qe[k] < (COUNT_) ? qe[k] : (COUNT_) - 1); \
    cn[k] = col[i_];                                                  \
    vn[k] = val[i_];                                                  \
  }
  // the B rows of the chunk whose col / val MI_HEAVY_FETCH brought, into register set X_: every load of the chunk in the air
  // before any is used
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
#define MI_HEAVY_CHUNK(P0_) (end - (P0_) < E ? (end - (P0_) > 0 ? end - (P0_) : 0) : E)
  // the row's chain over one image: CSR order, one fmaf per entry and output element.  ONE float per lane (N lanes: two waves
  // at N = 128) — a float4 per lane leaves the chain to N/4 lanes of one wave at ≈ 32 cycles per entry (measured: 1.7 µs per
  // 128 entries, half of the kernel); 16 entries per batch, their LDS reads all issued before the first FMA needs one
#define MI_HEAVY_CHAIN(IMG_, VALS_, COUNT_)                                                                        \
  if (tid < N) {                                                                                                   \
    const int src = (IMG_) + tid;                                                                                  \
    int e = 0;                                                                                                     \
    for (; e + 16 <= (COUNT_); e += 16) {                                                                          \
      f32x4 v4[4];                                                                                                 \
      float xx[16];                                                                                                \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) v4[i] = *reinterpret_cast<const f32x4*>(&heavy_lds[(VALS_) + e + 4 * i]); \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) xx[i] = heavy_lds[src + (e + i) * N];                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
        acc1 = __builtin_fmaf(v4[i].x, xx[4 * i], acc1);                                                           \
        acc1 = __builtin_fmaf(v4[i].y, xx[4 * i + 1], acc1);                                                       \
        acc1 = __builtin_fmaf(v4[i].z, xx[4 * i + 2], acc1);                                                       \
        acc1 = __builtin_fmaf(v4[i].w, xx[4 * i + 3], acc1);                                                       \
      }                                                                                                            \
    }                                                                                                              \
    for (; e < (COUNT_); ++e) acc1 = __builtin_fmaf(heavy_lds[(VALS_) + e], heavy_lds[src + e * N], acc1);         \
  }
  // One step: the chunk at p (`count` entries) lands in image I_, the barrier, the loads of the next chunk take the freed
  // registers, col / val of the chunk after it, then the chain over image I_.  c1: entries of the next chunk.
#define MI_HEAVY_STEP(I_)                                                                   \
  {                                                                                         \
    const int img = (I_) * kHeavyFloats, vals = 2 * kHeavyFloats + (I_) * kHeavyMaxE;      \
    MI_HEAVY_LAND(xa, xva, img, vals)                                                       \
    __syncthreads(); /* image I_ is whole; everybody is done with the other image */        \
    if (c1 > 0) {                                                                           \
      MI_HEAVY_ISSUE(xa, xva)                                                               \
      const int p2 = p + count + c1, c2 = MI_HEAVY_CHUNK(p2);                               \
      if (c2 > 0) MI_HEAVY_FETCH(p2, c2)                                                    \
    }                                                                                       \
    MI_HEAVY_CHAIN(img, vals, count)                                                        \
    p += count;                                                                             \
    count = c1;                                                                             \
    c1 = MI_HEAVY_CHUNK(p + count);                                                         \
  }

  int p = start;
  int count = MI_HEAVY_CHUNK(p);
  int c1 = MI_HEAVY_CHUNK(p + count);
  if (count > 0) {
    MI_HEAVY_FETCH(p, count)
    MI_HEAVY_ISSUE(xa, xva)
    if (c1 > 0) MI_HEAVY_FETCH(p + count, c1)
  }
  while (count > 0) {
    MI_HEAVY_STEP(0)
    if (count <= 0) break;
    MI_HEAVY_STEP(1)
  }
#undef MI_HEAVY_FETCH
#undef MI_HEAVY_ISSUE
#undef MI_HEAVY_LAND
#undef MI_HEAVY_CHUNK
#undef MI_HEAVY_CHAIN
#undef MI_HEAVY_STEP
  if (tid < N) {
    if (bias) acc1 += bias[tid];
    __builtin_nontemporal_store(acc1, C + (long)row * ldc + tid);
  }
}

}  // namespace

namespace mi {

int launch_heavy_rows(const int32_t* rowptr, const int32_t* col, const float* val, int32_t M, int32_t N, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, LongArg la, hipStream_t s) {
  (void)M;
  if (la.order == nullptr || la.nslots <= 0) return MI_OK;
  if (N < 4 || N % 4 != 0 || N > kHeavyThreads) return MI_EINVAL;  // (one chain lane per column; the dispatcher sends float4 shapes of ≤ 512 columns)
  int E = kHeavyMaxE;
  while (E > 1 && ((long)E * N > kHeavyFloats || (long)E * (N / 4) > (long)kHeavyThreads * kHeavyKQ)) E >>= 1;
  const size_t lds = (size_t)(2 * kHeavyFloats + 2 * kHeavyMaxE) * sizeof(float);
  MI_HIP_TRY(hipFuncSetAttribute((const void*)spmm_heavy_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(spmm_heavy_rows_kernel, dim3((unsigned)la.nslots), dim3(kHeavyThreads), lds, s, rowptr, col, val, B, C, N,
                     (long)ldb, (long)ldc, bias, E, la);
  return check_launch();
}

}  // namespace mi
