// Fused "sparsify on the fly" SpMM for gfx950:  C[b] = A[b] · B[b]  where A is given
// DENSE and its exact zeros are skipped.
//
// The reference's main call form is naiveSpMM.apply(dense_a, b): every call (and, for
// batched inputs, every slice) first runs `a.to_sparse_csr()`, moves the index arrays
// through the host, and only then launches spmm_kernel (reference matmuls.py:289-297,
// :178-187).  This kernel needs no CSR in memory:
//
//  * G = N/4 lanes own an output row (16 B each), so a wave works on R = 64/G rows at
//    once (BERT's N = 64: 4 rows per wave, every lane busy);
//  * each row group reads its row of A 4·G columns at a time (one 16-B load per lane),
//    finds the non-zeros with four wave ballots, ranks them in ascending column order
//    with mbcnt, and writes (column, value) pairs into a small per-wave LDS list — the
//    row's CSR slice, staged in LDS and never written to memory;
//  * the list is then walked with one group-uniform (broadcast) ds_read_b64 per
//    non-zero feeding the usual gather + fmaf chain, four non-zeros per round.
//
// Non-zeros are consumed in ascending column order = the CSR order of to_sparse_csr(),
// so the result is bit-identical to mi_dense_to_csr_* + mi_spmm_csr_batched_f32.
//
// B rows are gathered from global memory (they sit in L2 / Infinity Cache for the shapes this
// path serves).  Staging a whole item's B in LDS (BERT's V_bh = 512×64 = 128 KiB) was built and
// measured: it forces one 768-thread workgroup per CU and ran SLOWER than this form at both
// densities tried (B 32·12 × 512×512 × 512×64: 0.229 vs 0.181 ms at 10 % non-zeros, 0.869 vs
// 0.812 ms at 100 %) — occupancy, not the L2 round trip, is what this kernel needs.
#include <algorithm>

#include "mi_common.h"

#ifndef MI_SKIP_ABL
#define MI_SKIP_ABL 0  // developer probes only (timing, wrong results): 1 no B gathers in the walk, 2 no list building AND no walk
                       // (a walk over an unwritten list would gather from garbage addresses), 4 no walk
#endif

namespace {

using mi::f32x4;

__device__ __forceinline__ f32x4 fma4(float a, f32x4 x, f32x4 acc) {
  acc.x = __builtin_fmaf(a, x.x, acc.x);
  acc.y = __builtin_fmaf(a, x.y, acc.y);
  acc.z = __builtin_fmaf(a, x.z, acc.z);
  acc.w = __builtin_fmaf(a, x.w, acc.w);
  return acc;
}

__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

struct Pair {
  int k;
  float v;
};

// grid = 1-D over (item, row block), XCD-aware (below); block = WAVES*64 threads.
// G lanes per row (N ≤ 4·G), R = 64/G rows per wave, chunk = 4·G columns of A per row.
template <int G, bool VECA, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void spmm_dense_skip_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int K,
    int N, long lda, long ldb, long ldc, long strideA, long strideB, long strideC,
    int rows_per_block, const float* __restrict__ bias, int batch, unsigned row_blocks,
    const int* __restrict__ gate) {
  // gated form (mi_spmm_dense_skip_gated_f32): the launch is a no-op unless *gate != 0 (one scalar load)
  if (gate != nullptr && *gate == 0) return;
  // column tile of 256 output columns (gridDim.y > 1 only in the gated form)
  if (blockIdx.y != 0) {
    B += 256L * blockIdx.y;
    C += 256L * blockIdx.y;
    if (bias) bias += 256L * blockIdx.y;
    N = min(256, N - 256 * (int)blockIdx.y);
  } else if (gridDim.y > 1) {
    N = min(256, N);
  }
  constexpr int R = 64 / G;
  constexpr int CH = 4 * G;
  constexpr int CAP = 3 * CH;            // list entries per row group: several sparse chunks share one walk
  constexpr int LIST_STRIDE = CAP + 1;  // pairs; +1 keeps the R broadcast reads on different banks
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int gl = lane & (G - 1);
  const int g = lane / G;
  // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8, so give every XCD whole items
  // (item = xcd + 8·n): all row blocks of an item then share one L2 copy of that item's B
  // instead of pulling it into up to 8 L2s.  Placement affects speed only.
  const unsigned xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
  const long item = xcd + 8L * (idx / row_blocks);
  if (item >= batch) return;
  const unsigned rb = idx % row_blocks;
  const float* Ai = A + item * strideA;
  const float* Bi = B + item * strideB;
  float* Ci = C + item * strideC;

  Pair* list = reinterpret_cast<Pair*>(smem) + (wave * R + g) * LIST_STRIDE;

  const bool on = gl * 4 < N;
  // this lane's group as a lane mask
  const unsigned long long gm = G == 64 ? ~0ull : (((1ull << G) - 1ull) << (g * G));
  const int row_end = min(M, (int)(rb + 1) * rows_per_block);

  for (int row0 = rb * rows_per_block + wave * R; row0 < row_end; row0 += WAVES * R) {
    const int row = row0 + g;
    const bool valid = row < row_end;
    const float* arow = Ai + (long)(valid ? row : row0) * lda;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};

    // chunk loader: 4 columns per lane, zero beyond K or for a padding row
    auto load_chunk = [&](int k0) {
      const int kc = k0 + gl * 4;
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      if (valid && kc < K) {
        if (VECA && kc + 3 < K) {
          a = *reinterpret_cast<const f32x4*>(arow + kc);
        } else {
          a.x = arow[kc + 0];
          if (kc + 1 < K) a.y = arow[kc + 1];
          if (kc + 2 < K) a.z = arow[kc + 2];
          if (kc + 3 < K) a.w = arow[kc + 3];
        }
      }
      return a;
    };
    f32x4 a_next = load_chunk(0);
    int fill = 0;  // entries waiting in this group's list (group-uniform)
    for (int k0 = 0; k0 < K; k0 += CH) {
      const int kc = k0 + gl * 4;
      const f32x4 a = a_next;
      a_next = load_chunk(k0 + CH);  // in flight while this chunk is compacted
      // NaN counts as non-zero, -0.0 as zero (same test as to_sparse_csr / mi_dense_to_csr)
      const bool n0 = a.x != 0.0f, n1 = a.y != 0.0f, n2 = a.z != 0.0f, n3 = a.w != 0.0f;
      const unsigned long long m0 = __ballot(n0) & gm, m1 = __ballot(n1) & gm, m2 = __ballot(n2) & gm,
                               m3 = __ballot(n3) & gm;
      // rank in ascending column order: everything held by lower lanes of the group, then own
      int rank = fill + lanes_below(m0) + lanes_below(m1) + lanes_below(m2) + lanes_below(m3);
      if (!(MI_SKIP_ABL & 2)) {
      if (n0) list[rank++] = Pair{kc + 0, a.x};
      if (n1) list[rank++] = Pair{kc + 1, a.y};
      if (n2) list[rank++] = Pair{kc + 2, a.z};
      if (n3) list[rank] = Pair{kc + 3, a.w};
      }
      fill += __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) +
              __builtin_popcountll(m3);
      // walk the list once it could not take another chunk, or at the end of the row
      if (!__any(fill > CAP - CH) && k0 + CH < K) continue;
      // same-wave LDS traffic is executed in order; only the compiler must not reorder
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      for (int i = 0; __any(i < fill) && !(MI_SKIP_ABL & 6); i += 4) {
        Pair p[4];
        f32x4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = i + u < fill ? i + u : 0;  // entry 0 is always readable; unused when masked
          p[u] = list[idx];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool take = on && i + u < fill;
          x[u] = (take && !(MI_SKIP_ABL & 1)) ? *reinterpret_cast<const f32x4*>(Bi + (long)p[u].k * ldb + gl * 4)
                      : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (on && i + u < fill) acc = fma4(p[u].v, x[u], acc);
      }
      fill = 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    if (valid && on) {
      float* dst = Ci + (long)row * ldc + gl * 4;
      if (bias) acc += *reinterpret_cast<const f32x4*>(bias + gl * 4);
      __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(dst));
    }
  }
}

template <int G>
int launch_for_width(bool veca, const float* A, const float* B, float* C, int batch, int M, int K, int N, long lda,
                     long ldb, long ldc, long sA, long sB, long sC, const float* bias, hipStream_t s,
                     const int* gate = nullptr, unsigned col_tiles = 1) {
  constexpr int WAVES = 4;
  constexpr int R = 64 / G;
  constexpr size_t lds_bytes = (size_t)WAVES * R * (3 * 4 * G + 1) * sizeof(Pair);
  const int rows_per_block = WAVES * R;
  const long row_blocks = ((long)M + rows_per_block - 1) / rows_per_block;
  const long blocks = 8L * ((batch + 7) / 8) * row_blocks;  // every XCD gets the same count; extras exit
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  if (veca)
    hipLaunchKernelGGL((spmm_dense_skip_kernel<G, true, WAVES>), dim3((unsigned)blocks, col_tiles), dim3(WAVES * 64),
                       lds_bytes, s, A, B, C, M, K, N, lda, ldb, ldc, sA, sB, sC, rows_per_block, bias, batch,
                       (unsigned)row_blocks, gate);
  else
    hipLaunchKernelGGL((spmm_dense_skip_kernel<G, false, WAVES>), dim3((unsigned)blocks, col_tiles), dim3(WAVES * 64),
                       lds_bytes, s, A, B, C, M, K, N, lda, ldb, ldc, sA, sB, sC, rows_per_block, bias, batch,
                       (unsigned)row_blocks, gate);
  return mi::check_launch();
}

// flag |= 1 when x (rows × cols, leading dimension ld) holds an inf or a nan.  Grid-stride over 16-byte pieces when the
// rows are dense in memory (ld == cols), element-wise otherwise; one atomic per wave that saw one.
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ x, long rows, long cols, long ld,
                                                             int* __restrict__ flag) {
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nthreads = (long)gridDim.x * blockDim.x;
  bool bad = false;
  auto nonfinite = [](float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; };
  if (ld == cols && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
    const long n = rows * cols, n4 = n / 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (long i = tid; i < n4; i += nthreads) {
      const f32x4 v = x4[i];
      bad |= nonfinite(v.x) | nonfinite(v.y) | nonfinite(v.z) | nonfinite(v.w);
    }
    for (long i = n4 * 4 + tid; i < n; i += nthreads) bad |= nonfinite(x[i]);
  } else {
    const long n = rows * cols;
    for (long i = tid; i < n; i += nthreads) bad |= nonfinite(x[(i / cols) * ld + i % cols]);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace

extern "C" {

int mi_spmm_dense_skip_supported(int32_t N, int64_t lda, int64_t ldb, int64_t ldc, const float* A,
                                 const float* B, const float* C) {
  (void)lda;
  (void)A;
  return N > 0 && N <= 256 && N % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && mi::aligned16(B) && mi::aligned16(C);
}

int mi_spmm_dense_skip_f32(const float* A, int64_t lda, int64_t strideA, int32_t batch, int32_t M,
                           int32_t K, int32_t N, const float* B, int64_t ldb, int64_t strideB,
                           const float* bias, float* C, int64_t ldc, int64_t strideC,
                           mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (batch < 0 || M < 0 || K < 0 || N < 0 || strideA < 0 || strideB < 0 || strideC < 0) return MI_EINVAL;
  if (batch == 0 || M == 0 || N == 0) return MI_OK;
  if (!C || ldc < N) return MI_EINVAL;
  if (K > 0 && (!A || !B || lda < K || ldb < N)) return MI_EINVAL;
  if (!mi_spmm_dense_skip_supported(N, lda, ldb, ldc, A, B, C) || strideB % 4 != 0 || strideC % 4 != 0 ||
      (bias && !mi::aligned16(bias)))
    return MI_EINVAL;
  const bool veca = lda % 4 == 0 && strideA % 4 == 0 && mi::aligned16(A);
#define MI_SKIP(G_) \
  return launch_for_width<G_>(veca, A, B, C, batch, M, K, N, lda, ldb, ldc, strideA, strideB, strideC, bias, s)
  if (N <= 64) MI_SKIP(16);
  if (N <= 128) MI_SKIP(32);
  MI_SKIP(64);
#undef MI_SKIP
}

int mi_spmm_dense_skip_gated_f32(const float* A, int64_t lda, int64_t strideA, int32_t batch, int32_t M,
                                 int32_t K, int32_t N, const float* B, int64_t ldb, int64_t strideB,
                                 const float* bias, float* C, int64_t ldc, int64_t strideC,
                                 const int32_t* gate, mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (batch < 0 || M < 0 || K < 0 || N < 0 || strideA < 0 || strideB < 0 || strideC < 0) return MI_EINVAL;
  if (batch == 0 || M == 0 || N == 0) return MI_OK;
  if (!C || ldc < N) return MI_EINVAL;
  if (K > 0 && (!A || !B || lda < K || ldb < N)) return MI_EINVAL;
  if (N % 4 != 0 || ldb % 4 != 0 || ldc % 4 != 0 || !mi::aligned16(B) || !mi::aligned16(C) || strideB % 4 != 0 ||
      strideC % 4 != 0 || (bias && !mi::aligned16(bias)))
    return MI_EINVAL;
  const unsigned tiles = (unsigned)((N + 255) / 256);
  if (tiles > 65535u) return MI_ERANGE;
  const bool veca = lda % 4 == 0 && strideA % 4 == 0 && mi::aligned16(A);
#define MI_SKIP(G_)                                                                                                 \
  return launch_for_width<G_>(veca, A, B, C, batch, M, K, N, lda, ldb, ldc, strideA, strideB, strideC, bias, s, gate, \
                              tiles)
  if (N <= 64) MI_SKIP(16);
  if (N <= 128) MI_SKIP(32);
  MI_SKIP(64);
#undef MI_SKIP
}

int mi_nonfinite_flag_f32(const float* x, int64_t rows, int64_t cols, int64_t ld, int32_t* flag, mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!flag || rows < 0 || cols < 0 || (rows > 0 && cols > 0 && (!x || ld < cols))) return MI_EINVAL;
  if (hipMemsetAsync(flag, 0, sizeof(int32_t), s) != hipSuccess) return mi::check_launch();
  const long n = rows * cols;
  if (n == 0) return MI_OK;
  const long want = (n / 4 + 255) / 256;
  const unsigned blocks = (unsigned)std::min<long>(std::max<long>(want, 1), 2048);
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(blocks), dim3(256), 0, s, x, (long)rows, (long)cols, (long)ld, flag);
  return mi::check_launch();
}

}  // extern "C"
