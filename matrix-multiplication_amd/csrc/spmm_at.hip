// Transpose-free Aᵀ·X for batched CSR A whose X rows fit LDS — the gradient of the dense operand in pruned attention:
//   dV[i] = P[i]ᵀ · dC[i],   P[i] a CSR item [M × K] (rows = queries, columns = keys), dC[i] [M × N], dV[i] [K × N], N ≤ 64.
// Replaces, for these shapes, csr_transpose_batched + the permuted batched product (matmuls._batched_csr_backward): with
// attention probabilities the pattern is new on every step, so the transpose, its permutation and the shifted index
// arrays were rebuilt per step (profiles/r05_attention_csr_fresh_before.log: forward + backward 0.30 ms with a frozen
// pattern, 0.68 ms with a fresh one, against 0.35 ms for the dense classes).  Reference: the per-call conversion of
// matmuls.py:289-297 and the backward at :245-256.
//
// Design (no counterpart in the reference): the output tile lives in REGISTERS.  A workgroup of 16 waves owns 16·CW
// consecutive columns of one item (= rows of dV), wave w the CW columns [k0, k0 + CW) as CW accumulators per lane, lane =
// output column (n).  Every wave walks ALL rows of the item in ascending order; a row's column indices arrive as one
// coalesced load per 64 entries, a ballot picks the entries that fall in the wave's columns, and each picked entry is one
// fmaf into accumulator (col − k0) — a wave-uniform register index — with the row of dC read from LDS, where the workgroup
// staged the item's dC once (M × N × 4 ≤ 128 KiB per tile of rows; taller items go tile by tile).
// Order: for an output element (k, n) the terms arrive by ascending row q, entries of one row in CSR order — exactly the
// chain of the transposed product (a stable transpose lists column k's entries by ascending row), so the result is
// bit-identical to csr_transpose + the row-split product, and to oracle.csr_transpose + oracle.spmm_csr.
// Cost model: a wave pays ≈8 instructions per row it scans and ≈10 per entry it owns; the scan is redundant 16× per
// workgroup (every wave reads every row's indices — L1 hits after the first wave), about a third of the work at 10 % kept.
#include "mi_common.h"
#include "mi_lanes.h"

namespace {

using mi::f32x4;

constexpr int kAtWaves = 16;
constexpr int kAtLdsRowFloats = 64;           // a staged row of dC: N ≤ 64 floats at stride 64
constexpr int kAtTileRows = 512;              // 512 × 64 × 4 B = 128 KiB of the CU's 160 KiB

template <int CW>
__global__ __launch_bounds__(kAtWaves * 64) void spmm_at_kernel(
    const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ X, long ldx, long strideX, float* __restrict__ Y, long ldy, long strideY, int M, int K,
    int N, int units_per_item, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [tile rows][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long item = blockIdx.x / (unsigned)units_per_item;
  const int cb = (int)(blockIdx.x % (unsigned)units_per_item);
  const int k0 = (cb * kAtWaves + wave) * CW;  // this wave's first column of A = first row of Y
  const int* rp = rowptr + item * ((long)M + 1);
  const float* Xi = X + item * strideX;
  float acc[CW];
#pragma unroll
  for (int j = 0; j < CW; ++j) acc[j] = 0.f;

  // one chunk of ≤ 64 entries of a row: the entries in [k0, k0 + CW) in CSR order, each one fmaf into its accumulator
  auto take = [&](int c, float v, float x) {
    const unsigned rel = (unsigned)(c - k0);  // absent lanes carry c = −1 − k0 … never < CW
    unsigned long long mask = __ballot(rel < (unsigned)CW);
    while (mask) {
      const int i = __builtin_ctzll(mask);
      mask &= mask - 1;
      const int kk = __builtin_amdgcn_readlane((int)rel, i);  // wave-uniform register index
      const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
      acc[kk] = __builtin_fmaf(a, x, acc[kk]);
    }
  };

  for (int t0 = 0; t0 < M; t0 += kAtTileRows) {
    const int t1 = t0 + kAtTileRows < M ? t0 + kAtTileRows : M;
    if (t0 > 0) __syncthreads();  // every wave is done with the previous tile's rows
    if (vec_ok) {                 // N % 4 == 0, 16-byte aligned rows: a float4 per thread and pass
      const int nv = N >> 2;      // float4 per row (≤ 16)
      for (int idx = tid; idx < (t1 - t0) * nv; idx += kAtWaves * 64) {
        const int r = idx / nv, c4 = idx - r * nv;
        *reinterpret_cast<f32x4*>(xs + r * kAtLdsRowFloats + 4 * c4) =
            *reinterpret_cast<const f32x4*>(Xi + (long)(t0 + r) * ldx + 4 * c4);
      }
    } else {
      for (int idx = tid; idx < (t1 - t0) * N; idx += kAtWaves * 64) {
        const int r = idx / N, c = idx - r * N;
        xs[r * kAtLdsRowFloats + c] = Xi[(long)(t0 + r) * ldx + c];
      }
    }
    __syncthreads();
    if (k0 >= K) continue;  // (a wave beyond the last column still joins the barriers above)
    for (int q = t0; q < t1; q += 4) {  // four rows per step: their first chunks travel together
      int b[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) b[u] = rp[q + u < t1 ? q + u : t1];  // wave-uniform: scalar loads
      int c[4];
      float v[4], x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = b[u] + lane;
        const bool has = idx < b[u + 1];
        c[u] = has ? col[idx] : -1;
        v[u] = has ? val[idx] : 0.f;
        x[u] = xs[(q + u - t0 < t1 - t0 ? q + u - t0 : 0) * kAtLdsRowFloats + lane];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        take(c[u], v[u], x[u]);
        for (int p = b[u] + 64; p < b[u + 1]; p += 64) {  // rows beyond 64 entries: chunk by chunk
          const int idx = p + lane;
          const bool has = idx < b[u + 1];
          take(has ? col[idx] : -1, has ? val[idx] : 0.f, x[u]);
        }
      }
    }
  }
  if (k0 >= K || lane >= N) return;
  float* Yw = Y + item * strideY + (long)k0 * ldy + lane;
#pragma unroll
  for (int j = 0; j < CW; ++j)
    if (k0 + j < K) __builtin_nontemporal_store(acc[j], Yw + (long)j * ldy);
}

// torch's batched CSR indices (int64: crow [batch, M + 1] counting from 0 in every item, col [batch · per_item]) → what the
// batched kernels read (int32: offsets with the item's base per_item · i added — the "rowptr of rowptrs" — and columns), in
// ONE launch: thread t narrows entries 2t, 2t + 1 of the concatenation [crow | col] (16 bytes in, 8 out).
__global__ __launch_bounds__(256) void batched_csr_narrow_kernel(const long* __restrict__ crow, const long* __restrict__ col,
                                                                 int* __restrict__ off32, int* __restrict__ col32, long n_off,
                                                                 long n_col, int rows1, long per_item) {
  const long t = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long i = t + u;
    if (i < n_off) {
      off32[i] = (int)(crow[i] + (i / rows1) * per_item);
    } else if (i - n_off < n_col) {
      col32[i - n_off] = (int)col[i - n_off];
    }
  }
}

}  // namespace

extern "C" {

int mi_batched_csr_narrow_i64(const int64_t* crow, const int64_t* col, int32_t batch, int32_t M, int64_t per_item,
                              int32_t* offsets, int32_t* columns, mi_stream_t stream) {
  if (batch < 0 || M < 0 || per_item < 0) return MI_EINVAL;
  if ((int64_t)batch * per_item > 0x7fffffffLL) return MI_ERANGE;
  if (batch == 0) return MI_OK;
  if (!crow || !offsets || (per_item > 0 && (!col || !columns))) return MI_EINVAL;
  const long n_off = (long)batch * ((long)M + 1), n_col = (long)batch * per_item;
  const long blocks = ((n_off + n_col + 1) / 2 + 255) / 256;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  hipLaunchKernelGGL(batched_csr_narrow_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const long*>(crow), reinterpret_cast<const long*>(col), offsets, columns, n_off, n_col,
                     M + 1, (long)per_item);
  return mi::check_launch();
}

// Y[i] (K × N) = A[i]ᵀ · X[i] for batched CSR A (rowptr [batch, M + 1] with the items' base offsets, as
// mi_spmm_csr_batched_f32 takes it), X [batch, M, N], N ≤ 64.  Returns MI_OK after launching, 1 (nothing launched)
// when the shape is not covered — the caller then transposes (mi_csr_transpose_batched_f32) and multiplies.
int mi_spmm_csr_batched_at_f32(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz_total,
                               int32_t batch, int32_t M, int32_t K, int32_t N, const float* X, int64_t ldx,
                               int64_t strideX, float* Y, int64_t ldy, int64_t strideY, mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M < 0 || K < 0 || N < 0 || nnz_total < 0 || batch < 0 || strideX < 0 || strideY < 0) return MI_EINVAL;
  if (nnz_total > 0x7fffffffLL) return MI_ERANGE;
  if (K == 0 || N == 0 || batch == 0) return MI_OK;
  if (!rowptr || !Y) return MI_EINVAL;
  if (nnz_total > 0 && (!col || !val || !X)) return MI_EINVAL;
  if (ldx < N || ldy < N) return MI_EINVAL;
  if (N > 64 || M == 0) return 1;  // wider outputs / nothing to sum: the transposed product's kernels (they also write the zeros)
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 256;
    }
    return n;
  }();
  // columns per wave: 32 or 16 — whichever costs the busiest CU less (a wave pays ≈8 instructions per row it scans and
  // ≈10 per entry it owns; a CU runs one 16-wave workgroup at a time: the staged rows take 128 KiB of its LDS)
  auto cost = [&](int cw) {
    const long upi = ((long)K + kAtWaves * cw - 1) / (kAtWaves * cw);
    const long rounds = (upi * batch + cus - 1) / cus;
    const double own = (double)nnz_total / (double)batch * (double)cw / (double)K;
    return (double)rounds * (8.0 * M + 10.0 * own);
  };
  const int cw = cost(32) <= cost(16) ? 32 : 16;
  const long upi = ((long)K + kAtWaves * cw - 1) / (kAtWaves * cw);
  const long blocks = upi * batch;
  if (blocks > 0x7fffffffL) return MI_ERANGE;
  const int tile = M < kAtTileRows ? M : kAtTileRows;
  const size_t lds = (size_t)tile * kAtLdsRowFloats * sizeof(float);
  const int vec_ok = (N % 4 == 0) && (ldx % 4 == 0) && (strideX % 4 == 0) && mi::aligned16(X);
#define MI_AT(CW_)                                                                                                       \
  do {                                                                                                                   \
    auto k = spmm_at_kernel<CW_>;                                                                                        \
    if (lds > 64 * 1024) MI_HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kAtWaves * 64), lds, s, rowptr, col, val, X, (long)ldx, (long)strideX, Y, \
                       (long)ldy, (long)strideY, M, K, N, (int)upi, vec_ok);                                            \
  } while (0)
  if (cw == 32) MI_AT(32);
  else MI_AT(16);
#undef MI_AT
  return mi::check_launch();
}

}  // extern "C"
