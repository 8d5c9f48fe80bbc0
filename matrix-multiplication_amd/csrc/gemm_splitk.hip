// Deterministic split-k for dense fp32 products with FEW output tiles and a LONG k (weight gradients Aᵀ·B over thousands of
// tokens, 256 × 256 × 65536): one k-ordered chain per output element leaves most of the chip idle — 768 × 768 outputs are 36 tiles
// of 128 × 128 on 256 CUs (1.16 – 2.14 × rocBLAS at k = 8 – 64 K, profiles/r03_gemm_shapes.log).  The reference's criterion for
// this product is torch.allclose at 1e-5 (tests/cublas_kernel_test.py:27-28) and its cuBLAS order is unspecified
// (src/baseline_mm.cu:96-101), so the order may be chosen — but it must be FIXED: here k is cut into S equal ranges, S a
// function of the SHAPE alone (mi_gemm_split_count), every range is the usual k-ordered fmaf chain started from zero (the tile
// kernels of gemm_f32.hip, launched as a batch of S items whose operands are the k-ranges: no new product kernel), and the S
// partial sums of an element are added in index order, ((p0 + p1) + p2) + …, then the bias.  The oracle restates exactly that
// (oracle_gemm_f32), so parity stays bit for bit; results also stay within the reference tests' 1e-5 of torch.
// Same pattern as the long-row rule of the sparse product (spmm_long.hip): a fixed split, restated in the oracle.
#include "mi_common.h"

namespace {

// C[i, j] = ((P[0][i, j] + P[1][i, j]) + …) + bias[j]; P [S][m][n] contiguous.  Four elements per thread where n allows.
__global__ __launch_bounds__(256) void gemm_splitk_combine_kernel(const float* __restrict__ P, int S, long mn, int n, float* __restrict__ C,
                                                                  long ldc, const float* __restrict__ bias) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= mn) return;
  float t = P[idx];
  for (int s = 1; s < S; ++s) t = __fadd_rn(t, P[(long)s * mn + idx]);
  const long i = idx / n, j = idx - i * n;
  if (bias) t = __fadd_rn(t, bias[j]);
  C[i * ldc + j] = t;
}

}  // namespace

extern "C" {

// How many ranges k is cut into — a function of the shape alone.  1: the plain chain.
//   * one product (batch == 1) with k ≥ 4096 whose 128 × 128 output tiles number fewer than 128 (half the CUs);
//   * S = the largest power of two ≤ min(320 / tiles, k / 1024), halved until k is a multiple of 32·S (whole k-tiles per range);
//   * at least 2.
int mi_gemm_split_count(int32_t m, int32_t n, int32_t k, int32_t batch) {
  if (batch != 1 || k < 4096 || m <= 0 || n <= 0) return 1;
  const long tiles = (((long)m + 127) / 128) * (((long)n + 127) / 128);
  if (tiles >= 128) return 1;
  long cap = 320 / tiles;
  if (cap > k / 1024) cap = k / 1024;
  int S = 1;
  while (2L * S <= cap) S *= 2;
  while (S > 1 && k % (32 * S) != 0) S /= 2;
  return S;
}

size_t mi_gemm_workspace_bytes(int32_t m, int32_t n, int32_t k, int32_t batch) {
  const int S = mi_gemm_split_count(m, n, k, batch);
  return S > 1 ? (size_t)S * (size_t)m * (size_t)n * sizeof(float) : 0;
}

int mi_gemm_ws_f32(int transa, int transb, int32_t m, int32_t n, int32_t k, const float* A, int64_t lda, int64_t strideA,
                   const float* B, int64_t ldb, int64_t strideB, const float* bias, float* C, int64_t ldc, int64_t strideC,
                   int32_t batch, void* workspace, size_t workspace_bytes, mi_stream_t stream) {
  const int S = (m >= 0 && n >= 0 && k >= 0 && batch >= 0) ? mi_gemm_split_count(m, n, k, batch) : 1;
  if (S <= 1) return mi_gemm_bias_f32(transa, transb, m, n, k, A, lda, strideA, B, ldb, strideB, bias, C, ldc, strideC, batch, stream);
  if (!workspace || !mi::aligned16(workspace)) return MI_EINVAL;
  if (workspace_bytes < mi_gemm_workspace_bytes(m, n, k, batch)) return MI_ENOMEM;  // (never silently unsplit: the order is part of the result)
  if (!C || ldc < n) return MI_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* P = static_cast<float*>(workspace);
  const int32_t ks = k / S;
  // the S ranges as a batch: item s reads rows / columns [s·ks, (s+1)·ks) of the k dimension of both operands
  const int64_t stepA = (int64_t)ks * (transa ? lda : 1), stepB = (int64_t)ks * (transb ? 1 : ldb);
  int st = mi_gemm_bias_f32(transa, transb, m, n, ks, A, lda, stepA, B, ldb, stepB, nullptr, P, n, (int64_t)m * n, S, stream);
  if (st != MI_OK) return st;
  const long mn = (long)m * n;
  hipLaunchKernelGGL(gemm_splitk_combine_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, s, P, S, mn, n, C, (long)ldc, bias);
  return mi::check_launch();
}

}  // extern "C"
