// Rows beyond the long-row threshold ("skewed matrices"): listed by the main kernels on the way (spmm_device.h:
// long_list_append) or by find_long_rows_kernel, summed in a fixed order that the oracle restates (oracle_spmm_csr_long_f32) by
// the staged kernel of spmm_heavy.hip.  This file: the rule, the list, its workspace.  Contract: include/mi_spmm.h
// (mi_spmm_csr_ws_f32, MI_LONG_ROWS_*).
// New relative to the reference, whose kernel walks every row with one warp (src/naive_sparse_mm.cu:60-92).
#include "spmm_device.h"
#include "spmm_internal.h"

namespace {

using mi::LongArg;

// ---------------------------------------------------------------------------
// Skewed matrices.  A row is owned by one wave, which sustains only a few GB/s of gathers, so
// a row with 10⁵–10⁶ non-zeros would be a serial tail of tens of milliseconds.  When the caller
// supplies a workspace (custom_mm always does), rows with more than kLongRow non-zeros are
// skipped AND listed by the kernels above (`la`: long_list_append; find_long_rows_kernel builds the same
// list for plans whose kernel lives elsewhere, and for prepared lists), and summed
// by S = clamp(len / 32768, 1, 128) groups of 16 chains: the row's 1024-non-zero chunks are
// dealt round-robin to 16·S chains (chain q runs the fmaf chain over chunks q, q+16S, q+32S, …
// in increasing position), group g owns chains 16g … 16g+15 and adds them in that order,
// and the S workgroup sums are added in order g = 0 … S-1 — by the same workgroup when S = 1,
// else through the workspace by whichever of the S workgroups delivers its partial row LAST
// (an arrival counter per row; agent-scope release by every deliverer, acquire by the last: the sum
// itself is always taken in the order g = 0 … S-1, so it does not depend on who arrives when).
// That is a different — fixed, launch-independent, a function of the row length only — summation
// order for those rows; oracle_spmm_csr_long_f32 restates it, so results stay bit-identical to the oracle.
// ---------------------------------------------------------------------------

__global__ void find_long_rows_kernel(const int* __restrict__ rowptr, int M, LongArg la) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  const int len = rowptr[r + 1] - rowptr[r];
  if (len > kLongRow) long_list_append(la, (int)r, len);
}

}  // namespace

namespace mi {

LongWs long_ws_layout(int64_t nnz, int32_t N) {
  LongWs w;
  w.cap_e = nnz / kLongRow + 1;
  w.cap_p = nnz >> kLongSplitShift;
  w.cap_s = w.cap_e + w.cap_p;
  w.owner_off = 4 + (size_t)kLongEnt * (size_t)w.cap_e;
  const size_t ints = w.owner_off + (size_t)w.cap_s;
  w.partial_off = (ints * sizeof(int) + 15) / 16 * 16;
  w.adapt_off = (w.partial_off + (size_t)w.cap_p * (size_t)N * sizeof(float) + 15) / 16 * 16;
  w.bytes = w.adapt_off + kAdaptSlots * sizeof(int);
  return w;
}

int launch_find_long_rows(const int32_t* rowptr, int32_t M, const LongArg& la, hipStream_t s) {
  if (M <= 0) return MI_OK;
  hipLaunchKernelGGL(find_long_rows_kernel, dim3((unsigned)(((long)M + 255) / 256)), dim3(256), 0, s, rowptr, M, la);
  return check_launch();
}

int launch_long_rows(int* ws, const LongWs& lw, const int32_t* rowptr, const int32_t* col, const float* val, const float* B,
                     float* C, int32_t N, int64_t ldb, int64_t ldc, const float* bias, bool reset, hipStream_t s) {
  // a workgroup per group of chains and 64 columns, the chains staged through LDS — or, with many rows, one wave per chain
  // (spmm_heavy.hip): the sums of the comment above, every width ≥ 4 (narrower products are never split)
  return launch_long_rows_staged(ws, lw, rowptr, col, val, B, C, N, ldb, ldc, bias, reset, s);
}

}  // namespace mi

extern "C" {

size_t mi_spmm_csr_workspace_bytes(int64_t nnz, int32_t N) {
  return nnz > 0 && N > 0 ? mi::long_ws_layout(nnz, N).bytes : 0;
}

int mi_spmm_long_rows_prepare(const int32_t* rowptr, int32_t M, int64_t nnz, int32_t N, void* workspace,
                              size_t workspace_bytes, mi_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M < 0 || nnz < 0 || N < 0) return MI_EINVAL;
  if (nnz > 0x7fffffffLL) return MI_ERANGE;
  if (nnz <= kLongRow || M == 0 || N == 0) return MI_OK;  // no row can be long: the list is never read
  if (!rowptr || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return MI_EINVAL;
  const mi::LongWs lw = mi::long_ws_layout(nnz, N);
  if (workspace_bytes < lw.bytes) return MI_ENOMEM;
  int* ws = static_cast<int*>(workspace);
  MI_HIP_TRY(hipMemsetAsync(ws, 0, 16, s));
  const mi::LongArg la = {kLongRow, (int)lw.cap_e, (int)lw.cap_s, (int)lw.cap_p, ws, nullptr, nullptr, 0};
  hipLaunchKernelGGL(find_long_rows_kernel, dim3((unsigned)(((long)M + 255) / 256)), dim3(256), 0, s, rowptr, M, la);
  return mi::check_launch();
}

int mi_spmm_long_row_threshold(void) { return kLongRow; }

}  // extern "C"
