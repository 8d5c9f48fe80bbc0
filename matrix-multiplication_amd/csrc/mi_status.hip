// Status codes, error bookkeeping, the host-side COO→CSR inspector step and the
// dummy kernel of the C-ABI (include/mi_spmm.h).
#include "mi_common.h"

namespace {
thread_local int g_last_hip_error = 0;

// Counterpart of dummyKernel (reference src/baseline_mm.cu:24-35), which
// printf()s the thread id from a 64×64 launch; this one records it instead.
__global__ void dummy_kernel(int* out) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  out[tid] = tid;
}
}  // namespace

namespace mi {
int record_hip_error(hipError_t e) {
  g_last_hip_error = static_cast<int>(e);
  return MI_EHIP;
}
}  // namespace mi

extern "C" {

int mi_spmm_abi_version(void) { return MI_SPMM_ABI_VERSION; }

const char* mi_status_string(int status) {
  switch (status) {
    case MI_OK: return "ok";
    case MI_EINVAL: return "invalid argument";
    case MI_ERANGE: return "size exceeds the 32-bit index range of this path";
    case MI_EHIP: return "HIP runtime error";
    case MI_ENOMEM: return "workspace too small";
    case MI_EUNSORTED: return "COO input is not sorted by row";
    default: return "unknown status";
  }
}

int mi_last_hip_error(void) { return g_last_hip_error; }

const char* mi_last_hip_error_string(void) {
  return hipGetErrorString(static_cast<hipError_t>(g_last_hip_error));
}

// Host inspector step.  Follows the contract of TiledSpMM_coo2csr (reference
// src/sparse_mm.cu:110-134): count entries per row, prefix-sum, then place
// entries in input order — which is only a valid CSR when the COO is sorted by
// row; unlike the reference this checks that and reports MI_EUNSORTED.
int mi_coo_to_csr_host(int32_t M, int64_t nnz, const int32_t* coo_row, const int32_t* coo_col,
                       const float* coo_val, int32_t* rowptr, int32_t* col_out, float* val_out) {
  if (M < 0 || nnz < 0 || !rowptr) return MI_EINVAL;
  if (nnz > 0x7fffffffLL) return MI_ERANGE;
  if (nnz > 0 && (!coo_row || !coo_col || !coo_val || !col_out || !val_out)) return MI_EINVAL;
  for (int32_t r = 0; r <= M; ++r) rowptr[r] = 0;
  int32_t prev = 0;
  for (int64_t i = 0; i < nnz; ++i) {
    const int32_t r = coo_row[i];
    if (r < 0 || r >= M) return MI_EINVAL;
    if (r < prev) return MI_EUNSORTED;
    prev = r;
    rowptr[r + 1]++;
  }
  for (int32_t r = 0; r < M; ++r) rowptr[r + 1] += rowptr[r];
  for (int64_t i = 0; i < nnz; ++i) {
    col_out[i] = coo_col[i];
    val_out[i] = coo_val[i];
  }
  return MI_OK;
}

int mi_dummy_kernel(int32_t* out, mi_stream_t stream) {
  if (!out) return MI_EINVAL;
  hipLaunchKernelGGL(dummy_kernel, dim3(64), dim3(64), 0, static_cast<hipStream_t>(stream), out);
  return mi::check_launch();
}

}  // extern "C"
