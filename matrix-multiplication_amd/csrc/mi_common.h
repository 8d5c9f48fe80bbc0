// Shared host-side helpers for the C-ABI translation units (not installed).
#ifndef MI_COMMON_H_
#define MI_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi_spmm.h"

namespace mi {

// Records a failing hipError_t for mi_last_hip_error() and maps it to MI_EHIP.
int record_hip_error(hipError_t e);

// Call after every kernel launch: picks up launch-configuration errors without
// synchronising (the reference checks nothing after <<<>>>).
inline int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MI_OK : record_hip_error(e);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int pow2_ceil(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// spmm_slab.hip — the LDS-slab kernel for moderate density (caller checks the vec4 requirements).
int launch_spmm_slab(const int32_t* rowptr, const int32_t* col, const float* val, const float* B, float* C,
                     int32_t M, int32_t K, int32_t N, int64_t ldb, int64_t ldc, const float* bias,
                     int long_thresh, hipStream_t s);

// spmm_ldsb.hip — an item's whole B resident in LDS (batched products of small matrices); the caller checks the
// vec4 requirements and spmm_ldsb_fits(); rows beyond long_thresh non-zeros are skipped (the caller lists them)
bool spmm_ldsb_fits(int32_t K, int32_t N);
int spmm_ldsb_tiles(int32_t K, int32_t N);  // column tiles the plan cuts N into (0: does not fit)
// perm (may be null): the value of entry p is val[perm[p]] (the transposed pattern of a batched CSR with its
// permutation: no gathered copy of the values)
int launch_spmm_ldsb(const int32_t* rowptr, const int32_t* col, const float* val, const float* B, float* C,
                     int32_t batch, int32_t M, int32_t K, int32_t N, int64_t ldb, int64_t ldc, int64_t strideB,
                     int64_t strideC, const float* bias, int long_thresh, hipStream_t s, const int32_t* perm = nullptr,
                     int64_t nnz_total = -1);  // nnz_total (≥ 4): lets the quad form clamp its 16-byte loads of col / val;
                                               // returns 1 (nothing launched) when only that form covers the shape and it cannot run

// the same kernel on column-major operands (X = Bᵀ [N, ldx], Y = Cᵀ [N, ldy]); caller checks the requirements
int launch_spmm_slab_colmajor(const int32_t* rowptr, const int32_t* col, const float* val, const float* X, float* Y,
                              int32_t M, int32_t K, int32_t N, int64_t ldx, int64_t ldy, hipStream_t s);

// spmm_csr.hip — one wave per row on row-major B with COLUMN-major output (the executor's output transpose
// fused into the epilogue).  MI_OK = launched (or, with launch = false, "would launch"); 1 = AUTO would not
// run the one-wave-per-row kernel here; negative = error.
int launch_spmm_wave_row_colmajor_out(const int32_t* rowptr, const int32_t* col, const float* val, int64_t nnz,
                                      int32_t M, int32_t K, int32_t N, const float* B, int64_t ldb, float* Ccm,
                                      int64_t ldc, bool launch, hipStream_t s);

// gemm_f32_duo.hip — the persistent two-halves-in-anti-phase form of the dense fp32 product.  MI_OK = launched;
// 1 = the shape is not made of whole tiles (or, unless `force`, of too few of them): the caller takes gemm_f32.hip's
// tile kernels; negative = error.  Same bits either way.
int launch_gemm_duo(int transa, int transb, int32_t m, int32_t n, int32_t k, const float* A, int64_t lda,
                    int64_t strideA, const float* B, int64_t ldb, int64_t strideB, const float* bias, float* C,
                    int64_t ldc, int64_t strideC, int32_t batch, bool force, hipStream_t s);

}  // namespace mi

#define MI_HIP_TRY(expr)                                  \
  do {                                                    \
    hipError_t _e = (expr);                               \
    if (_e != hipSuccess) return mi::record_hip_error(_e); \
  } while (0)

#endif  // MI_COMMON_H_
