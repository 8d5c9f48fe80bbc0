// Interfaces between the CSR-transpose translation units (not installed): csr_transpose.hip (general and one-sweep plans, the
// C-ABI) and csr_transpose_items.hip (one workgroup per small item).
#ifndef MI_CSR_TRANSPOSE_INTERNAL_H_
#define MI_CSR_TRANSPOSE_INTERNAL_H_

#include "mi_common.h"

namespace mi {
// Does the one-workgroup-per-item LDS plan take this batch?  The number of waves per workgroup (16, 8 or 4), or 0.
int tr_item_lds_waves(int64_t nnz, int32_t batch, int32_t M, int32_t K);
int launch_tr_item_lds(int waves, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val, int32_t batch,
                       int32_t M, int32_t K, int32_t* t_rowptr, int32_t* t_col, float* t_val, hipStream_t s);
}  // namespace mi

#endif  // MI_CSR_TRANSPOSE_INTERNAL_H_
