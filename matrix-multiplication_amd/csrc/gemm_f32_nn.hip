// The dense fp32 kernels of gemm_f32.hip for A·B as a translation unit of their own (four parallel compiles
// instead of one long one; see "Translation units" in gemm_f32.hip).
#define MI_GEMM_TU_NAME gemm_f32_tu_nn
#define MI_GEMM_TU_TA false
#define MI_GEMM_TU_TB false
#include "gemm_f32.hip"
