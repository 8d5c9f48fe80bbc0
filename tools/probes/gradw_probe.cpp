// Developer probe (not part of the product): the FC weight-gradient product dY^T.x (m = 3072, n = 768, k = 16384,
// A^T.B) with pieces compiled out (-DMI_GEMM_ABL: 1 no C stores, 2 no MFMAs, 4 no global operand loads; timing only).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMI_GEMM_ABL=2 -Iinclude -Imatrix-multiplication_amd/csrc tools/probes/gradw_probe.cpp -o tools/probes/gradw_probe_abl2
#include "../../matrix-multiplication_amd/csrc/mi_status.hip"
#define MI_GEMM_SINGLE_TU  // everything in this unit
#include "../../matrix-multiplication_amd/csrc/gemm_f32.hip"
#include <cstdio>
#include <vector>
int main() {
  const int m = 3072, n = 768, k = 16384;
  float *A, *B, *C;
  (void)hipMalloc(&A, (size_t)m * k * 4); (void)hipMalloc(&B, (size_t)n * k * 4); (void)hipMalloc(&C, (size_t)m * n * 4);
  {
    std::vector<float> h((size_t)m * k);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) * (1.0f / 16777216.0f); }
    (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, h.data(), (size_t)n * k * 4, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) {
    (void)hipEventRecord(e0);
    int st = mi_gemm_f32(1, 0, m, n, k, A, m, 0, B, n, 0, C, n, 0, 1, nullptr);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("gradw abl=%d  %.3f ms (status %d)\n", MI_GEMM_ABL, ms, st);
  }
  return 0;
}
