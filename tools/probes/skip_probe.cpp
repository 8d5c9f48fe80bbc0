// Developer probe: the fused dense-skip kernel at BERT's probs.V shape (384 x 512x512 x 512x64, 10 % kept),
// with pieces compiled out (-DMI_SKIP_ABL=1/2/4: no B gathers / no list building and no walk / no walk; timing only).
// Measured on MI355X: full 0.190 ms; no B gathers 0.116; no walk 0.084; no list, no walk 0.080 — reading A costs
// 0.08 ms (5 TB/s), the walk's 10 M gathers of 256 B from L2 0.074 ms (35 TB/s: the L2 bandwidth), its LDS reads + FMAs 0.03.
#include "../../matrix-multiplication_amd/csrc/mi_status.hip"
#include "../../matrix-multiplication_amd/csrc/spmm_dense_skip.hip"
#include <cstdio>
#include <vector>
int main() {
  const int batch = 384, M = 512, K = 512, N = 64;
  std::vector<float> h((size_t)batch * M * K);
  unsigned x = 1;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) % 10 == 0) ? (x >> 8) * (1.0f / 16777216.0f) + 0.1f : 0.f; }
  float *A, *B, *C;
  (void)hipMalloc(&A, h.size() * 4); (void)hipMalloc(&B, (size_t)batch * K * N * 4); (void)hipMalloc(&C, (size_t)batch * M * N * 4);
  (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(B, 0, (size_t)batch * K * N * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 4; ++it) {
    (void)hipEventRecord(e0);
    int st = mi_spmm_dense_skip_f32(A, K, (long)M * K, batch, M, K, N, B, N, (long)K * N, nullptr, C, N, (long)M * N, nullptr);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("abl=%d status %d  %.3f ms\n", MI_SKIP_ABL, st, ms);
  }
  return 0;
}
