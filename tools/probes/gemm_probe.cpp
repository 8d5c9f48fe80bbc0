// Developer probe (not part of the product): per-phase cycles of gemm_f32_kernel at BERT's q.kT shape.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMI_GEMM_TIMING -Iinclude -Imatrix-multiplication_amd/csrc tools/probes/gemm_probe.cpp -o tools/probes/gemm_probe
#include "../../matrix-multiplication_amd/csrc/mi_status.hip"
#define MI_GEMM_SINGLE_TU  // everything in this unit
#include "../../matrix-multiplication_amd/csrc/gemm_f32.hip"
#include "../../matrix-multiplication_amd/csrc/gemm_f32_duo.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  // default: BERT q.kT (m = n = 512, k = 64, A.B^T); "pv": probs.V (m = 512, n = 64, k = 512, A.B); "dv": P^T.dCtx
  const char* which = argc > 1 ? argv[1] : "qk";
  const bool pv = which[0] == 'p', dv = which[0] == 'd';
  const int batch = 384, m = 512, n = (pv || dv) ? 64 : 512, k = (pv || dv) ? 512 : 64;
  const int ta = dv ? 1 : 0, tb = (pv || dv) ? 0 : 1;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)batch * m * k * 4); hipMalloc(&B, (size_t)batch * n * k * 4); hipMalloc(&C, (size_t)batch * m * n * 4);
  {  // random operands: MFMA power (and with it the clock) depends on the data
    std::vector<float> h((size_t)batch * m * k);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) * (1.0f / 16777216.0f); }
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)batch * n * k * 4, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 6; ++it) {
#ifdef MI_GEMM_TIMING
    unsigned long long zero[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_phase), zero, sizeof(zero));
#endif
    hipEventRecord(e0);
    int st = mi_gemm_f32(ta, tb, m, n, k, A, ta ? m : k, (long)m * k, B, tb ? k : n, (long)n * k, C, n, (long)m * n, batch, nullptr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s abl=%d  %.3f ms (status %d)\n", which, MI_GEMM_ABL, ms, st);
#ifdef MI_GEMM_TIMING
    unsigned long long ph[16];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_gemm_phase), sizeof(ph));
    const double wgs = 384.0 * 16;
    printf("status %d  %.3f ms; mean cycles per workgroup (wave 0):\n", st, ms);
    const char* names[] = {"load wait + LDS store (2 k-tiles)", "barrier A (x2)", "MFMA issue (x2)", "barrier B (x2)", "epilogue"};
    double tot = 0;
    for (int q = 0; q < 5; ++q) { printf("  %-36s %10.0f\n", names[q], ph[q] / wgs); tot += ph[q] / wgs; }
    printf("  total %10.0f cycles per workgroup; %d workgroups per CU -> %.1f us of wave-0 time per CU slot\n", tot, (int)(wgs / 256), tot * wgs / 256 / 2.4e3);
#endif
  }
  return 0;
}
