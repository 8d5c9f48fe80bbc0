// Developer probe (not part of the product): the persistent two-halves GEMM (gemm_f32_duo.hip) beside the tile
// kernels of gemm_f32.hip on BERT-base attention's products and a few squares — bit-equality and time.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Imatrix-multiplication_amd/csrc tools/probes/duo_probe.cpp -o tools/probes/duo_probe
//   (-DMI_DUO_ABL=1/2/4: no C stores / no MFMAs / no operand loads in the duo kernel — timing only)
#include "../../matrix-multiplication_amd/csrc/mi_status.hip"
#define MI_GEMM_SINGLE_TU  // everything in this unit
#include "../../matrix-multiplication_amd/csrc/gemm_f32.hip"
#include "../../matrix-multiplication_amd/csrc/gemm_f32_duo.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

struct Shape { const char* name; int ta, tb, batch, m, n, k; int share_a = 0; };  // share_a: every item reads the same A (stride 0: cache-resident)

int main(int argc, char** argv) {
  const Shape shapes[] = {
      {"qk   NT 384x(512x64.64x512)", 0, 1, 384, 512, 512, 64},
      {"pv   NN 384x(512x512.512x64)", 0, 0, 384, 512, 64, 512},
      {"dv   TN 384x(512x512.512x64)", 1, 0, 384, 512, 64, 512},
      {"pvh  NN 192x(512x512.512x64): one round of workgroups", 0, 0, 192, 512, 64, 512},
      {"pvq  NN 96x(512x512.512x64): half a round", 0, 0, 96, 512, 64, 512},
      {"pvd  NN 768x(512x512.512x64): four rounds", 0, 0, 768, 512, 64, 512},
      {"pvsh NN same, A shared by all items (no HBM stream)", 0, 0, 384, 512, 64, 512, 1},
      {"dvsh TN same, A shared by all items (no HBM stream)", 1, 0, 384, 512, 64, 512, 1},
      {"sq4k NN 4096^3", 0, 0, 1, 4096, 4096, 4096},
      {"sq4k NT 4096^3", 0, 1, 1, 4096, 4096, 4096},
      {"sq4k TN 4096^3", 1, 0, 1, 4096, 4096, 4096},
      {"sq4k TT 4096^3", 1, 1, 1, 4096, 4096, 4096},
      {"sq8k NT 8192^3", 0, 1, 1, 8192, 8192, 8192},
      {"fc   NT 16384x768.768x3072", 0, 1, 1, 16384, 3072, 768},
      {"fcb  NN 16384x3072.3072x768", 0, 0, 1, 16384, 768, 3072},
      {"gw1  TN 768x768 k16384", 1, 0, 1, 768, 768, 16384},
      {"gw2  TN 1536x768 k8192", 1, 0, 1, 1536, 768, 8192},
      {"gw3  TN 1152x1152 k8192", 1, 0, 1, 1152, 1152, 8192},
      {"gw4  TN 2048x512 k8192", 1, 0, 1, 2048, 512, 8192},
      {"gw5  TN 1024x1024 k8192", 1, 0, 1, 1024, 1024, 8192},
      {"gw6  TN 256x256 k65536", 1, 0, 1, 256, 256, 65536},
      {"gw7  TN 3072x768 k16384", 1, 0, 1, 3072, 768, 16384},
  };
  const char* only = argc > 1 ? argv[1] : nullptr;
  const bool check = MI_DUO_ABL == 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Shape& s : shapes) {
    if (only && !strstr(s.name, only)) continue;
    const size_t na = (size_t)s.batch * s.m * s.k, nb = (size_t)s.batch * s.n * s.k, nc = (size_t)s.batch * s.m * s.n;
    float *A, *B, *C1, *C2;
    hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C1, nc * 4); hipMalloc(&C2, nc * 4);
    {
      std::vector<float> h(std::max(na, nb));
      unsigned x = 12345;
      for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
      hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice);
      for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
      hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice);
    }
    hipMemset(C1, 0xff, nc * 4); hipMemset(C2, 0xee, nc * 4);
    const long lda = s.ta ? s.m : s.k, ldb = s.tb ? s.k : s.n;
    std::vector<float> t[3];
    int st[3] = {0, 0, 0};
    // blocks of back-to-back launches (the clock governor follows sustained load: a launch that starts from an idle
    // chip runs at a different clock), plans interleaved block by block
    const int reps = s.m >= 8192 ? 5 : 30;
    for (int round = 0; round < 5; ++round)
      for (int plan = 1; plan <= 2; ++plan) {
        mi_gemm_set_plan(plan);
        for (int r = 0; r < reps + 3; ++r) {
          if (r == 3) hipEventRecord(e0);
          st[plan] = mi_gemm_f32(s.ta, s.tb, s.m, s.n, s.k, A, lda, s.share_a ? 0 : (long)s.m * s.k, B, ldb, (long)s.n * s.k,
                                 plan == 1 ? C1 : C2, s.n, (long)s.m * s.n, s.batch, nullptr);
        }
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (round >= 1) t[plan].push_back(ms / reps);
      }
#ifdef MI_DUO_TIMING
    {
      static unsigned long long st_[256][8][64], re_[256][2];
      hipMemcpyFromSymbol(st_, HIP_SYMBOL(g_duo_stamps), sizeof(st_));
      hipMemcpyFromSymbol(re_, HIP_SYMBOL(g_duo_real), sizeof(re_));
      for (int wg : {0, 100}) {
        const double real_ns = (re_[wg][1] - re_[wg][0]) * 10.0;
        printf("  wg %d: stamps 0..43 span %.1f us by s_memrealtime, %lld cycles -> shader clock %.2f GHz\n", wg, real_ns / 1e3,
               (long long)(st_[wg][0][43] - st_[wg][0][0]), (st_[wg][0][43] - st_[wg][0][0]) / real_ns);
        for (int wv : {0, 4}) {
          printf("   wave %d deltas (cycles): ", wv);
          for (int e = 1; e < 44; ++e) printf("%s%lld", (e % 4 == 1) ? " | " : " ", (long long)(st_[wg][wv][e] - st_[wg][wv][e - 1]));
          printf("\n   total stamped cycles %lld -> clock %.2f GHz if whole body\n", (long long)(st_[wg][wv][43] - st_[wg][wv][0]), 0.0);
        }
      }
    }
#endif
    const double flop = 2.0 * s.batch * s.m * (double)s.n * s.k;
    for (int plan = 1; plan <= 2; ++plan) {
      std::sort(t[plan].begin(), t[plan].end());
      printf("%-32s %s  median %.4f ms  min %.4f ms  %.1f TFLOP/s (median)  status %d\n", s.name, plan == 1 ? "tiles" : "duo  ",
             t[plan][t[plan].size() / 2], t[plan][0], flop / t[plan][t[plan].size() / 2] / 1e9, st[plan]);
    }
    if (check && st[1] == 0 && st[2] == 0) {
      std::vector<float> h1(nc), h2(nc);
      hipMemcpy(h1.data(), C1, nc * 4, hipMemcpyDeviceToHost);
      hipMemcpy(h2.data(), C2, nc * 4, hipMemcpyDeviceToHost);
      size_t diff = 0, first = 0;
      for (size_t i = 0; i < nc; ++i)
        if (memcmp(&h1[i], &h2[i], 4) != 0) { if (!diff) first = i; ++diff; }
      printf("    bit-equal: %s (%zu of %zu differ%s)\n", diff ? "NO" : "yes", diff, nc, diff ? "" : "");
      if (diff) printf("    first difference at %zu: tiles %.9g duo %.9g\n", first, h1[first], h2[first]);
    }
    fflush(stdout);
    hipFree(A); hipFree(B); hipFree(C1); hipFree(C2);
  }
  mi_gemm_set_plan(0);
  return 0;
}
