// Developer probe: streaming-store ceilings for a 403 MB output (BERT q.kT scores) — dword vs dwordx4 per lane,
// nt vs plain, 256 workgroups of W waves each writing 128x128-float tiles row by row (the epilogue's shape).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VEC, bool NT>
__global__ void k(float* C, long total_floats, int per_wg_tiles) {
  // each wave writes 64x64 patches of a 128x128 tile of a 512-wide matrix: rows of 64 floats (256 B)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long tile0 = (long)blockIdx.x * per_wg_tiles;
  for (int t = wave / 4; t < per_wg_tiles; t += nw / 4) {
    const long tile = tile0 + t;               // tile of 128x128 in a [rows][512] matrix
    const long tm = tile / 4, tn = tile % 4;
    const int w4 = wave & 3;
    float* base = C + (tm * 128 + (w4 >> 1) * 64) * 512 + tn * 128 + (w4 & 1) * 64;
    if (VEC == 1) {
      for (int r = 0; r < 64; r += 2) {  // one instr: 2 rows x 32 floats
        float* p = base + (long)(r + (lane >> 5)) * 512 + (lane & 31);
        if (NT) { __builtin_nontemporal_store(1.0f, p); __builtin_nontemporal_store(2.0f, p + 32); }
        else { p[0] = 1.0f; p[32] = 2.0f; }
      }
    } else {
      for (int r = 0; r < 64; r += 4) {  // one instr: 4 rows x 64 floats (16 lanes x 16 B per row)
        f32x4* p = reinterpret_cast<f32x4*>(base + (long)(r + (lane >> 4)) * 512 + (lane & 15) * 4);
        if (NT) __builtin_nontemporal_store(f32x4{1, 2, 3, 4}, p); else *p = f32x4{1, 2, 3, 4};
      }
    }
  }
}
int main() {
  const long total = 384L * 512 * 512;
  float* C; hipMalloc(&C, total * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles = 6144, per = tiles / 256;
  auto run = [&](const char* name, auto kern, int waves) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, C, total, per);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, C, total, per);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-28s waves/CU %2d  %.4f ms  %.2f TB/s\n", name, waves, ms, total * 4 / ms / 1e9);
  };
  for (int w : {4, 8, 16}) {
    run("dword   nt", k<1, true>, w);
    run("dword   plain", k<1, false>, w);
    run("dwordx4 nt", k<4, true>, w);
    run("dwordx4 plain", k<4, false>, w);
  }
  return 0;
}
