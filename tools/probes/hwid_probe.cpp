// Developer probe: where the 8 waves of a 512-thread workgroup land (SIMD id, wave slot) — HW_REG_HW_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
  extern __shared__ float lds[];
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_ID, 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
  if (threadIdx.x == 0) lds[0] = 1.f;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 8 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 139264);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 139264, 0, d);
  unsigned h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 6; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) printf("  w%d simd %u slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
    printf("\n");
  }
  int pat[256] = {0};
  for (int b = 0; b < 256; ++b) { int p = 0; for (int w = 0; w < 8; ++w) p = p * 4 + ((h[b * 8 + w] >> 4) & 3); int f = 0; for (int i = 0; i < 256; ++i) if (pat[i] == p + 1) f = 1; if (!f) for (int i = 0; i < 256; ++i) if (!pat[i]) { pat[i] = p + 1; break; } }
  printf("distinct SIMD patterns (base-4 digits w0..w7):"); for (int i = 0; i < 256 && pat[i]; ++i) { int p = pat[i] - 1; printf(" "); for (int w = 7; w >= 0; --w) printf("%d", (p >> (2 * w)) & 3); } printf("\n");
  return 0;
}
