// Developer probe: how fast can the chip retire the STORE side of a counting-sort scatter pass?
// Nothing is loaded or ranked: persistent workgroups (tile order as in csr_transpose.hip's scatter kernel)
// write tiles of 8192 entries whose entries leave as runs of P entries, run b of tile t landing at
//   b · (ntiles · P) + t · P            (the layout a radix pass with 8192/P uniformly filled bins produces).
// Variants: one 8-byte array (intermediate pass) / two 4-byte arrays (last pass); grid 256 / 64 / 8 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int TILE = 8192;
constexpr int THREADS = 1024;

template <bool TWO>
__global__ __launch_bounds__(THREADS) void store_kernel(uint2* __restrict__ out8, unsigned* __restrict__ outa,
                                                        unsigned* __restrict__ outb, int ntiles, int logp, int grid_xcd) {
  int t = blockIdx.x, t_end = ntiles, t_step = gridDim.x;
  if (grid_xcd) {
    const int per = (ntiles + 7) / 8, xcd = blockIdx.x & 7;
    t = xcd * per + (blockIdx.x >> 3);
    t_end = (xcd + 1) * per < ntiles ? (xcd + 1) * per : ntiles;
    t_step = gridDim.x / 8;
  }
  const long binsz = (long)ntiles << logp;
  for (; t < t_end; t += t_step) {
#pragma unroll 2
    for (int i = threadIdx.x; i < TILE; i += THREADS) {
      const long dst = (long)(i >> logp) * binsz + ((long)t << logp) + (i & ((1 << logp) - 1));
      if (TWO) {
        outa[dst] = (unsigned)i;
        outb[dst] = (unsigned)t;
      } else {
        out8[dst] = make_uint2((unsigned)i, (unsigned)t);
      }
    }
  }
}

int main() {
  const long n = (long)13440 * TILE;  // ≈ 110 M entries
  const int ntiles = (int)(n / TILE);
  uint2* out8;
  unsigned *outa, *outb;
  (void)hipMalloc(&out8, n * 8);
  (void)hipMalloc(&outa, n * 4);
  (void)hipMalloc(&outb, n * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int two = 0; two < 2; ++two)
    for (int grid : {256, 512, 64, 8})
      for (int logp : {2, 3, 4, 5, 6, 7, 9, 13}) {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
          (void)hipEventRecord(e0);
          if (two) hipLaunchKernelGGL(store_kernel<true>, dim3(grid), dim3(THREADS), 0, 0, out8, outa, outb, ntiles, logp, 1);
          else hipLaunchKernelGGL(store_kernel<false>, dim3(grid), dim3(THREADS), 0, 0, out8, outa, outb, ntiles, logp, 1);
          (void)hipEventRecord(e1);
          (void)hipDeviceSynchronize();
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (it > 0 && ms < best) best = ms;
        }
        printf("%s grid %3d  run of %4d entries (%5d B pieces): %.3f ms  %.2f TB/s\n", two ? "2x4B" : "1x8B", grid,
               1 << logp, (1 << logp) * (two ? 4 : 8), best, n * 8 / best * 1e-9);
        fflush(stdout);
      }
  return 0;
}
