// Developer probe: is v_mfma_f32_16x16x4_f32 bit-for-bit a k-ordered fmaf chain (like 32x32x2 is)?
// One wave computes D = A(16x4)·B(4x16) + C with random operands over many trials and compares with
// fmaf(a3,b3,fmaf(a2,b2,fmaf(a1,b1,fmaf(a0,b0,c)))) evaluated per element on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* A, const float* B, const float* C, float* D, float* R, int trials) {
  const int lane = threadIdx.x;
  for (int t = 0; t < trials; ++t) {
    const float* a = A + t * 64;   // a[k*16 + i]  (i = row 0..15, k = 0..3)
    const float* b = B + t * 64;   // b[k*16 + j]
    const float* c = C + t * 256;  // c[i*16 + j]
    // operand layout of 16x16x4: lane l supplies A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16];
    // D register r of lane l is D[i = 4*(l/16) + r][j = l % 16]
    const float av = a[(lane / 16) * 16 + lane % 16], bv = b[(lane / 16) * 16 + lane % 16];
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[(4 * (lane / 16) + r) * 16 + lane % 16];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * (lane / 16) + r, j = lane % 16;
      D[t * 256 + i * 16 + j] = acc[r];
      float s = c[i * 16 + j];
      for (int k = 0; k < 4; ++k) s = __builtin_fmaf(a[k * 16 + i], b[k * 16 + j], s);
      R[t * 256 + i * 16 + j] = s;
    }
  }
}
int main() {
  const int trials = 4096;
  std::vector<float> hA(trials * 64), hB(trials * 64), hC(trials * 256);
  unsigned x = 7;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * ((x & 3) ? 1.0f : 1024.0f); };
  for (auto& v : hA) v = rnd();
  for (auto& v : hB) v = rnd();
  for (auto& v : hC) v = rnd();
  float *A, *B, *C, *D, *R;
  (void)hipMalloc(&A, hA.size() * 4); (void)hipMalloc(&B, hB.size() * 4); (void)hipMalloc(&C, hC.size() * 4);
  (void)hipMalloc(&D, hC.size() * 4); (void)hipMalloc(&R, hC.size() * 4);
  (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(C, hC.data(), hC.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, A, B, C, D, R, trials);
  (void)hipDeviceSynchronize();
  std::vector<float> hD(hC.size()), hR(hC.size());
  (void)hipMemcpy(hD.data(), D, hD.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost);
  long diff = 0;
  for (size_t i = 0; i < hD.size(); ++i) diff += (*(unsigned*)&hD[i] != *(unsigned*)&hR[i]);
  printf("v_mfma_f32_16x16x4_f32 vs k-ordered fmaf chain: %ld of %zu elements differ\n", diff, hD.size());
  return 0;
}
