// Developer probe: is the scalar offset of a raw buffer access part of the range check on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/bufrange_probe.cpp -o tools/probes/bufrange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float* buf, float* out, int soff, int voff, int recs) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, recs, 0x00020000);
  const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, voff + (int)threadIdx.x * 4, soff, 0);
  out[threadIdx.x] = __builtin_bit_cast(float, v);
  __builtin_amdgcn_raw_buffer_store_b32(0x42280000u /* 42.0f */, r, voff + (int)threadIdx.x * 4 + 2048, soff, 0);
}
int main() {
  float *buf, *out;
  hipMalloc(&buf, 8192); hipMalloc(&out, 4096);
  float h[2048];
  for (int i = 0; i < 2048; ++i) h[i] = 1000.f + i;
  struct { int soff, voff, recs; const char* what; } cases[] = {
      {0, 0, 64, "records 64 B, no offsets: lanes 0-15 in range"},
      {128, 0, 64, "records 64 B, soffset 128: in range only if soffset is NOT checked"},
      {0, 128, 64, "records 64 B, voffset 128: out of range"},
      {32, 0, 64, "records 64 B, soffset 32: lanes 0-7 in range if soffset is checked, 0-15 if not"},
  };
  for (auto& c : cases) {
    hipMemcpy(buf, h, 8192, hipMemcpyHostToDevice);
    hipMemset(out, 0xff, 4096);
    hipLaunchKernelGGL(probe, dim3(1), dim3(16), 0, 0, buf, out, c.soff, c.voff, c.recs);
    float o[128], b[2048];
    hipMemcpy(o, out, 512, hipMemcpyDeviceToHost);
    hipMemcpy(b, buf, 8192, hipMemcpyDeviceToHost);
    printf("%s\n  dword loads :", c.what);
    for (int i = 0; i < 16; ++i) printf(" %g", o[i]);
    int stored = 0;
    for (int i = 0; i < 2048; ++i) stored += b[i] == 42.f;
    printf("\n  stores that landed (voffset + 2048, all beyond the records): %d\n", stored);
  }
  return 0;
}
