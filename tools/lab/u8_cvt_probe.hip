// Lab (round 6): how v_cvt_pk_u8_f32 rounds and saturates on gfx950 -- decides VG_U8_CVT in csrc/vg_common.h.
//   hipcc --offload-arch=gfx950 -O2 tools/lab/u8_cvt_probe.hip -o tools/lab/u8_cvt_probe && tools/lab/u8_cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}
int main() {
  const float v[] = {0.0f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 25.999f, 26.0f, 26.5f, 27.5f, 254.5f, 255.4f, 255.5f, 300.0f, -0.4f, -0.6f, -5.0f, 1e9f};
  const int n = sizeof(v) / sizeof(v[0]);
  float* din; unsigned* dout; unsigned h[32];
  hipMalloc(&din, sizeof(v)); hipMalloc(&dout, sizeof(h));
  hipMemcpy(din, v, sizeof(v), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(din, dout, n);
  hipMemcpy(h, dout, n * sizeof(unsigned), hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("v_cvt_pk_u8_f32(%g) = %u\n", v[i], h[i]);
  float nanv = __builtin_nanf("");
  hipMemcpy(din, &nanv, 4, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(din, dout, 1);
  hipMemcpy(h, dout, 4, hipMemcpyDeviceToHost);
  printf("v_cvt_pk_u8_f32(nan) = %u\n", h[0]);
  return 0;
}
