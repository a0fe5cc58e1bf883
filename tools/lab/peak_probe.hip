// Measured peaks on the box the bench runs on (SURVEY 8d asks for them next to the vendor nominals):
//   * dense bf16 MFMA: every wave issues independent v_mfma_f32_32x32x16_bf16 chains from registers
//   * HBM: streaming copy of a 2 GiB buffer (read + write bytes / time)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/peak_probe tools/lab/peak_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out;
  hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  mfma_kernel<<<cus * 8, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_kernel<<<cus * 8, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)cus * 8 * 4 * iters * 4.0 * (2.0 * 32 * 32 * 16);
  const double mfma_tflops = flop / (ms * 1e-3) / 1e12;
  const long bytes = 2L << 30;
  uint4 *a, *b;
  hipMalloc(&a, bytes);
  hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes);
  copy_kernel<<<cus * 16, 256>>>(a, b, bytes / 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) copy_kernel<<<cus * 16, 256>>>(a, b, bytes / 16);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  const double hbm_tbs = 5.0 * 2.0 * bytes / (ms * 1e-3) / 1e12;
  printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, \"mfma_bf16_dense_tflops\": %.1f, "
         "\"hbm_copy_tb_per_s\": %.2f}\\n", prop.name, cus, prop.clockRate / 1000, mfma_tflops, hbm_tbs);
  return 0;
}
