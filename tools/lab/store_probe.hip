// How fast can 256 CUs WRITE?  (a) streaming 16-byte stores over a contiguous buffer, (b) the GEMM epilogue's pattern:
// every block writes a 256 x 256 bf16 tile of a row-major [M][N] matrix, 8 rows x 128 B per wave-instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/store_probe tools/lab/store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512) void stream_store(uint4* dst, long n) {
  const uint4 v = {1u, 2u, 3u, 4u};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = v;
}

// tile (mt, nt) of [M][N] bf16; 8 waves; wave w writes rows w*32 .. w*32+31 in 4 passes of 8 rows x 512 B?  The epilogue
// writes per wave-instruction 8 rows x 128 B (64 columns): lane -> row = l >> 3, 16-byte chunk = l & 7.
__global__ __launch_bounds__(512) void tile_store(char* dst, int M, int N, int mode) {
  const int ntn = N / 256;
  const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint4 v = {1u, 2u, 3u, 4u};
  if (mode == 0) {
    // epilogue pattern: wave (wr, wc) owns 2 x 64 rows and 2 x 32 columns; per band of 16 rows two instructions
    const int wr = wave >> 2, wc = wave & 3;
    for (int band = 0; band < 8; ++band)
      for (int ps = 0; ps < 2; ++ps) {
        const int row = mt * 256 + (band >> 2) * 128 + wr * 64 + (band & 3) * 16 + ps * 8 + (lane >> 3);
        const int c8 = lane & 7;                                   // 8-column chunk of the 64-column strip
        const int col = nt * 256 + (c8 >> 2) * 128 + wc * 32 + (c8 & 3) * 8;
        if (row < M) *reinterpret_cast<uint4*>(dst + ((long)row * N + col) * 2) = v;
      }
  } else {
    // whole 512-byte row segments: wave w writes rows w*32 .. +31, two rows per instruction (32 lanes x 16 B each)
    for (int it = 0; it < 16; ++it) {
      const int row = mt * 256 + wave * 32 + it * 2 + (lane >> 5);
      const int col = nt * 256 + (lane & 31) * 8;
      if (row < M) *reinterpret_cast<uint4*>(dst + ((long)row * N + col) * 2) = v;
    }
  }
}

int main() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  char* buf;
  const long bytes = 1L << 30;
  hipMalloc(&buf, bytes);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) stream_store<<<256 * 8, 512>>>((uint4*)buf, bytes / 16);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  printf("streaming stores: %.2f TB/s\n", 4.0 * bytes / (ms * 1e-3) / 1e12);
  const int M = 16000;
  for (int N : {1024, 4096})
    for (int mode : {0, 1}) {
      const int blocks = ((M + 255) / 256) * (N / 256);
      // rotate over 8 output matrices so that nothing stays in cache
      const long mat = (long)M * N * 2;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int r = 0; r < 8; ++r) tile_store<<<blocks, 512>>>(buf + r * mat, M, N, mode);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      printf("tile stores N=%d mode %d (%s): %.1f us per matrix, %.2f TB/s\n", N, mode, mode ? "512-B row segments" : "epilogue pattern",
             ms * 1e3 / 8, 8.0 * mat / (ms * 1e-3) / 1e12);
    }
  return 0;
}
