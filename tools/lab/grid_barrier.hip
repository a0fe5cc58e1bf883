// Lab: what one grid-wide barrier costs inside a persistent launch on MI355X (256 CUs, 8 XCDs), for the decode
// step's "one launch per frame" design question.  Variants:
//   flat   : every block adds 1 to one counter (device-scope atomic), spins on it
//   tree   : blocks of an XCD (blockIdx & 7) meet on a per-XCD counter, the last arrival adds to the root, everyone
//            spins on the root's generation word
// Every spin is bounded (bails out and raises a flag) so that a mistake cannot hang the GPU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier tools/lab/grid_barrier.hip && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int SPIN_MAX = 1 << 20;

__device__ __forceinline__ unsigned ld_acq(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void flat_kernel(unsigned* cnt, int iters, int* bad, float* sink, int work) {
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 1.f;      // stand-in for a phase's arithmetic
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned target = (unsigned)(it + 1) * gridDim.x;
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (ld_acq(cnt) < target) { if (++spins > SPIN_MAX) { *bad = 1; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    if (*(volatile int*)bad) return;                                    // somebody gave up: everyone leaves
  }
  if (acc == 12345.f) *sink = acc;
}

__global__ void tree_kernel(unsigned* xcnt, unsigned* root, unsigned* gen, int iters, int* bad, float* sink, int work) {
  const int x = blockIdx.x & 7, per = gridDim.x / 8;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int w = 0; w < work; ++w) acc = acc * 1.0001f + 1.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(xcnt + x * 32, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)(it + 1) * per - 1) {                       // last block of this XCD
        const unsigned r = __hip_atomic_fetch_add(root, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (r == (unsigned)(it + 1) * 8 - 1) __hip_atomic_store(gen, (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      int spins = 0;
      while (ld_acq(gen) < (unsigned)(it + 1)) { if (++spins > SPIN_MAX) { *bad = 1; break; } __builtin_amdgcn_s_sleep(1); }
    }
    __syncthreads();
    if (*(volatile int*)bad) return;                                    // somebody gave up: everyone leaves
  }
  if (acc == 12345.f) *sink = acc;
}

int main() {
  unsigned* buf; int* bad; float* sink;
  CK(hipMalloc(&buf, 4096)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 2000;
  for (int blocks : {64, 128, 256}) {
    for (int threads : {256, 1024}) {
      for (int variant = 0; variant < 2; ++variant) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(buf, 0, 4096)); CK(hipMemset(bad, 0, 4));
          CK(hipEventRecord(a));
          if (variant == 0) flat_kernel<<<blocks, threads>>>(buf, iters, bad, sink, 0);
          else tree_kernel<<<blocks, threads>>>(buf, buf + 512, buf + 768, iters, bad, sink, 0);
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b));
          best = ms < best ? ms : best;
        }
        int hbad = 0; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        printf("blocks %4d x %4d threads  %s  %.3f us per barrier%s\n", blocks, threads, variant ? "tree" : "flat",
               best * 1e3f / iters, hbad ? "  (SPIN LIMIT HIT)" : "");
      }
    }
  }
  return 0;
}
