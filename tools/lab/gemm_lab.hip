// GEMM main-loop laboratory (not part of the product): NT bf16 GEMM variants timed with HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gemm_lab.hip -o tools/lab/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../vae-gslm_amd/csrc/vg_common.h"

using namespace vg;
namespace vg_host { void set_error(const char*, ...) {} int check_launch(const char*) { return 0; }
int prof_begin(int, double, hipStream_t) { return -1; } void prof_end(int, hipStream_t) {} }

constexpr int BK = 64;

template <int R, int NW>
VG_DEVICE void dma_rows(__amdgpu_buffer_rsrc_t rsrc, char* tile, long ld_bytes, int rc0, int k0, int wave, int lane) {
  constexpr int PER_WAVE = (R / 8) / NW;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int piece = j * NW + wave;
    const int row = piece * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const unsigned voff = (unsigned)((long)(rc0 + row) * ld_bytes + (long)(k0 + chunk * 8) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(void, tile + piece * 1024), 16, voff, 0, 0, 0);
  }
}

struct P { const bf16_t* A; const bf16_t* B; bf16_t* C; int M, N, K; };
static int g_nsets = 1;
static size_t g_strideA = 0, g_strideB = 0;   // elements between operand sets

// ABL: 0 full, 1 no DMA in loop, 2 no MFMA, 3 no ds_read (constant fragments)
// STAGES: LDS ring depth (2 = wait vmcnt(0) each tile; 3 = one tile stays in flight)
template <int BM, int BN, int WM, int WN, int STAGES, int ABL>
__global__ __launch_bounds__(WM* WN * 64) void lab_kernel(P p) {
  constexpr int NW = WM * WN, TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int PIECES = ((BM + BN) / 8) / NW;     // DMA instructions per wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM, nwg = ntn * ntm;
  const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
  const int nkt = p.K / BK;
  const long ldb = (long)p.K * 2;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((long)p.M * ldb), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((long)p.N * ldb), 0x00020000);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = zero16();

  auto issue = [&](int kt) {
    char* st = smem + (kt % STAGES) * STAGE;
    dma_rows<BM, NW>(ra, st, ldb, m0, kt * BK, wave, lane);
    dma_rows<BN, NW>(rb, st + A_BYTES, ldb, n0, kt * BK, wave, lane);
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nkt) issue(s);

  for (int kt = 0; kt < nkt; ++kt) {
    // tiles kt .. kt+STAGES-2 are in flight; wait until tile kt has landed
    if constexpr (STAGES == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (ABL != 1 && kt + STAGES - 1 < nkt) issue(kt + STAGES - 1);
    const char* ta = smem + (kt % STAGES) * STAGE;
    const char* tb = ta + A_BYTES;
    if constexpr (ABL == 4) {
      bf16x8 pa[2][TM], pb[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) pa[0][i] = RowTile<bf16_t, 64>::frag(ta, wm * (BM / WM) + i * 32 + (lane & 31), 0, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) pb[0][j] = RowTile<bf16_t, 64>::frag(tb, wn * (BN / WN) + j * 32 + (lane & 31), 0, lane);
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        const int cur = s & 1;
        if (s + 1 < BK / 16) {
#pragma unroll
          for (int i = 0; i < TM; ++i) pa[cur ^ 1][i] = RowTile<bf16_t, 64>::frag(ta, wm * (BM / WM) + i * 32 + (lane & 31), s + 1, lane);
#pragma unroll
          for (int j = 0; j < TN; ++j) pb[cur ^ 1][j] = RowTile<bf16_t, 64>::frag(tb, wn * (BN / WN) + j * 32 + (lane & 31), s + 1, lane);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[cur][i], pb[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      continue;
    }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (ABL == 3) { for (int e = 0; e < 8; ++e) fa[i][e] = (bf16_t)(float)(lane + e + i); }
        else fa[i] = RowTile<bf16_t, 64>::frag(ta, wm * (BM / WM) + i * 32 + (lane & 31), s, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (ABL == 3) { for (int e = 0; e < 8; ++e) fb[j][e] = (bf16_t)(float)(lane - e + j); }
        else fb[j] = RowTile<bf16_t, 64>::frag(tb, wn * (BN / WN) + j * 32 + (lane & 31), s, lane);
      }
      if (ABL == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j]));
      } else {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  // epilogue: LDS strip transpose, 16-byte stores
  constexpr int SW = TN * 32 + 4, CPR = TN * 4, RPP = 64 / CPR;
  __syncthreads();
  float* strip = reinterpret_cast<float*>(smem) + wave * (32 * SW);
  const int crow = lane / CPR, cch = lane % CPR;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) strip[acc_row(rr, lane) * SW + j * 32 + (lane & 31)] = acc[i][j][rr];
    for (int ps = 0; ps < 32 / RPP; ++ps) {
      const int rloc = ps * RPP + crow;
      const int m = m0 + wm * (BM / WM) + i * 32 + rloc, n = n0 + wn * (BN / WN) + cch * 8;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8 + 4);
      if (m >= p.M || n >= p.N) continue;
      bf16x8 o = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                  (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
      *reinterpret_cast<bf16x8*>(p.C + (long)m * p.N + n) = o;
    }
  }
}

template <int BM, int BN, int WM, int WN, int STAGES, int ABL>
float run(const P& p, const char* name, int iters) {
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
  auto k = lab_kernel<BM, BN, WM, WN, STAGES, ABL>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  auto at = [&](int i) { P q = p; q.A = p.A + (size_t)(i % g_nsets) * g_strideA; q.B = p.B + (size_t)(i % g_nsets) * g_strideB; return q; };
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(ntn * ntm), dim3(WM * WN * 64), lds, 0, at(i));
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(ntn * ntm), dim3(WM * WN * 64), lds, 0, at(i + 3));
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double t = ms * 1e-3 / iters, fl = 2.0 * p.M * p.N * p.K;
  printf("%-38s %4dx%4dx%4d  %7.1f us  %7.1f TF  (lds %zu KB)\n", name, p.M, p.N, p.K, t * 1e6, fl / t / 1e12, lds / 1024);
  fflush(stdout);
  return (float)t;
}

typedef __attribute__((ext_vector_type(4))) float f32x4v;

// ---- variant B: v_mfma_f32_16x16x32_bf16, wave tile (BM/WM) x (BN/WN) as 16x16 tiles, optional
//      software-pipelined fragment reads (PIPE = 1: fragments of k-step s+1 are read before the MFMAs of s)
template <int BM, int BN, int WM, int WN, int PIPE, int STAGES = 2>
__global__ __launch_bounds__(WM* WN * 64) void lab16_kernel(P p) {
  constexpr int PIECES16 = ((BM + BN) / 8) / (WM * WN);
  constexpr int NW = WM * WN, TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM, nwg = ntn * ntm;
  const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
  const int nkt = p.K / BK;
  const long ldb = (long)p.K * 2;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((long)p.M * ldb), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((long)p.N * ldb), 0x00020000);
  f32x4v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int kt) {
    char* st = smem + (kt % STAGES) * STAGE;
    dma_rows<BM, NW>(ra, st, ldb, m0, kt * BK, wave, lane);
    dma_rows<BN, NW>(rb, st + A_BYTES, ldb, n0, kt * BK, wave, lane);
  };
  // fragment of k-step s (32 deep): row = base + (lane & 15), chunk = 4 s + (lane >> 4)
  auto frag = [&](const char* tile, int row0, int s) -> bf16x8 {
    const int row = row0 + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(tile + RowTile<bf16_t, 64>::chunk_off(row, 4 * s + (lane >> 4)));
  };
#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (s0 < nkt) issue(s0);
  for (int kt = 0; kt < nkt; ++kt) {
    // tiles kt .. kt+STAGES-2 are in flight: wait for tile kt only
    const int ahead = min(STAGES - 2, nkt - 1 - kt);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES16) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES16) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + STAGES - 1 < nkt) issue(kt + STAGES - 1);
    const char* ta = smem + (kt % STAGES) * STAGE;
    const char* tb = ta + A_BYTES;
    bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = frag(ta, wm * (BM / WM) + i * 16, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = frag(tb, wn * (BN / WN) + j * 16, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cur = PIPE ? (s & 1) : 0;
      if (PIPE) {
        if (s + 1 < 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[cur ^ 1][i] = frag(ta, wm * (BM / WM) + i * 16, s + 1);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[cur ^ 1][j] = frag(tb, wn * (BN / WN) + j * 16, s + 1);
        }
      } else if (s > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = frag(ta, wm * (BM / WM) + i * 16, s);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = frag(tb, wn * (BN / WN) + j * 16, s);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }
  // epilogue: per 16-row band through an LDS strip [16][TN*16 + 4]
  constexpr int SW = TN * 16 + 4, CPR = TN * 2, RPP = 64 / CPR;
  __syncthreads();
  float* strip = reinterpret_cast<float*>(smem) + wave * (16 * SW);
  const int crow = lane / CPR, cch = lane % CPR;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) strip[((lane >> 4) * 4 + rr) * SW + j * 16 + (lane & 15)] = acc[i][j][rr];
    for (int ps = 0; ps < (16 + RPP - 1) / RPP; ++ps) {
      const int rloc = ps * RPP + crow;
      if (rloc >= 16) continue;
      const int m = m0 + wm * (BM / WM) + i * 16 + rloc, n = n0 + wn * (BN / WN) + cch * 8;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8 + 4);
      if (m >= p.M || n >= p.N) continue;
      bf16x8 o = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                  (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
      *reinterpret_cast<bf16x8*>(p.C + (long)m * p.N + n) = o;
    }
  }
}

template <int BM, int BN, int WM, int WN, int PIPE, int STAGES = 2>
float run16(const P& p, const char* name, int iters) {
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
  auto k = lab16_kernel<BM, BN, WM, WN, PIPE, STAGES>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  auto at = [&](int i) { P q = p; q.A = p.A + (size_t)(i % g_nsets) * g_strideA; q.B = p.B + (size_t)(i % g_nsets) * g_strideB; return q; };
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(ntn * ntm), dim3(WM * WN * 64), lds, 0, at(i));
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(ntn * ntm), dim3(WM * WN * 64), lds, 0, at(i + 3));
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double t = ms * 1e-3 / iters, fl = 2.0 * p.M * p.N * p.K;
  printf("%-38s %4dx%4dx%4d  %7.1f us  %7.1f TF\n", name, p.M, p.N, p.K, t * 1e6, fl / t / 1e12);
  fflush(stdout);
  return (float)t;
}


// ---- variant C (round 5, exploration for the next main loop; RESULT: as written the compiler keeps 256 VGPRs + 256 AGPRs
//      and spills 343 registers (576 B of scratch per lane): 88 TFLOP/s.  With one wave per SIMD and every register in
//      use the allocation has to be done by hand -- see DESIGN.md "what is ranked next"): FOUR waves of 128 x 128, K steps of 32 on a four-slot ring
//      (32 KB per slot: A 256 x 32 and B 256 x 32, rows of 64 bytes = one 16 x 32 MFMA fragment block per KB, no swizzle needed),
//      the fragment reads of step s + 1 placed between the MFMAs of step s (sched_group_barrier), one barrier per step.
template <int PIPE>
__global__ __launch_bounds__(256) void lab4_kernel(P p) {
  constexpr int BM = 256, BN = 256, BKS = 32, NS = 4, SLOT = (BM + BN) * BKS * 2, A_BYTES = BM * BKS * 2;
  constexpr int PIECES = SLOT / 1024 / 4;            // LDS-DMA instructions per wave and step (8)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM, nwg = ntn * ntm;
  const int orig = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
  const int nks = p.K / BKS;
  const long ldb = (long)p.K * 2;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((long)p.M * ldb), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((long)p.N * ldb), 0x00020000);
  f32x4v acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // one DMA instruction = 16 rows x 64 bytes (lane: row l >> 2, 16-byte chunk l & 3) -> 1 KB of LDS in lane order
  auto issue = [&](int ks) {
    char* st = smem + (ks % NS) * SLOT;
#pragma unroll
    for (int j = 0; j < PIECES / 2; ++j) {
      const int piece = j * 4 + wave;                  // 16 pieces of 16 rows per operand
      const int row = piece * 16 + (lane >> 2);
      const unsigned va = (unsigned)((long)(m0 + row) * ldb + (long)(ks * BKS + (lane & 3) * 8) * 2);
      const unsigned vb = (unsigned)((long)(n0 + row) * ldb + (long)(ks * BKS + (lane & 3) * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(void, st + piece * 1024), 16, va, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(void, st + A_BYTES + piece * 1024), 16, vb, 0, 0, 0);
    }
  };
  // fragment block b (16 rows) of an operand image: lane reads row (lane & 15), chunk (lane >> 4): 1 KB in lane order
  auto frag = [&](const char* img, int b) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(img + b * 1024 + (lane & 15) * 64 + (lane >> 4) * 16);
  };
  bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
  auto read_step = [&](int ks, bf16x8 (&fa)[8], bf16x8 (&fb)[8]) {
    const char* st = smem + (ks % NS) * SLOT;
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = frag(st, wm * 8 + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[j] = frag(st + A_BYTES, wn * 8 + j);
  };
  auto step = [&](int ks, bf16x8 (&fa)[8], bf16x8 (&fb)[8], bf16x8 (&na)[8], bf16x8 (&nb)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the fragments of step ks are in registers (every wave: slot ks is free)
    if (ks + 1 < nks) {
      if (ks + 3 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
      else if (ks + 2 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (ks + NS < nks) issue(ks + NS);
    if (ks + 1 < nks) read_step(ks + 1, na, nb);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    if constexpr (PIPE == 1) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
      }
    }
  };
#pragma unroll
  for (int s0 = 0; s0 < NS; ++s0)
    if (s0 < nks) issue(s0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES) : "memory");
  __builtin_amdgcn_s_barrier();
  read_step(0, fa0, fb0);
  for (int ks = 0; ks < nks; ks += 2) {
    step(ks, fa0, fb0, fa1, fb1);
    if (ks + 1 < nks) step(ks + 1, fa1, fb1, fa0, fb0);
  }
  constexpr int SW = 8 * 16 + 4, CPR = 16, RPP = 4;
  __syncthreads();
  float* strip = reinterpret_cast<float*>(smem) + wave * (16 * SW);
  const int crow = lane / CPR, cch = lane % CPR;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) strip[((lane >> 4) * 4 + rr) * SW + j * 16 + (lane & 15)] = acc[i][j][rr];
    for (int ps = 0; ps < 16 / RPP; ++ps) {
      const int rloc = ps * RPP + crow;
      const int m = m0 + wm * 128 + i * 16 + rloc, n = n0 + wn * 128 + cch * 8;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8 + 4);
      if (m >= p.M || n >= p.N) continue;
      bf16x8 o = {(bf16_t)lo[0], (bf16_t)lo[1], (bf16_t)lo[2], (bf16_t)lo[3],
                  (bf16_t)hi[0], (bf16_t)hi[1], (bf16_t)hi[2], (bf16_t)hi[3]};
      *reinterpret_cast<bf16x8*>(p.C + (long)m * p.N + n) = o;
    }
  }
}

template <int PIPE>
float run4(const P& p, const char* name, int iters) {
  constexpr size_t lds = 4 * 512 * 32 * 2;
  auto k = lab4_kernel<PIPE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  auto at = [&](int i) { P q = p; q.A = p.A + (size_t)(i % g_nsets) * g_strideA; q.B = p.B + (size_t)(i % g_nsets) * g_strideB; return q; };
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(ntn * ntm), dim3(256), lds, 0, at(i));
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(ntn * ntm), dim3(256), lds, 0, at(i + 3));
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double t = ms * 1e-3 / iters, fl = 2.0 * p.M * p.N * p.K;
  printf("%-38s %4dx%4dx%4d  %7.1f us  %7.1f TF\n", name, p.M, p.N, p.K, t * 1e6, fl / t / 1e12);
  fflush(stdout);
  return (float)t;
}

int main() {
  const int M = getenv("M") ? atoi(getenv("M")) : 8000;
  const int shapes[3][2] = {{4096, 1024}, {1024, 1024}, {1024, 4096}};
  for (auto& sh : shapes) {
    const int N = sh[0], K = sh[1];
    std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K);
    srand(1);
    auto rnd = [] { union { float f; uint32_t u; } c; c.f = (rand() / (float)RAND_MAX) * 2.f - 1.f; return (uint16_t)(c.u >> 16); };
    for (auto& v : ha) v = rnd();
    for (auto& v : hb) v = rnd();
    P p;
    g_nsets = getenv("COLD") ? (int)(300e6 / ((ha.size() + hb.size()) * 2)) + 2 : 1;
    g_strideA = ha.size(); g_strideB = hb.size();
    hipMalloc((void**)&p.A, ha.size() * 2 * g_nsets); hipMalloc((void**)&p.B, hb.size() * 2 * g_nsets); hipMalloc((void**)&p.C, (size_t)M * N * 2);
    for (int sidx = 0; sidx < g_nsets; ++sidx) {
      hipMemcpy((void*)(p.A + (size_t)sidx * g_strideA), ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
      hipMemcpy((void*)(p.B + (size_t)sidx * g_strideB), hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    }
    printf("operand sets: %d\n", g_nsets);
    p.M = M; p.N = N; p.K = K;
    const int it = 36;
    run16<128, 128, 2, 2, 1>(p, "128x128 4w 2st mfma16 pipe", it);
    run16<128, 128, 2, 4, 1, 3>(p, "128x128 8w(2x4) 3st mfma16 pipe", it);
    run16<128, 128, 2, 4, 1, 4>(p, "128x128 8w(2x4) 4st mfma16 pipe", it);
    run16<128, 128, 4, 2, 1, 4>(p, "128x128 8w(4x2) 4st mfma16 pipe", it);
    run16<256, 128, 4, 2, 1, 2>(p, "256x128 8w(4x2) 2st mfma16 pipe", it);
    run16<256, 128, 4, 2, 1, 3>(p, "256x128 8w(4x2) 3st mfma16 pipe", it);
    run16<256, 256, 2, 4, 1, 2>(p, "256x256 8w(2x4) 2st mfma16 pipe", it);
    // round 5: FOUR waves of 128 x 128 (one per SIMD, 256 accumulator registers per lane in AGPRs): half the fragment
    // reads per MFMA of the 8-wave layouts, the reads of k-step s + 1 issued before the MFMAs of s inside the wave
    run16<256, 256, 2, 2, 1, 2>(p, "256x256 4w(2x2) 2st mfma16 pipe", it);
    run16<256, 256, 2, 2, 0, 2>(p, "256x256 4w(2x2) 2st mfma16 nopipe", it);
    run4<0>(p, "256x256 4w k32 ring4 (compiler order)", it);
    run4<1>(p, "256x256 4w k32 ring4 (reads between MFMAs)", it);
    hipFree((void*)p.A); hipFree((void*)p.B); hipFree((void*)p.C);
  }
  return 0;
}
