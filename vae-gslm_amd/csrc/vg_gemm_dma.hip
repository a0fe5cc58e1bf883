// bf16 MFMA GEMM, LDS-DMA pipelined variant (the fast path of vg_gemm).
//
// Same operand modes, LDS images and epilogue as vg_gemm.hip, but the global->LDS
// staging uses `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB per wave-instruction,
// no VGPR round trip, hardware range check = zero fill past the end of the
// buffer).  The LDS images stay swizzled: the DMA writes lane-linear, so the
// swizzle is applied to each lane's SOURCE address and undone by the same XOR on
// the fragment reads (both-sides rule).
//
// Pipeline (one barrier per K tile): the tile for step k+1 is issued right after
// the barrier that publishes tile k, and lands while the MFMAs of tile k run.
// Tile shapes are template parameters; the host picks per problem shape.
#include "vg_gemm_tile.h"

namespace {

template <bool A_TR, bool B_TR, int BM, int BN, int WM, int WN, int STAGES, bool COLSUM = false, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(WM* WN * 64) void gemm_dma_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int PIECES = ((BM + BN) / 8) / NW;            // LDS-DMA instructions per wave per K tile
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;     // 16x16 MFMA tiles per wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware remap (bijective): blocks that share an XCD get a contiguous run of tiles,
  // n fastest, so the A row-panel and the B panels they share stay in that XCD's L2.
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = ntn * ntm;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  int mt = wg / ntn, nt = wg % ntn;
  if (p.group_m > 0) {      // bands of group_m row-tiles, m fastest inside a band: the 32 tiles in flight on an XCD
    const int gsz = p.group_m * ntn, gid = wg / gsz, first = gid * p.group_m;   // form a group_m x (32 / group_m)
    const int gm = min(ntm - first, p.group_m), rem = wg - gid * gsz;           // block -> fewer distinct panels
    mt = first + rem % gm;
    nt = rem / gm;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nkt = (kend - kbeg + BK - 1) / BK;   // K tails are zero-filled (rows past K, or chunks past a row's end)

  // buffer descriptors: the hardware range check zero-fills rows past the end of each operand
  const long lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  const long a_bytes = A_TR ? (long)(p.K - 1) * lda_b + (long)p.M * 2 : (long)(p.M - 1) * lda_b + (long)p.K * 2;
  const long b_bytes = B_TR ? (long)(p.K - 1) * ldb_b + (long)p.N * 2 : (long)(p.N - 1) * ldb_b + (long)p.K * 2;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0,
                                                                (int)min(a_bytes, 0x7fffffffL), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0,
                                                                (int)min(b_bytes, 0x7fffffffL), 0x00020000);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // COLSUM (weight-gradient launches): row sums of A = the bias gradient, from one extra MFMA of the
  // staged A fragments against a constant all-ones operand (no extra LDS traffic).  All blocks of an
  // m-panel stage the same A tiles, so the K-steps are dealt round-robin over the panel's blocks and
  // over the WN waves that share a fragment set: every block pays 1/ntn of the extra MFMAs.
  f32x4 csum[COLSUM ? TM : 1];
  bf16x8 ones;
  const int cs_n = nt;
  int cs_turn = 0, cs_wave = 0;             // kt % ntn and (kt / ntn) % WN, kept incrementally (no division per K-step)
  bool cs_any = false;
  if constexpr (COLSUM) {
#pragma unroll
    for (int i = 0; i < TM; ++i) csum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  }

  // ring of STAGES LDS stages: tiles kt .. kt+STAGES-2 are in flight while tile kt is consumed
  auto issue = [&](int kt) {
    char* st = smem + (kt % STAGES) * STAGE;
    dma_tile<A_TR, BM, NW>(ra, st, lda_b, m0, kbeg + kt * BK, wave, lane, kend);
    dma_tile<B_TR, BN, NW>(rb, st + A_BYTES, ldb_b, n0, kbeg + kt * BK, wave, lane, kend);
  };
#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (s0 < nkt) issue(s0);

  const int arow = wm * (BM / WM), bcol = wn * (BN / WN);
  for (int kt = 0; kt < nkt; ++kt) {
    char* cur = smem + (kt % STAGES) * STAGE;
    // counted wait: only this wave's pieces of tile kt must have landed, younger tiles stay in flight
    if (STAGES >= 3 && kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // everyone's pieces landed; tile kt-1 is no longer read
    if (kt + STAGES - 1 < nkt) issue(kt + STAGES - 1);
    const char* ta = cur;
    const char* tb = cur + A_BYTES;
    // two 32-deep k-steps per tile; the fragments of step 1 are read while the MFMAs of step 0 run
    bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = frag_of<A_TR>(ta, arow + i * 16, 0, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = frag_of<B_TR>(tb, bcol + j * 16, 0, lane);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[1][i] = frag_of<A_TR>(ta, arow + i * 16, 1, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[1][j] = frag_of<B_TR>(tb, bcol + j * 16, 1, lane);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
      if constexpr (COLSUM) {
        if (p.colsum_out != nullptr &&
            (p.colsum_rr ? (cs_turn == cs_n && cs_wave == wn) : (cs_n == 0 && wn == 0))) {
          cs_any = true;
#pragma unroll
          for (int i = 0; i < TM; ++i)
            csum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s][i], ones, csum[i], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (COLSUM) {
      if (++cs_turn == ntn) {
        cs_turn = 0;
        if (++cs_wave == WN) cs_wave = 0;
      }
    }
  }

  // ------------------------------------------------------------ epilogue
  static_assert(NW * 16 * (TN * 16 + 4) * 4 <= STAGES * STAGE, "epilogue strips must fit in the stage buffers");
  if constexpr (COLSUM) {
    if (cs_any && (lane & 15) == 0) {      // all columns of csum are equal
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int m = m0 + arow + i * 16 + 4 * (lane >> 4) + rr;
          if (m < p.M) atomicAdd(p.colsum_out + m, csum[i][rr]);
        }
    }
  }
  if constexpr (EPI == EPI_GENERIC) tile_epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, wg, nwg);
  else tile_epilogue_lean<BM, BN, WM, WN, false, false, EPI>(p, acc, smem, m0, n0);
}

// =====================================================================================
// 32-deep K tiles, 4-stage ring (vg_gemm tile_cfg 6 / 7).
//
// With 64-deep tiles a 256x256 block has room for two 64 KB stages only: one tile is in flight, its request can
// only be issued once the barrier has retired the previous stage, and the fill pipe idles from the moment that
// tile lands until the next barrier (measured: 1.37 us per 64-deep step on L2-resident operands, 1.85-2.1 us under
// load, against 0.85 us of MFMA work).  Four 32 KB stages keep THREE tiles (96 KB) in flight behind the one being
// consumed, requested three steps ahead, so the global->LDS stream never drains; and because a tile is exactly
// one MFMA step, its fragments are read into the second register set while the MFMAs of the previous tile run, so
// no MFMA waits for an LDS read issued after a barrier.
//  ROW image: [R rows][64 B], 16-byte chunk position p of row r holds source chunk p ^ ((r >> 2) & 3): the 16 rows
//             x one chunk a 16-lane group reads tile one 256-byte bank row
//  TR  image: R/128 sub-images of [32 krows][256 B], same granule swizzle as the 64-deep image
constexpr int BK32 = 32;

template <bool TR, int R, int NW>
VG_DEVICE void dma_tile32(__amdgpu_buffer_rsrc_t rsrc, char* tile, long ld_bytes, int rc0, int k0, int wave, int lane,
                          int klim) {
  constexpr int PER_WAVE = (R / 16) / NW;
  static_assert(PER_WAVE >= 1, "tile too small for the wave count");
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int piece = j * NW + wave;           // 1-KiB piece index inside the tile
    unsigned voff;
    if constexpr (!TR) {
      const int row = piece * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      voff = (unsigned)((long)(rc0 + row) * ld_bytes + (long)(k0 + chunk * 8) * 2);
      if (k0 + chunk * 8 >= klim) voff = 0x7ffffff0u;   // K tail: zero-fill through the range check
    } else {
      const int sub = piece >> 3;              // 128-column sub-image (8 pieces of 4 k-rows)
      const int krow = (piece & 7) * 4 + (lane >> 4);
      const int p16 = lane & 15;
      const int gran = (p16 >> 2) ^ (krow & 3);
      const int half = ((p16 >> 1) & 1) ^ ((krow >> 3) & 1);
      const int col = sub * 128 + gran * 32 + half * 16 + (p16 & 1) * 8;
      voff = (unsigned)((long)(k0 + krow) * ld_bytes + (long)(rc0 + col) * 2);
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(void, tile + piece * 1024), 16, voff, 0, 0, 0);
  }
}

template <bool TR>
VG_DEVICE bf16x8 frag32_of(const char* tile, int rc, int lane) {
  if constexpr (!TR) {
    const int row = rc + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((((lane >> 4) ^ (row >> 2)) & 3) << 4));
  } else {
    return TrTile<bf16_t, 128>::frag16(tile + (rc >> 7) * (32 * 256), 0, rc & 127, 0, lane);
  }
}

template <bool A_TR, bool B_TR, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_dma32_kernel(GemmParams p) {
  constexpr int NW = WM * WN, STAGES = 4;
  constexpr int PIECES = ((BM + BN) / 16) / NW;           // LDS-DMA instructions per wave per K tile
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int A_BYTES = BM * BK32 * 2, B_BYTES = BN * BK32 * 2, STAGE = A_BYTES + B_BYTES;
  static_assert(NW * 16 * (TN * 16 + 4) * 4 <= STAGES * STAGE, "epilogue strips must fit in the stage buffers");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware remap + bands of group_m row-tiles (see gemm_dma_kernel)
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = ntn * ntm;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  int mt = wg / ntn, nt = wg % ntn;
  if (p.group_m > 0) {
    const int gsz = p.group_m * ntn, gid = wg / gsz, first = gid * p.group_m;
    const int gm = min(ntm - first, p.group_m), rem = wg - gid * gsz;
    mt = first + rem % gm;
    nt = rem / gm;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nkt = (kend - kbeg + BK32 - 1) / BK32;

  const long lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  const long a_bytes = A_TR ? (long)(p.K - 1) * lda_b + (long)p.M * 2 : (long)(p.M - 1) * lda_b + (long)p.K * 2;
  const long b_bytes = B_TR ? (long)(p.K - 1) * ldb_b + (long)p.N * 2 : (long)(p.N - 1) * ldb_b + (long)p.K * 2;
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0,
                                                                (int)min(a_bytes, 0x7fffffffL), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0,
                                                                (int)min(b_bytes, 0x7fffffffL), 0x00020000);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int kt) {
    char* st = smem + (kt & (STAGES - 1)) * STAGE;
    dma_tile32<A_TR, BM, NW>(ra, st, lda_b, m0, kbeg + kt * BK32, wave, lane, kend);
    dma_tile32<B_TR, BN, NW>(rb, st + A_BYTES, ldb_b, n0, kbeg + kt * BK32, wave, lane, kend);
  };
  const int arow = wm * (BM / WM), bcol = wn * (BN / WN);
  // two fragment sets with COMPILE-TIME indices (a run-time set index would put the arrays in scratch)
  bf16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  auto read_frags = [&](int kt, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
    const char* ta = smem + (kt & (STAGES - 1)) * STAGE;
    const char* tb = ta + A_BYTES;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = frag32_of<A_TR>(ta, arow + i * 16, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = frag32_of<B_TR>(tb, bcol + j * 16, lane);
  };
  // waits until this wave's pieces of tile kt have landed; up to `ahead` tiles requested after it stay in flight
  auto wait_tile = [&](int kt, int ahead) {
    const int younger = min(ahead, nkt - 1 - kt);
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES) : "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // one K tile: publish tile kt+1 and read its fragments into the other set while the MFMAs of tile kt run
  auto step = [&](int kt, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN], bf16x8 (&na)[TM], bf16x8 (&nb)[TN]) {
    if (kt + 1 < nkt) {
      // after this barrier tile kt+1 is visible to every wave and nobody reads tile kt's stage any more (its
      // fragments are in registers), so that stage takes tile kt+4; tiles kt+2, kt+3 are still travelling
      wait_tile(kt + 1, 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + STAGES < nkt) issue(kt + STAGES);
      read_frags(kt + 1, na, nb);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // all four stages are free: tiles 0..3 are requested at once
#pragma unroll
  for (int s0 = 0; s0 < STAGES; ++s0)
    if (s0 < nkt) issue(s0);
  wait_tile(0, 3);
  __builtin_amdgcn_s_barrier();
  read_frags(0, fa0, fb0);
  for (int kt = 0; kt < nkt; kt += 2) {
    step(kt, fa0, fb0, fa1, fb1);
    if (kt + 1 < nkt) step(kt + 1, fa1, fb1, fa0, fb0);
  }
  tile_epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, wg, nwg);
}

template <bool A_TR, bool B_TR, int BM, int BN, int WM, int WN, int STAGES = 2, bool COLSUM = false, int EPI = EPI_GENERIC>
int launch_cfg(const GemmParams& p, int splits, hipStream_t stream) {
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
  auto k = gemm_dma_kernel<A_TR, B_TR, BM, BN, WM, WN, STAGES, COLSUM, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  VG_LAUNCH(k, dim3(ntn * ntm, 1, splits), dim3(WM * WN * 64), lds, stream, p);
  return 0;
}

template <bool A_TR, bool B_TR, int BM, int BN, int WM, int WN>
int launch_cfg32(const GemmParams& p, int splits, hipStream_t stream) {
  constexpr size_t lds = (size_t)4 * (BM + BN) * BK32 * 2;
  auto k = gemm_dma32_kernel<A_TR, B_TR, BM, BN, WM, WN>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  VG_LAUNCH(k, dim3(ntn * ntm, 1, splits), dim3(WM * WN * 64), lds, stream, p);
  return 0;
}

template <bool A_TR, bool B_TR>
int launch_mode(const GemmParams& p, int cfg, int splits, hipStream_t stream) {
  if constexpr (A_TR && B_TR) {   // fused bias gradient: 128x128 tiles only (register budget)
    if (p.colsum_out) return launch_cfg<A_TR, B_TR, 128, 128, 2, 2, 2, true>(p, splits, stream);
  }
  switch (cfg) {
    case 1:
      if constexpr (!A_TR) {     // the N <= 512 products of the conv stacks: plain / bias / residual / ReLU epilogues
        if (lean_epilogue_of(p, splits) == EPI_PLAIN) return launch_cfg<A_TR, B_TR, 128, 128, 2, 2, 2, false, EPI_PLAIN>(p, splits, stream);
      }
      return launch_cfg<A_TR, B_TR, 128, 128, 2, 2>(p, splits, stream);
    case 2: return launch_cfg<A_TR, B_TR, 256, 128, 4, 2>(p, splits, stream);
    case 3: return launch_cfg<A_TR, B_TR, 256, 256, 2, 4>(p, splits, stream);
    case 4: return launch_cfg<A_TR, B_TR, 128, 256, 2, 4>(p, splits, stream);
    case 5: return launch_cfg<A_TR, B_TR, 256, 128, 4, 2, 3>(p, splits, stream);   // 3-stage ring (144 KiB LDS)
    case 6: return launch_cfg32<A_TR, B_TR, 256, 256, 2, 4>(p, splits, stream);     // 32-deep tiles, 4 stages
    case 9:                                                                          // 192 rows: row-image A only
      if constexpr (!A_TR) return launch_cfg<A_TR, B_TR, 192, 256, 2, 4>(p, splits, stream);
      else return -1;
    case 7: return launch_cfg32<A_TR, B_TR, 128, 128, 2, 2>(p, splits, stream);
    default: return -1;
  }
}

}  // namespace

namespace vg_host {
// returns 0 if launched, -1 if this variant does not apply (caller falls back)
int gemm_dma_launch(const GemmParams& p, int a_tr, int b_tr, int cfg, int splits, hipStream_t stream) {
  if (cfg <= 0) return -1;
  if (!a_tr && !b_tr) return launch_mode<false, false>(p, cfg, splits, stream);
  if (!a_tr && b_tr) return launch_mode<false, true>(p, cfg, splits, stream);
  if (a_tr && b_tr) return launch_mode<true, true>(p, cfg, splits, stream);
  return -1;
}
}  // namespace vg_host
