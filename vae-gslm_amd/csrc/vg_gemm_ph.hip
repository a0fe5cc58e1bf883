// bf16 MFMA GEMM, phase-pipelined 256x256 tile (vg_gemm tile_cfg 11 / 12 / 13, vg_gemm_grouped): the main loops the large
// products of the training step run on.
//
// Why other main loops than gemm_dma_kernel.  That kernel keeps ONE 64 KB K tile in flight, requested after the
// barrier that retires the previous one: its request burst, its landing and the 24 fragment reads of the tile all
// queue behind that barrier, and both waves of a SIMD reach their MFMA run together.  Here the K tile is cut into
// four 16 KB half-tile images that stream through a slot ring several phases ahead of their reader, and the two
// wave groups of the block (one wave of each per SIMD) are kept out of step, so that on every SIMD one wave issues
// MFMAs while its partner reads fragments and issues the next images' LDS-DMA.
//
//   block   256 x 256 x 64, 8 waves = 2 (rows) x 4 (columns); wave (wr, wc) owns rows {wr*64 .. +63} of BOTH row halves
//           of the tile and columns {wc*32 .. +31} of BOTH column halves, so a phase reads one specific half-tile
//           image and an image is dead right after its single reading phase: that is what lets the images stream
//   images  s = 4 t + h of K tile t: h = 0 B half 0, 1 A half 0, 2 B half 1, 3 A half 1 (same LDS images as
//           gemm_dma_kernel: [128 rows][64 k] row image or [64 k][128] k-major, swizzled on the SOURCE address)
//   tails   M / N by the hardware range check of the buffer descriptor, whose window moves with the K tile; requests
//           past the last K tile have an empty window and become zero fills of slots nobody reads, so every phase
//           keeps the same counted wait and there is no tail code.  K must be a multiple of 64 per split.
//
// Three schedules, measured per operand mode at the layer shapes (tools/lab/ph_check.py, one box, cold operands;
// TFLOP/s of 2-stage 256x256 / ring (cfg 11) / complementary (12) / complementary with long phases (13)):
//   NT (row images)            FFN-out fwd 1041 / 1033-1092 / 1107-1162 / 1105     -> complementary
//   NN (B k-major)             FFN-in dgrad 787 / 1154-1181 / 992-1005 / 1244      -> complementary, long phases
//   TN (both k-major, split-K) W1 wgrad 722 / 825-840 / 774-778 / 868              -> complementary, long phases
// (a first version with two barriers per 16-MFMA phase and the second group one barrier behind reached 1087-1135 /
// 1069-1084 / 755-765 and was dropped)
// In-kernel stamps of the complementary loop (a diagnostic build of round 2, not kept): an interval is ~650 cycles for 2 x 256 cycles
// of MFMA issue (79 %), the chip holds ~2.0 GHz under it; entry to first MFMA 2.0 us; the plain bf16 store epilogue
// of a 256x256 tile takes 8.7 us = 3.8 TB/s over 256 CUs, i.e. it runs at the HBM write rate and only overlapping
// it with another tile's main loop can hide it.
#include "vg_gemm_tile.h"

namespace {

constexpr int HALF_BYTES = 16384;            // one half-tile image
constexpr int BUF_BYTES = 4 * HALF_BYTES;    // B0 A0 B1 A1 of one K tile

// per-lane source offsets of this wave's two 1-KiB pieces of a half-tile image (same images as dma_tile<..., 128, 8>)
// HR = rows (row image) / columns (k-major image) of the operand half an image holds: 128, or 96 for the A operand of
// the 192-row tile (tile_cfg 15).  The image keeps its 128-wide layout; the part past HR is never read, and its
// requests get an offset outside every buffer window (the range check turns them into zero fills without a fetch), so
// every wave still issues two pieces per image and the counted waits do not change.
constexpr unsigned OOB_OFFSET = 0xfffffff0u;
template <bool TR, int HR = 128>
VG_DEVICE void piece_offsets(unsigned (&voff)[2][2], long ld_bytes, int rc0, int wave, int lane) {
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = j * 8 + wave;
      if constexpr (!TR) {
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        voff[half][j] = (unsigned)((long)(rc0 + half * HR + row) * ld_bytes + chunk * 16);
        if (HR < 128 && row >= HR) voff[half][j] = OOB_OFFSET;
      } else {
        const int krow = piece * 4 + (lane >> 4);
        const int p16 = lane & 15;
        const int gran = (p16 >> 2) ^ (krow & 3);
        const int hf = ((p16 >> 1) & 1) ^ ((krow >> 3) & 1);
        const int col = gran * 32 + hf * 16 + (p16 & 1) * 8;
        voff[half][j] = (unsigned)((long)krow * ld_bytes + (long)(rc0 + half * HR + col) * 2);
        if (HR < 128 && col >= HR) voff[half][j] = OOB_OFFSET;
      }
    }
}

// Fragment addressing with the lane-dependent part hoisted: a 16x16x32 operand fragment of a half-tile image is
// base[variant] + compile-time immediate.
//  row image: row = r0 + (lane & 15), chunk (4 s + (lane >> 4)) ^ ((row >> 1) & 7): two bases (s = 0, 1)
//  k-major  : two ds_read_b64_tr_b16 at k-rows ka = 32 s + 8 g + q and ka + 4, columns c0 + 4 p (g = lane >> 4,
//             q = (lane & 15) >> 2, p = lane & 3): the granule swizzle depends on (c0 >> 5) & 3 and (c0 >> 4) & 1
template <bool TR>
VG_DEVICE bf16x8 frag_at(const char* img, int base, int imm) {
  if constexpr (!TR) {
    return *reinterpret_cast<const bf16x8*>(img + base + imm);
  } else {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, img + base + imm));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, img + base + imm + 1024));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

// One wave's fragment reader for its block of an operand half: NT 16-wide tiles starting at element r0 of the half.
template <bool TR, int NT>
struct BlockReader {
  int base[TR ? NT : 2];
  VG_DEVICE void init(int r0, int lane) {
    if constexpr (!TR) {
      const int row = r0 + (lane & 15);
#pragma unroll
      for (int s = 0; s < 2; ++s) base[s] = row * 128 + ((((4 * s + (lane >> 4)) ^ ((row >> 1) & 7))) << 4);
    } else {
      const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int c0 = r0 + i * 16;
        base[i] = (8 * g + q) * 256 + ((((c0 >> 5) & 3) ^ q) << 6) + ((((c0 >> 4) & 1) ^ (g & 1)) << 5) + 8 * p;
      }
    }
  }
  // fragment of tile i, k-step s (32 deep)
  VG_DEVICE bf16x8 get(const char* img, int i, int s) const {
    if constexpr (!TR) return frag_at<false>(img, base[s], i * 2048);
    else return frag_at<true>(img, base[i], s * 8192);
  }
};

// What a block needs to know about its tile: origin, K range, operand windows, per-lane piece offsets.
// The operand windows are kept as RUNNING scalars (base pointer and bytes left of K tile `tcur`, moved by advance()
// once per K tile): a request is then a descriptor of four ready words plus the M0 write -- recomputing the window
// from the tile index cost ~16 scalar instructions per request, eight requests per wave and K tile, inside segments
// whose length is what bounds the loop.
template <bool A_TR, bool B_TR, int BM_ = 256>
struct TileCtx {
  int m0, n0, wg, nwg, nkt;
  int a_step, b_step;          // bytes from one K tile to the next
  int a_left, b_left;          // bytes from the window base of K tile `tcur` to the end of the operand (may go <= 0)
  const char* a_cur;
  const char* b_cur;
  unsigned va[2][2], vb[2][2];
  // Round 6 (EPI_DACT8 on the long-phase 256 x 256 schedule): the tile's 8-bit stored derivative -- 256 rows x 256 bytes =
  // four 128-row x 128-byte images, exactly one K tile's worth of ring slots -- travels through the LDS-DMA ring as "K tile
  // nkt": the requests that would address the K tile past the end (zero fills of slots nobody reads) fetch it instead, two
  // K tiles before the main loop ends, and the epilogue reads it out of LDS.  Before, the epilogue requested it from global
  // memory at the end of the loop and waited out one round trip to (cold) HBM per tile: 4 rounds x ~4.7 us of the FFN-in
  // dgrad's 20 us over the plain product.  Image h = (row half h >> 1, byte-column half h & 1), row-image swizzle.
  const char* x_cur = nullptr;
  int x_left = 0;
  unsigned vx[2][2];
  VG_DEVICE void init_aux(const GemmParams& p, int wave, int lane) {
    const long ld = p.ldc;                                   // one byte per element
    x_cur = reinterpret_cast<const char*>(p.aux_in) + n0;
    x_left = (int)((long)(p.M - 1) * ld + p.N - n0);           // rows past M fall outside: zero fill
    piece_offsets<false>(vx, ld, m0, wave, lane);
  }
  VG_DEVICE void request_aux(int h, char* dst) const {
    const int ch = (h & 1) * 128;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x_cur + ch), 0, max(x_left - ch, 0), 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst), 16, vx[h >> 1][0], 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst + 8 * 1024), 16, vx[h >> 1][1], 0, 0, 0);
  }

  // `tile` = this block's tile index before the XCD-aware remap, `z` = its K slice
  VG_DEVICE void init(const GemmParams& p, int tile, int z, int wave, int lane) {
    const int kbeg = z * p.k_per_split;
    init_range(p, tile, kbeg, (min(p.K, kbeg + p.k_per_split) - kbeg) / BK, 1, wave, lane);
  }
  // K tiles [kbeg / 64, kbeg / 64 + ntiles) of output tile `tile`; order 1: apply the XCD-aware tile order of a plain
  // launch, 0: tile = mt * ntn + nt, 2: the same line walked along the shorter of the two tile dimensions
  VG_DEVICE void init_range(const GemmParams& p, int tile, int kbeg, int ntiles, int order, int wave, int lane) {
    const bool remap = order == 1;
    constexpr int BM = BM_, BN = 256;
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    nwg = ntn * ntm;
    // blocks that share an XCD get a contiguous run of tiles (bijective remap) ...
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = tile & 7;
    wg = remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3) : tile;
    int mt = wg / ntn, nt = wg % ntn;
    if (order == 2 && ntm < ntn) {
      nt = wg / ntm;
      mt = wg - nt * ntm;
    }
    if (remap && p.group_m > 0) {    // ... walked m-fastest inside bands of group_m row-tiles (fewer distinct operand panels per XCD)
      const int gsz = p.group_m * ntn, gid = wg / gsz, first = gid * p.group_m;
      const int gm = min(ntm - first, p.group_m), rem = wg - gid * gsz;
      mt = first + rem % gm;
      nt = rem / gm;
    }
#ifdef VG_LAB_SAMETILE               // lab: every block reads (and writes) tile 0 -- a full chip with no L2 misses
    mt = VG_LAB_SAMETILE == 2 ? mt & 7 : 0;
    nt = 0;
#endif
    m0 = mt * BM;
    n0 = nt * BN;
    nkt = ntiles;                      // whole K tiles only (the host checks)
    // The descriptor base advances with the K tile and its size shrinks by the same amount, so the hardware range
    // check stays exact (rows past M / N and k-rows past K read zeros) and the per-lane offsets never change.
    // The host guarantees that both operands span less than 2^31 bytes.
    const long lda_b = p.lda * 2, ldb_b = p.ldb * 2;
    const long a_bytes = A_TR ? (long)(p.K - 1) * lda_b + (long)p.M * 2 : (long)(p.M - 1) * lda_b + (long)p.K * 2;
    const long b_bytes = B_TR ? (long)(p.K - 1) * ldb_b + (long)p.N * 2 : (long)(p.N - 1) * ldb_b + (long)p.K * 2;
    a_step = (int)(A_TR ? (long)BK * lda_b : (long)BK * 2);
    b_step = (int)(B_TR ? (long)BK * ldb_b : (long)BK * 2);
    const long a_first = A_TR ? (long)kbeg * lda_b : (long)kbeg * 2, b_first = B_TR ? (long)kbeg * ldb_b : (long)kbeg * 2;
    a_cur = reinterpret_cast<const char*>(p.A) + a_first;
    b_cur = reinterpret_cast<const char*>(p.B) + b_first;
    a_left = (int)(a_bytes - a_first);
    b_left = (int)(b_bytes - b_first);
    piece_offsets<A_TR, BM_ / 2>(va, lda_b, m0, wave, lane);
    piece_offsets<B_TR>(vb, ldb_b, n0, wave, lane);
  }
  // move the windows by `n` K tiles
  VG_DEVICE void advance(int n = 1) {
    a_cur += (long)n * a_step;
    b_cur += (long)n * b_step;
    a_left -= n * a_step;
    b_left -= n * b_step;
  }
  // image h of the K tile `back` tiles before the current windows -> `dst` (this wave's first piece of the slot)
  VG_DEVICE void request(int h, char* dst, int back = 0) const {
    const bool is_a = h & 1;
    const int half = h >> 1;
    const int step = is_a ? a_step : b_step;
    const char* base = (is_a ? a_cur : b_cur) - (long)back * step;
    const int left = (is_a ? a_left : b_left) + back * step;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, max(left, 0), 0x00020000);
    const unsigned o0 = is_a ? va[half][0] : vb[half][0], o1 = is_a ? va[half][1] : vb[half][1];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst), 16, o0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst + 8 * 1024), 16, o1, 0, 0, 0);
  }
};

VG_DEVICE bool ph_aux_lds_enabled(const GemmParams& p) { return p.aux_ring != 0; }

using P0 = std::integral_constant<int, 0>;
using P1 = std::integral_constant<int, 1>;
using P2 = std::integral_constant<int, 2>;
using P3 = std::integral_constant<int, 3>;

VG_DEVICE void phase_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------
// Complementary schedule (tile_cfg 12): four phases of 16 MFMAs per K tile, ONE barrier per phase, 8 slots.
//   P0 reads A-half 0 + B-half 0, MFMA a0 x b0;  P1 reads B-half 1, a0 x b1;  P2 reads A-half 1, a1 x b1;
//   P3 reads nothing, a1 x b0 (b0 is still in registers).
// The two wave groups run the same phase in the same barrier interval but in opposite order: group X (wr = 0)
// issues MFMA_k first and then reads the fragments of phase k + 1, group Y (wr = 1) reads the fragments of phase k
// first and then issues MFMA_k: the matrix pipe of a SIMD passes from its X wave to its Y wave in mid-interval.
//   interval k: both groups request image k + 7 in their read segment and wait (vmcnt(8): all but the 4 youngest
//   images) before the closing barrier for what is read in interval k + 1 -- X's fragments of phase k + 2, Y's of
//   phase k + 1, i.e. images <= k + 3.  Image (t, h) lives in slot (t & 1, h); its successor is requested >= 1
//   interval after the last interval that reads the slot, behind the barrier every reader has passed.
template <bool A_TR, bool B_TR>
VG_DEVICE void px_main_loop(TileCtx<A_TR, B_TR>& c, f32x4 (&acc)[8][4], char* smem, int wave, int lane) {
  const int wr = wave >> 2, wc = wave & 3;
  BlockReader<A_TR, 4> rda;
  BlockReader<B_TR, 2> rdb;
  rda.init(wr * 64, lane);
  rdb.init(wc * 32, lane);
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  const int nkt = c.nkt;
  // the windows of `c` stand at K tile t + 2 inside the loop (`back` = 1 reaches K tile t + 1)
  auto request = [&](int t, int h, int back) { c.request(h, smem + (t & 1) * BUF_BYTES + h * HALF_BYTES + wave * 1024, back); };
  auto reads = [&](auto phc, const char* buf) {
    constexpr int PH = decltype(phc)::value;
    if constexpr (PH == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) fb0[j][s] = rdb.get(buf + 0 * HALF_BYTES, j, s);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(buf + 1 * HALF_BYTES, i, s);
    } else if constexpr (PH == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) fb1[j][s] = rdb.get(buf + 2 * HALF_BYTES, j, s);
    } else if constexpr (PH == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(buf + 3 * HALF_BYTES, i, s);
    }
  };
  auto mfmas = [&](auto phc) {
    constexpr int PH = decltype(phc)::value;
    constexpr int A0 = (PH >= 2) ? 4 : 0, B0 = (PH == 1 || PH == 2) ? 2 : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 b = (B0 == 0) ? fb0[j][s] : fb1[j][s];
          acc[A0 + i][B0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s], b, acc[A0 + i][B0 + j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  // request of interval k = 4 t + PH: image k + 7 = (K tile t + 1, h = 3) for PH = 0, (t + 2, h = PH - 1) else
  auto req = [&](auto phc, int t) {
    constexpr int PH = decltype(phc)::value;
    if constexpr (PH == 0) request(t + 1, 3, 1);
    else request(t + 2, PH - 1, 0);
  };
  auto close = [&]() {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    phase_barrier();
  };

  // ---- prologue: images 0..6; images 0 and 1 visible at the first barrier, image 2 at the second
#pragma unroll
  for (int s = 0; s < 4; ++s) request(0, s, 0);
  c.advance();
#pragma unroll
  for (int s = 0; s < 3; ++s) request(1, s, 0);
  c.advance();                     // windows at K tile 2 = t + 2 for the first loop iteration
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  phase_barrier();
  if (wr == 0) {
    // ------------------------------------------------ group X: MFMA_k, then the fragments of phase k + 1
    reads(P0{}, smem);
    close();
    for (int t = 0; t < nkt; ++t) {
      const char* buf = smem + (t & 1) * BUF_BYTES;
      const char* nxt = smem + ((t + 1) & 1) * BUF_BYTES;
      mfmas(P0{}); __builtin_amdgcn_sched_barrier(0); reads(P1{}, buf); req(P0{}, t); close();
      mfmas(P1{}); __builtin_amdgcn_sched_barrier(0); reads(P2{}, buf); req(P1{}, t); close();
      mfmas(P2{}); __builtin_amdgcn_sched_barrier(0); req(P2{}, t); close();
      mfmas(P3{}); __builtin_amdgcn_sched_barrier(0); reads(P0{}, nxt); req(P3{}, t); close();
      c.advance();
    }
  } else {
    // ------------------------------------------------ group Y: the fragments of phase k, then MFMA_k
    close();
    for (int t = 0; t < nkt; ++t) {
      const char* buf = smem + (t & 1) * BUF_BYTES;
      reads(P0{}, buf); req(P0{}, t); __builtin_amdgcn_sched_barrier(0); mfmas(P0{}); close();
      reads(P1{}, buf); req(P1{}, t); __builtin_amdgcn_sched_barrier(0); mfmas(P1{}); close();
      reads(P2{}, buf); req(P2{}, t); __builtin_amdgcn_sched_barrier(0); mfmas(P2{}); close();
      req(P3{}, t); __builtin_amdgcn_sched_barrier(0); mfmas(P3{}); close();
      c.advance();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero fills past the end must not land in the strips
}

// ---------------------------------------------------------------------------------------------------------
// Ring schedule (tile_cfg 11): TWO phases of 32 MFMAs per K tile, two barriers per phase with the second wave
// group one barrier behind (it reads while the first computes and vice versa), and a 10-slot image ring (all
// 160 KB of LDS) so that the images still travel 2-3 phases ahead:
//   phase A  reads b0, b1, a0 (16 fragments), MFMA a0 x b0, a0 x b1;   phase B  reads a1 (8), MFMA a1 x b1, a1 x b0
//   image s = 4 t + h sits in slot s mod 10; phase g requests images 2 g + 8 and 2 g + 9 at the START of its MFMA
//   segment -- one phase after the last fragment read of the images they replace (2 g - 2, 2 g - 1), which every
//   wave has completed before the barrier in front of that segment -- and each read segment ends with a counted
//   wait that retires what the NEXT phase reads (vmcnt(8) after phase A, vmcnt(6) after phase B).
// With k-major operands every fragment is two transposing LDS reads: their read segments are twice as long as a
// row image's, and keeping the LDS-DMA requests out of them is what makes this schedule the faster one there.
template <bool A_TR, bool B_TR>
VG_DEVICE void ring_main_loop(TileCtx<A_TR, B_TR>& c, f32x4 (&acc)[8][4], char* smem, int wave, int lane) {
  constexpr int NSLOT = 10;
  const int wr = wave >> 2, wc = wave & 3;
  BlockReader<A_TR, 4> rda;
  BlockReader<B_TR, 2> rdb;
  rda.init(wr * 64, lane);
  rdb.init(wc * 32, lane);
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  const int nkt = c.nkt;
  // requests always address the K tile the windows of `c` stand at (t + 2 inside the loop)
  auto request = [&](int, int h, int slot) { c.request(h, smem + slot * HALF_BYTES + wave * 1024); };
  auto wrap = [](int s) { return s >= NSLOT ? s - NSLOT : s; };

  // ---- prologue: images 0..7 (K tiles 0 and 1); phase A of tile 0 reads images 0, 1, 2
#pragma unroll
  for (int s = 0; s < 4; ++s) request(0, s, s);
  c.advance();
#pragma unroll
  for (int s = 0; s < 4; ++s) request(1, s, 4 + s);
  c.advance();
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  phase_barrier();
  if (wr == 1) phase_barrier();    // second wave group runs half a phase behind the first

  int rs0 = 0;                     // ring slot of image 4 t (B half 0 of the current K tile)
  int ws0 = 8;                     // ring slot of the next image to request (image 2 g + 8)
  for (int t = 0; t < nkt; ++t) {
    const int s0 = rs0, s1 = wrap(rs0 + 1), s2 = wrap(rs0 + 2), s3 = wrap(rs0 + 3);
    // ---------------- phase A (g = 2 t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) fb0[j][s] = rdb.get(smem + s0 * HALF_BYTES, j, s);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(smem + s1 * HALF_BYTES, i, s);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) fb1[j][s] = rdb.get(smem + s2 * HALF_BYTES, j, s);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    phase_barrier();
    request(t + 2, 0, ws0);                     // images 4 t + 8, 4 t + 9
    request(t + 2, 1, wrap(ws0 + 1));
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][2 * b + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s], b == 0 ? fb0[j][s] : fb1[j][s],
                                                                        acc[i][2 * b + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    phase_barrier();
    // ---------------- phase B (g = 2 t + 1)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(smem + s3 * HALF_BYTES, i, s);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    phase_barrier();
    request(t + 2, 2, wrap(ws0 + 2));           // images 4 t + 10, 4 t + 11
    request(t + 2, 3, wrap(ws0 + 3));
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int b = 1; b >= 0; --b)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[4 + i][2 * b + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s], b == 0 ? fb0[j][s] : fb1[j][s],
                                                                            acc[4 + i][2 * b + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    phase_barrier();
    rs0 = wrap(rs0 + 4);
    ws0 = wrap(ws0 + 4);
    c.advance();
  }
  if (wr == 0) phase_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero fills past the end must not land in the strips
}

// ---------------------------------------------------------------------------------------------------------
// Complementary schedule with long phases (tile_cfg 13): the opposite-order wave groups and the single barrier per
// phase of px_main_loop, but TWO phases of 32 MFMAs per K tile (a barrier every 2 x 512 MFMA cycles instead of
// every 2 x 256) on the 10-slot image ring of ring_main_loop:
//   phase A  reads b (the wave's B half), a0 (16 fragments), MFMA a0 x b;   phase B  reads a1 (8), MFMA a1 x b
//   interval k (one phase): group X issues MFMA_k, then reads the fragments of phase k + 1; group Y reads the
//   fragments of phase k, then issues MFMA_k.  Both request images 2 k + 8 and 2 k + 9 in their read segment -- the
//   slots they take (images 2 k - 2, 2 k - 1) were last read by Y in interval k - 1, behind the barrier -- and wait
//   before the closing barrier for what interval k + 1 reads: images <= 2 k + 6 after a phase A (vmcnt(6)),
//   <= 2 k + 5 after a phase B (vmcnt(8)).
// Order inside a read segment (measured, same call): fragment reads first, LDS-DMA requests after them.  With the
// requests in front of the reads (either group) the k-major modes lose 12-35 % (dgrad->model 104 -> 119-127 us, the
// grouped weight gradients 350 -> 448-472 us; their fragments are two ds_read_b64_tr_b16 each and sit on the critical
// path of the next MFMA run) while the row-image mode does not move; requests after Y's MFMAs: 3-17 % slower;
// without s_setprio around the MFMAs: no difference.
// NTA = 16-row A tiles a wave owns in each row half: 4 (256-row tile) or 3 (192-row tile, tile_cfg 15: 24 MFMAs per
// phase instead of 32 against the same B fragments and the same barrier / request structure).
template <bool A_TR, bool B_TR, int NTA = 4, bool AUXL = false>
VG_DEVICE void px2_main_loop(TileCtx<A_TR, B_TR, 64 * NTA>& c, f32x4 (&acc)[2 * NTA][4], char* smem, int wave, int lane) {
  constexpr int NSLOT = 10;
  constexpr bool UNROLL5 = !A_TR && !B_TR;
  const int wr = wave >> 2, wc = wave & 3;
  // this schedule gives a wave 64 CONTIGUOUS columns (all of them inside one B half image): its rows of C are whole
  // 128-byte lines for the epilogue's stores and for the residual / stored-derivative reads
  const int bh = (wc >> 1) << 1;   // slot offset of this wave's B half inside a K tile's four images (0 or 2)
  BlockReader<A_TR, NTA> rda;
  BlockReader<B_TR, 4> rdb;
  rda.init(wr * 16 * NTA, lane);
  rdb.init((wc & 1) * 64, lane);
  bf16x8 fa[NTA][2], fb[4][2];
  const int nkt = c.nkt;
  auto wrap = [](int s) { return s >= NSLOT ? s - NSLOT : s; };
  // requests always address the K tile the windows of `c` stand at (t + 2 inside the loop)
  auto request = [&](int, int h, int slot) { c.request(h, smem + slot * HALF_BYTES + wave * 1024); };
  // fragments of phase A (images in slots s0 = B half 0, s0 + 1 = A half 0, s0 + 2 = B half 1) / phase B (s0 + 3)
  auto reads_a = [&](int s0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) fb[j][s] = rdb.get(smem + wrap(s0 + bh) * HALF_BYTES, j, s);
#pragma unroll
    for (int i = 0; i < NTA; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(smem + wrap(s0 + 1) * HALF_BYTES, i, s);
  };
  auto reads_b = [&](int s0) {
#pragma unroll
    for (int i = 0; i < NTA; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(smem + wrap(s0 + 3) * HALF_BYTES, i, s);
  };
  auto mfmas = [&](auto halfc) {
    constexpr int A0 = decltype(halfc)::value * NTA;
    __builtin_amdgcn_s_setprio(1);
    // (round 4, timing-only lab: the phase's 32 MFMAs 16x16x32 issued as 16 MFMAs 32x32x16 on the same fragment registers --
    // half the matrix instructions for the same bytes read -- ran the K = 4096 / N = 1024 product's loop in 91.6-92.8 us
    // against 85.3-86.9, NN 95.3-95.4 against 93.6-94.6: the shorter instruction is the better fit for this schedule)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < NTA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[A0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s], fb[j][s], acc[A0 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // requests of interval k: images 2 k + 8, 2 k + 9 = K tile t + 2, h = 0, 1 (phase A) / 2, 3 (phase B); ws0 = slot of 4 t + 8
  auto req = [&](auto halfc, int t, int ws0) {
    constexpr int H0 = decltype(halfc)::value * 2;
    if (AUXL && t + 2 == nkt) {          // (wave-uniform) the "K tile" past the end is the epilogue's 8-bit operand tile
      c.request_aux(H0, smem + wrap(ws0 + H0) * HALF_BYTES + wave * 1024);
      c.request_aux(H0 + 1, smem + wrap(ws0 + H0 + 1) * HALF_BYTES + wave * 1024);
      return;
    }
    request(t + 2, H0, wrap(ws0 + H0));
    request(t + 2, H0 + 1, wrap(ws0 + H0 + 1));
  };
  auto close = [&](auto halfc) {
    if constexpr (decltype(halfc)::value == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    phase_barrier();
  };

  // ---- prologue: images 0..7 (K tiles 0 and 1); images 0..2 visible at the first barrier, image 3 at the second
#pragma unroll
  for (int s = 0; s < 4; ++s) request(0, s, s);
  c.advance();
#pragma unroll
  for (int s = 0; s < 4; ++s) request(1, s, 4 + s);
  c.advance();
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  phase_barrier();
  int rs0 = 0, ws0 = 8;            // ring slots of image 4 t and of image 4 t + 8
  if (wr == 0) {
    // ------------------------------------------------ group X: MFMA_k, then the fragments of phase k + 1
    reads_a(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    phase_barrier();
    int t = 0;
    // 4 images per K tile on a 10-slot ring: the slot pattern repeats every 5 K tiles, so with five K tiles unrolled
    // every ring slot is a compile-time constant (no compare / select / shift / add per request and per read
    // segment).  Row-image operands only: measured +1-2 % there (FFN-out forward 110.2 -> 108.2 us), -7 % with a
    // k-major B (dgrad->model 105.3 -> 112.4 us) and spills with both operands k-major.
    for (; UNROLL5 && t + 5 <= nkt; t += 5) {
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int r = (4 * u) % NSLOT, w = (8 + 4 * u) % NSLOT;
        mfmas(P0{}); __builtin_amdgcn_sched_barrier(0); reads_b(r); req(P0{}, t, w); close(P0{});
        mfmas(P1{}); __builtin_amdgcn_sched_barrier(0); reads_a(wrap(r + 4)); req(P1{}, t, w); close(P1{});
        c.advance();
      }
    }
    for (; t < nkt; ++t) {
      mfmas(P0{}); __builtin_amdgcn_sched_barrier(0); reads_b(rs0); req(P0{}, t, ws0); close(P0{});
      mfmas(P1{}); __builtin_amdgcn_sched_barrier(0); reads_a(wrap(rs0 + 4)); req(P1{}, t, ws0); close(P1{});
      rs0 = wrap(rs0 + 4);
      ws0 = wrap(ws0 + 4);
      c.advance();
    }
  } else {
    // ------------------------------------------------ group Y: the fragments of phase k, then MFMA_k
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    phase_barrier();
    int t = 0;
    for (; UNROLL5 && t + 5 <= nkt; t += 5) {
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int r = (4 * u) % NSLOT, w = (8 + 4 * u) % NSLOT;
        reads_a(r); req(P0{}, t, w); __builtin_amdgcn_sched_barrier(0); mfmas(P0{}); close(P0{});
        reads_b(r); req(P1{}, t, w); __builtin_amdgcn_sched_barrier(0); mfmas(P1{}); close(P1{});
        c.advance();
      }
    }
    for (; t < nkt; ++t) {
      reads_a(rs0); req(P0{}, t, ws0); __builtin_amdgcn_sched_barrier(0); mfmas(P0{}); close(P0{});
      reads_b(rs0); req(P1{}, t, ws0); __builtin_amdgcn_sched_barrier(0); mfmas(P1{}); close(P1{});
      rs0 = wrap(rs0 + 4);
      ws0 = wrap(ws0 + 4);
      c.advance();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero fills past the end must not land in the strips
}

template <int TM>
VG_DEVICE void zero_acc(f32x4 (&acc)[TM][4]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// SCHED: 0 = complementary (px), 1 = ring, 2 = complementary with long phases (px2)
#ifdef VG_LAB_STAMPS
__device__ long long* g_lab_stamps = nullptr;     // diagnostic build only (tools/lab/variant.sh ... -DVG_LAB_STAMPS)
#endif

template <bool A_TR, bool B_TR, int SCHED, int EPI = EPI_GENERIC, int BM = 256>
__global__ __launch_bounds__(512) void gemm_ph_kernel(GemmParams p) {
  static_assert(BM == 256 || (BM == 192 && SCHED == 2), "the 192-row tile exists on the long-phase schedule only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef VG_LAB_STAMPS
  long long st0 = wall_clock64();
#endif
#ifdef VG_LAB_XCDMASK                 // lab: only the blocks of some XCDs work (timing experiment: the others exit)
  if (((VG_LAB_XCDMASK) >> (blockIdx.x & 7) & 1) == 0) return;
#endif
  TileCtx<A_TR, B_TR, BM> c;
  c.init(p, blockIdx.x, blockIdx.z, wave, lane);
  // (Round 4, measured and rejected: running half of the first round's blocks on half their K tiles so that the CUs fall
  // out of step and their epilogues no longer meet at the HBM -- with the removed work paid back by extra blocks, so that
  // the makespan in tile units stays the same -- made the FFN-in products 15-19 % SLOWER, 134.7 -> 160.3 us plain and
  // 183.5 -> 218.6 us with GELU + stored derivative at M = 16384: the blocks of an XCD read the same K slice of a few
  // operand panels at the same time, and that lockstep is what keeps the panels L2 hits.  De-phasing whole XCDs instead
  // (the first-round blocks of XCDs 4..7 started 5 / 10 / 15 us late, lockstep inside each XCD intact) gained nothing
  // either: plain 127 -> 130 / 134 / 144 us, GELU + stored derivative 165 -> 176 us.
  // Also measured and rejected: a PERSISTENT form of this kernel for launches of several rounds (one block per CU walks
  // its tiles, requests the next tile's first eight images right after the barrier that ends the main loop -- before
  // it writes the finished tile out -- and transposes through swizzled 32 KB strips in ring slots 8 and 9; bitwise
  // equal results).  In-kernel stamps had priced a tile of the K = 1024 products at 2.2 us from block entry to a running
  // pipeline + 16 x 1.3 us of K tiles + 4.3 us of epilogue + 0.5-1.5 us until the CU's next block enters (5 us after the
  // GELU epilogue), but hiding the first and the last of those changed nothing: FFN-in forward 127-129 -> 123-126 us on
  // cold operands, GELU 167-168 -> 166, QKV 94 -> 93-96, the training step 543-547 k -> 544-545 k tokens/s.  The
  // dispatcher already starts a CU's next block while the last one's stores drain, and the counted waits of the new
  // tile's prologue sit behind the finished tile's stores (vector-memory operations retire in order).)
  f32x4 acc[BM / 32][4];           // [a * 4 + i][b * 2 + j]; SCHED 2: [a * (BM / 64) + i][j]
  zero_acc(acc);
  // the 8-bit derivative through the ring (see TileCtx::init_aux): the k-major-B dgrad on 256-row tiles with >= 2 K tiles
  constexpr bool AUXL = SCHED == 2 && EPI == EPI_DACT8 && BM == 256 && !A_TR && B_TR;
  bool aux_lds = false;
  if constexpr (AUXL) {
    aux_lds = c.nkt >= 2 && ph_aux_lds_enabled(p);
    if (aux_lds) c.init_aux(p, wave, lane);
  }
  if constexpr (SCHED == 1) ring_main_loop<A_TR, B_TR>(c, acc, smem, wave, lane);
  else if constexpr (SCHED == 2) {
    if constexpr (AUXL) {
      if (aux_lds) px2_main_loop<A_TR, B_TR, BM / 64, true>(c, acc, smem, wave, lane);
      else px2_main_loop<A_TR, B_TR, BM / 64, false>(c, acc, smem, wave, lane);
    } else {
      px2_main_loop<A_TR, B_TR, BM / 64>(c, acc, smem, wave, lane);
    }
  }
  else px_main_loop<A_TR, B_TR>(c, acc, smem, wave, lane);
#ifdef VG_LAB_STAMPS
  long long st1 = wall_clock64();
#endif
  constexpr bool ILVC = SCHED != 2;
  if constexpr (EPI == EPI_GENERIC) tile_epilogue<BM, 256, 2, 4, true, ILVC>(p, acc, smem, c.m0, c.n0, c.wg, c.nwg);
  else if constexpr (AUXL) {
    if (aux_lds) {
      // images 4 nkt + h of the ring sit in slots (4 nkt + h) mod 10; the strips (35 KB) go where those four are not
      const int s0 = (4 * c.nkt) % 10;
      const int strip_slot = s0 == 0 ? 4 : (s0 == 8 ? 2 : (s0 == 2 ? 6 : 0));
      tile_epilogue_lean<BM, 256, 2, 4, true, ILVC, EPI>(p, acc, smem + strip_slot * HALF_BYTES, c.m0, c.n0, smem, s0);
    } else {
      tile_epilogue_lean<BM, 256, 2, 4, true, ILVC, EPI>(p, acc, smem, c.m0, c.n0);
    }
  }
  else tile_epilogue_lean<BM, 256, 2, 4, true, ILVC, EPI>(p, acc, smem, c.m0, c.n0);
#ifdef VG_LAB_STAMPS
  if (g_lab_stamps && tid == 0 && blockIdx.x < 4096) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long* o = g_lab_stamps + blockIdx.x * 8;
    o[0] = st0; o[1] = st1; o[2] = wall_clock64();
    o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Two blocks per CU (tile_cfg 14): 256x128x64 tile, 4 waves as 2 x 2 (one per SIMD), 80 KB of LDS, <= 256 VGPRs.
// Why.  With one 8-wave block per CU nothing runs on the matrix pipe while that block fills its ring (2-3 us) or
// writes its tile out (9 us plain, ~20 us with the GELU epilogue): at K = 1024 (22 us of main loop per tile) that is
// half of the launch.  Two INDEPENDENT blocks per CU -- each SIMD holds one wave of each -- fall out of step by
// themselves: one block's prologue, epilogue VALU work and store drain run under the other's MFMAs, and inside the
// main loops one wave's fragment reads and LDS-DMA issue run under its partner's MFMAs (what the complementary
// schedules arrange by hand between the two wave groups of one block).
//   images   3 per K tile, 16 KB each: B (128 columns), A half 0, A half 1 (rows 0..127 / 128..255 of the tile);
//            image s = 3 t + h lives in slot s mod 5.  Wave (wr, wc) owns rows {wr*64 .. +63} of BOTH A halves and
//            columns {wc*64 .. +63}: phase A reads B and A half 0 (32 MFMAs), phase B reads A half 1 (32 MFMAs, the B
//            fragments stay in registers), so an image is dead after its one reading phase.
//   phase    s_waitcnt vmcnt(8) (all but the two youngest images), ONE s_barrier, the requests whose slots that barrier
//            freed (phase A: image 3 t + 4; phase B: 3 t + 5 and 3 t + 6), fragment reads, MFMAs.
//   tails    as above: requests past the last K tile are zero fills of slots nobody reads.
template <bool TR>
VG_DEVICE void duo_piece_offsets(unsigned (&voff)[4], long ld_bytes, int rc0, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int piece = j * 4 + wave;
    if constexpr (!TR) {
      const int row = piece * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      voff[j] = (unsigned)((long)(rc0 + row) * ld_bytes + chunk * 16);
    } else {
      const int krow = piece * 4 + (lane >> 4);
      const int p16 = lane & 15;
      const int gran = (p16 >> 2) ^ (krow & 3);
      const int hf = ((p16 >> 1) & 1) ^ ((krow >> 3) & 1);
      const int col = gran * 32 + hf * 16 + (p16 & 1) * 8;
      voff[j] = (unsigned)((long)krow * ld_bytes + (long)(rc0 + col) * 2);
    }
  }
}

template <bool A_TR, bool B_TR>
struct DuoCtx {
  int m0, n0, wg, nwg, nkt;
  int a_step, b_step;          // bytes from one K tile to the next
  int a_left, b_left;          // bytes from the current window base to the end of the operand (may go <= 0)
  int a_half;                  // bytes from A half 0 to A half 1
  const char* a_cur;
  const char* b_cur;
  unsigned va[4], vb[4];

  VG_DEVICE void init(const GemmParams& p, int tile, int z, int wave, int lane) {
    constexpr int BM = 256, BN = 128;
    const int kbeg = z * p.k_per_split;
    nkt = (min(p.K, kbeg + p.k_per_split) - kbeg) / BK;
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    nwg = ntn * ntm;
    // blocks that share an XCD get a contiguous run of tiles, walked m-fastest inside bands of group_m row-tiles
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = tile & 7;
    wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
    int mt = wg / ntn, nt = wg % ntn;
    if (p.group_m > 0) {
      const int gsz = p.group_m * ntn, gid = wg / gsz, first = gid * p.group_m;
      const int gm = min(ntm - first, p.group_m), rem = wg - gid * gsz;
      mt = first + rem % gm;
      nt = rem / gm;
    }
    m0 = mt * BM;
    n0 = nt * BN;
    const long lda_b = p.lda * 2, ldb_b = p.ldb * 2;
    const long a_bytes = A_TR ? (long)(p.K - 1) * lda_b + (long)p.M * 2 : (long)(p.M - 1) * lda_b + (long)p.K * 2;
    const long b_bytes = B_TR ? (long)(p.K - 1) * ldb_b + (long)p.N * 2 : (long)(p.N - 1) * ldb_b + (long)p.K * 2;
    a_step = (int)(A_TR ? (long)BK * lda_b : (long)BK * 2);
    b_step = (int)(B_TR ? (long)BK * ldb_b : (long)BK * 2);
    a_half = (int)(A_TR ? 256L : 128L * lda_b);
    const long a_first = A_TR ? (long)kbeg * lda_b : (long)kbeg * 2, b_first = B_TR ? (long)kbeg * ldb_b : (long)kbeg * 2;
    a_cur = reinterpret_cast<const char*>(p.A) + a_first;
    b_cur = reinterpret_cast<const char*>(p.B) + b_first;
    a_left = (int)(a_bytes - a_first);
    b_left = (int)(b_bytes - b_first);
    duo_piece_offsets<A_TR>(va, lda_b, m0, wave, lane);
    duo_piece_offsets<B_TR>(vb, ldb_b, n0, wave, lane);
  }
  VG_DEVICE void advance() {
    a_cur += a_step;
    b_cur += b_step;
    a_left -= a_step;
    b_left -= b_step;
  }
  // image h (0 B, 1 A half 0, 2 A half 1) of the K tile `back` tiles before the current windows -> slot `dst`
  VG_DEVICE void request(int h, char* dst, int wave, int back = 0) const {
    const bool is_a = h != 0;
    const int shift = (is_a ? a_step : b_step) * back - (h == 2 ? a_half : 0);
    const char* base = (is_a ? a_cur : b_cur) - shift;
    const int left = (is_a ? a_left : b_left) + shift;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, max(left, 0), 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst + (j * 4 + wave) * 1024), 16, is_a ? va[j] : vb[j],
                                               0, 0, 0);
  }
};

template <bool A_TR, bool B_TR>
VG_DEVICE void duo_main_loop(DuoCtx<A_TR, B_TR>& c, f32x4 (&acc)[8][4], char* smem, int wave, int lane) {
  constexpr int NSLOT = 5;
  const int wr = wave >> 1, wc = wave & 1;
  BlockReader<A_TR, 4> rda;
  BlockReader<B_TR, 4> rdb;
  rda.init(wr * 64, lane);
  rdb.init(wc * 64, lane);
  bf16x8 fa[4][2], fb[4][2];
  auto wrap = [](int s) { return s >= NSLOT ? s - NSLOT : s; };
  auto slot = [&](int s) { return smem + s * HALF_BYTES; };
  auto mfmas = [&](auto halfc) {
    constexpr int A0 = decltype(halfc)::value * 4;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[A0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s], fb[j][s], acc[A0 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // prologue: images 0..3 (K tile 0 and the B image of K tile 1); the windows then stand at K tile t + 2
  c.request(0, slot(0), wave);
  c.request(1, slot(1), wave);
  c.request(2, slot(2), wave);
  c.advance();
  c.request(0, slot(3), wave);
  c.advance();
  int rs = 0;                      // slot of image 3 t
  const int nkt = c.nkt;
  for (int t = 0; t < nkt; ++t) {
    // ---- phase A: images 3 t (B), 3 t + 1 (A half 0)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    phase_barrier();
    c.request(1, slot(wrap(rs + 4)), wave, 1);                 // image 3 t + 4: A half 0 of K tile t + 1
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) fb[j][s] = rdb.get(slot(rs), j, s);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(slot(wrap(rs + 1)), i, s);
    mfmas(P0{});
    // ---- phase B: image 3 t + 2 (A half 1)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    phase_barrier();
    c.request(2, slot(rs), wave, 1);                           // image 3 t + 5: A half 1 of K tile t + 1
    c.request(0, slot(wrap(rs + 1)), wave, 0);                 // image 3 t + 6: B of K tile t + 2
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(slot(wrap(rs + 2)), i, s);
    mfmas(P1{});
    rs = wrap(rs + 3);
    c.advance();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero fills past the end must not land in the strips
}

template <bool A_TR, bool B_TR, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DuoCtx<A_TR, B_TR> c;
  c.init(p, blockIdx.x, blockIdx.z, wave, lane);
  f32x4 acc[8][4];                 // [A half * 4 + i][j]
  zero_acc(acc);
  duo_main_loop<A_TR, B_TR>(c, acc, smem, wave, lane);
  if constexpr (EPI == EPI_GENERIC) tile_epilogue<256, 128, 2, 2, true, false>(p, acc, smem, c.m0, c.n0, c.wg, c.nwg);
  else tile_epilogue_lean<256, 128, 2, 2, true, false, EPI>(p, acc, smem, c.m0, c.n0);
}

template <bool A_TR, bool B_TR, int EPI>
int launch_duo_epi(const GemmParams& p, int splits, hipStream_t stream) {
  constexpr size_t lds = 5 * HALF_BYTES;
  auto k = gemm_duo_kernel<A_TR, B_TR, EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int ntn = (p.N + 127) / 128, ntm = (p.M + 255) / 256;
  VG_LAUNCH(k, dim3(ntn * ntm, 1, splits), dim3(256), lds, stream, p);
  return 0;
}
template <bool A_TR, bool B_TR>
int launch_duo(const GemmParams& p, int splits, hipStream_t stream) {
  if constexpr (!A_TR) {
    switch (lean_epilogue_of(p, splits)) {
      case EPI_PLAIN: return launch_duo_epi<A_TR, B_TR, EPI_PLAIN>(p, splits, stream);
      case EPI_GELU_SAVE: return launch_duo_epi<A_TR, B_TR, EPI_GELU_SAVE>(p, splits, stream);
      case EPI_DACT: return launch_duo_epi<A_TR, B_TR, EPI_DACT>(p, splits, stream);
      default: break;
    }
  }
  return launch_duo_epi<A_TR, B_TR, EPI_GENERIC>(p, splits, stream);
}

// ---------------------------------------------------------------------------------------------------------
// Grouped launch (vg_gemm_grouped): up to VG_GROUP_MAX weight-gradient products in ONE persistent grid.
// Written for the weight gradients of a Transformer layer: dW1, dW2, dWqkv, dWo are 64 + 64 + 48 + 16 output tiles of
// 256x256 whose reduction runs over all M frames (250 K tiles at M = 16000).  Launched one by one each needs split-K
// to fill the chip (x4, x4, x5, x10), whose equal slices all reach their fp32 atomics together (67 MB for dW1 alone
// at the memory side's 1.3 TB/s); launched as whole tiles they are 192 equal blocks for 256 CUs.  Here every CU gets
// the same number of (tile, K tile) units; a segment that covers a tile's whole K adds to the gradient with plain
// 16-byte accesses, a partial one with fp32 atomics.
//
// Two ways of cutting the work:
//  * stream (plan.lockstep = 0): the units of all problems form one line that is cut into equal ranges; a block
//    walks its range, finishing one tile and starting the next.  Any mix of K extents.  The blocks of an XCD stand
//    at unrelated K offsets of unrelated tiles, so every block streams its own two operand panels from HBM
//    (measured: 2.08 GB fetched per layer for 0.53 GB of operands; 381 us).
//  * lockstep (all problems share K): tiles are dealt to XCDs in runs that are compact rectangles of the output
//    (walked along the problem's shorter tile dimension).  Whole rounds of one tile per block first; of the R < P
//    tiles left over, XCD x owns T_x and its 32 blocks split into T_x "head" blocks, which reduce K tiles
//    [0, kh) of one tile each, and 32 - T_x "tail" blocks, which share the K tiles [kh, nkt) of those same tiles
//    (kh = T_x nkt / 32 gives everyone the same number of units -- plus `epi` K tiles per epilogue a tail block ends
//    beyond the first; tail block j takes tiles j, j + n_tail, ... when that divides).  All head blocks start at K tile 0 together and all tail blocks at kh, so at any moment the
//    blocks of an XCD read the SAME K slice of a few dY / X panels: one HBM fetch serves a whole row / column of
//    the rectangle out of that XCD's L2 (6 x 4 heads + 2 x 4 tails: 16 panel slices per 32 units instead of 64;
//    measured 0.81 GB fetched per layer, 339 us).
//  * round 4: a last round that at least 7/8 of the blocks would take part in is run as a WHOLE round with the other
//    blocks idle (plan.rounds counts it, nothing is left over): its tiles keep their whole K range and add with plain
//    16-byte accesses -- 240 tiles split over 256 blocks would mean 30 heads + 2 tails per XCD, a plan the lockstep
//    rules refuse, i.e. the stream plan with two atomic epilogues per tile.
struct GroupPlan {
  int lockstep;
  int rounds;                       // whole rounds of one full-K tile per block
  int first[9];                     // XCD x owns leftover tiles [first[x], first[x + 1]) of the tile line
  int epi;                          // what one segment's epilogue costs a block, in K tiles (balances heads and tails)
};
// One product of the group as the kernel needs it (56 bytes: VG_GROUP_MAX of them fit the 4 KB kernel-argument block,
// which a full GemmParams per problem would not)
struct GroupProb {
  const void* A; const void* B; void* C;
  int M, N, K;
  int lda, ldb, ldc;
  int accumulate;      // 0: C holds zeros, whole-K tiles may store instead of read-modify-write
  int pad_;
};
struct GroupParams {
  GroupProb q[VG_GROUP_MAX];
  int unit0[VG_GROUP_MAX + 1];      // first work unit (one K tile of one output tile) of problem g; [n] = total
  int tile0[VG_GROUP_MAX + 1];      // first tile of problem g on the tile line
  int nkt[VG_GROUP_MAX];            // K tiles per output tile
  int n;
  GroupPlan plan;
};
static_assert(sizeof(GroupParams) <= 4096, "GroupParams must fit the kernel-argument block");

// the next segment of this block: problem g, tile (problem-local), first K tile, K tile count; false when done
struct GroupWalk {
  int stage, u, uend;               // stream: unit cursor; lockstep: stage = rounds done (+1 head done), tail cursor
  int xcd, slot, per;
  int kh, ktail, tx, ntail;
  VG_DEVICE void init(const GroupParams& gp) {
    const int P = gridDim.x;
    per = P >> 3;
    xcd = blockIdx.x & 7;
    slot = blockIdx.x >> 3;
    stage = 0;
    if (!gp.plan.lockstep) {
      const int r8 = P & 7;
      const int b = (xcd < r8 ? xcd * (per + 1) : r8 * (per + 1) + (xcd - r8) * per) + slot;
      const long total = gp.unit0[gp.n];
      u = (int)(total * b / P);
      uend = (int)(total * (b + 1) / P);
    } else {
      const int nkt = gp.nkt[0];
      tx = gp.plan.first[xcd + 1] - gp.plan.first[xcd];
      ntail = per - tx;
      kh = ntail > 0 ? min(nkt, (int)(((long)tx * nkt + (long)max(tx - ntail, 0) * gp.plan.epi) / per)) : nkt;
      ktail = nkt - kh;
      u = uend = 0;
      if (slot >= tx && ntail > 0) {
        const long U = (long)tx * ktail;
        const int j = slot - tx;
        u = (int)(U * j / ntail);
        uend = (int)(U * (j + 1) / ntail);
      }
    }
  }
  VG_DEVICE bool next(const GroupParams& gp, int& g, int& tile, int& kt, int& count) {
    int line;                       // index on the tile line (lockstep) / unit line (stream)
    if (!gp.plan.lockstep) {
      if (u >= uend) return false;
      g = 0;
      for (int i = 1; i < gp.n; ++i)
        if (u >= gp.unit0[i]) g = i;
      const int nkt = gp.nkt[g], local = u - gp.unit0[g];
      tile = local / nkt;
      kt = local - tile * nkt;
      count = min(nkt - kt, uend - u);
      u += count;
      return true;
    }
    if (stage < gp.plan.rounds) {
      line = stage * (per * 8) + xcd * per + slot;
      kt = 0;
      count = gp.nkt[0];
      ++stage;
      if (line >= gp.tile0[gp.n]) return false;     // a block that sits out the last (partly filled) round
    } else if (stage == gp.plan.rounds && slot < tx) {
      ++stage;
      if (kh == 0) return false;
      line = gp.plan.first[xcd] + slot;
      kt = 0;
      count = kh;
    } else {
      if (u >= uend) return false;
      const int q = u / ktail, off = u - q * ktail;
      count = min(ktail - off, uend - u);
      kt = kh + off;
      const int c = tx / ntail;
      line = gp.plan.first[xcd] + (tx % ntail == 0 ? (q % c) * ntail + q / c : q);
      u += count;
    }
    g = 0;
    for (int i = 1; i < gp.n; ++i)
      if (line >= gp.tile0[i]) g = i;
    tile = line - gp.tile0[g];
    return true;
  }
};

template <bool A_TR, bool B_TR>
__global__ __launch_bounds__(512) void gemm_ring_group_kernel(GroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  GroupWalk walk;
  walk.init(gp);
  int g, tile, kt, count;
  while (walk.next(gp, g, tile, kt, count)) {
    const GroupProb& q = gp.q[g];
    GemmParams p{};
    p.A = q.A; p.B = q.B; p.C = q.C;
    p.M = q.M; p.N = q.N; p.K = q.K;
    p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
    p.k_per_split = q.K;
    p.accumulate = q.accumulate;
    TileCtx<A_TR, B_TR> c;
    c.init_range(p, tile, kt * BK, count, gp.plan.lockstep ? 2 : 0, wave, lane);
    f32x4 acc[8][4];
    zero_acc(acc);
    px2_main_loop<A_TR, B_TR>(c, acc, smem, wave, lane);
    tile_epilogue_wgrad<256, 256, 2, 4, true, false>(p, acc, smem, c.m0, c.n0, count != gp.nkt[g]);
    __syncthreads();               // the strips are read out before the next segment's images land in them
  }
}

template <typename K>
void set_lds(K k, size_t lds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <bool A_TR, bool B_TR, int SCHED, int EPI = EPI_GENERIC, int BM = 256>
int launch_ph(const GemmParams& p, int splits, hipStream_t stream) {
  constexpr size_t lds = SCHED != 0 ? 10 * HALF_BYTES : 2 * BUF_BYTES;
  auto k = gemm_ph_kernel<A_TR, B_TR, SCHED, EPI, BM>;
  static bool attr_done = false;
  if (!attr_done) {
    set_lds(k, lds);
    attr_done = true;
  }
  const int ntn = (p.N + 255) / 256, ntm = (p.M + BM - 1) / BM;
  VG_LAUNCH(k, dim3(ntn * ntm, 1, splits), dim3(512), lds, stream, p);
  return 0;
}

template <bool A_TR, bool B_TR, int BM = 256>
int launch_px2(const GemmParams& p, int splits, hipStream_t stream) {
  if constexpr (!A_TR) {           // forward / dgrad products end in bf16 rows; the TN products are fp32 gradients
    switch (lean_epilogue_of(p, splits)) {
      case EPI_PLAIN: return launch_ph<A_TR, B_TR, 2, EPI_PLAIN, BM>(p, splits, stream);
      case EPI_GELU_SAVE: return launch_ph<A_TR, B_TR, 2, EPI_GELU_SAVE, BM>(p, splits, stream);
      case EPI_DACT: return launch_ph<A_TR, B_TR, 2, EPI_DACT, BM>(p, splits, stream);
      case EPI_SILU_SAVE: return launch_ph<A_TR, B_TR, 2, EPI_SILU_SAVE, BM>(p, splits, stream);
      case EPI_GELU_SAVE8: return launch_ph<A_TR, B_TR, 2, EPI_GELU_SAVE8, BM>(p, splits, stream);
      case EPI_DACT8: return launch_ph<A_TR, B_TR, 2, EPI_DACT8, BM>(p, splits, stream);
      default: break;
    }
  }
  return launch_ph<A_TR, B_TR, 2, EPI_GENERIC, BM>(p, splits, stream);
}

}  // namespace

namespace vg_host {
// phase-pipelined 256x256 tile; returns 0 if launched, -1 if this variant does not apply (caller falls back)
int gemm_ph_launch(const GemmParams& p, int a_tr, int b_tr, int cfg, int splits, hipStream_t stream) {
  if (a_tr && !b_tr) return -1;
  if (p.k_per_split % BK != 0 || p.K % BK != 0) return -1;       // whole K tiles in every split
  if (cfg == 12) {
    if (!a_tr && !b_tr) return launch_ph<false, false, 0>(p, splits, stream);
    if (!a_tr && b_tr) return launch_ph<false, true, 0>(p, splits, stream);
    return launch_ph<true, true, 0>(p, splits, stream);
  }
  if (cfg == 14) {
    if (!a_tr && !b_tr) return launch_duo<false, false>(p, splits, stream);
    if (!a_tr && b_tr) return launch_duo<false, true>(p, splits, stream);
    return launch_duo<true, true>(p, splits, stream);
  }
  if (cfg == 15) {                 // 192 x 256 tiles on the long-phase schedule (forward / dgrad products: A is a row image)
    if (a_tr) return -1;
    if (!b_tr) return launch_px2<false, false, 192>(p, splits, stream);
    return launch_px2<false, true, 192>(p, splits, stream);
  }
  if (cfg == 13) {
    if (!a_tr && !b_tr) return launch_px2<false, false>(p, splits, stream);
    if (!a_tr && b_tr) return launch_px2<false, true>(p, splits, stream);
    return launch_px2<true, true>(p, splits, stream);
  }
  if (!a_tr && !b_tr) return launch_ph<false, false, 1>(p, splits, stream);
  if (!a_tr && b_tr) return launch_ph<false, true, 1>(p, splits, stream);
  return launch_ph<true, true, 1>(p, splits, stream);
}

// grouped TN products (weight gradients); every problem: bf16, a_tr = b_tr = 1, whole K tiles, split_k = 1
int gemm_group_launch(const GemmParams* ps, const int* splits, int n, hipStream_t stream) {
  GroupParams gp;
  gp.n = n;
  long units = 0, tiles = 0;
  bool same_k = true;
  for (int i = 0; i < n; ++i) {
    if (splits[i] != 1) return -1;
    if (ps[i].lda > 0x7fffffffL || ps[i].ldb > 0x7fffffffL || ps[i].ldc > 0x7fffffffL) return -1;
    gp.q[i] = GroupProb{ps[i].A, ps[i].B, ps[i].C, ps[i].M, ps[i].N, ps[i].K, (int)ps[i].lda, (int)ps[i].ldb, (int)ps[i].ldc,
                        ps[i].accumulate, 0};
    gp.nkt[i] = ps[i].K / BK;
    gp.unit0[i] = (int)units;
    gp.tile0[i] = (int)tiles;
    same_k = same_k && gp.nkt[i] == gp.nkt[0];
    const long t = (long)((ps[i].N + 255) / 256) * ((ps[i].M + 255) / 256);
    tiles += t;
    units += t * gp.nkt[i];
    if (units > 0x3fffffffL) return -1;
  }
  for (int i = n; i <= VG_GROUP_MAX; ++i) {
    gp.unit0[i] = (int)units;
    gp.tile0[i] = (int)tiles;
  }
  // one block per CU; fewer when there is less than ~8 K tiles of work for each
  static const int cus = [] {
    int dev = 0, n_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    return n_cu > 0 ? n_cu : 256;
  }();
  const int blocks = (int)max(1L, min((long)cus, units / 8));
  // lockstep plan: all problems share K, a full grid, and tails that are not a long string of tiny segments
  static const int mode = [] {
    const char* e = getenv("VG_GROUP_PLAN");      // 0 = always stream, 1 = default
    return e ? atoi(e) : 1;
  }();
  gp.plan.lockstep = 0;
  gp.plan.rounds = 0;
  // a tail block ends tx / n_tail segments (each an atomic epilogue of ~25 us when 64 blocks end together), a head
  // block one: the heads take that many more K tiles.  Measured on the layer's four products (M = 16000): 0 -> 362 us,
  // 10 -> 349, 40 -> 339, 60 -> 361
  gp.plan.epi = 40;
  for (int x = 0; x <= 8; ++x) gp.plan.first[x] = 0;
  if (mode != 0 && same_k && blocks == cus && cus % 8 == 0 && n > 0) {
    const int per = cus / 8;
    int rounds = (int)(tiles / cus), left = (int)(tiles - (long)rounds * cus);
    if (left * 8 >= cus * 7) {        // a nearly full last round: whole tiles, the few other blocks sit it out
      ++rounds;
      left = 0;
    }
    bool ok = true;
    for (int x = 0; x <= 8; ++x) gp.plan.first[x] = left == 0 ? (int)tiles : rounds * cus + (int)((long)left * x / 8);
    for (int x = 0; x < 8 && ok; ++x) {
      const int tx = gp.plan.first[x + 1] - gp.plan.first[x], ntail = per - tx;
      // a tail block walks tx / ntail tile tails: keep that short (each ends in an atomic epilogue), and every
      // tail segment long enough to pay for its epilogue
      if (ntail > 0 && tx > 4 * ntail) ok = false;
      if (ntail > 0 && tx > 0 && (long)(gp.nkt[0] - (long)tx * gp.nkt[0] / per) < 8) ok = false;
    }
    if (ok) {
      gp.plan.lockstep = 1;
      gp.plan.rounds = rounds;
    }
  }
  constexpr size_t lds = 10 * HALF_BYTES;
  auto k = gemm_ring_group_kernel<true, true>;
  static bool attr_done = false;
  if (!attr_done) {
    set_lds(k, lds);
    attr_done = true;
  }
  VG_LAUNCH(k, dim3(blocks), dim3(512), lds, stream, gp);
  return 0;
}
}  // namespace vg_host

#ifdef VG_LAB_STAMPS
extern "C" int vg_lab_set_stamps(long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lab_stamps), &buf, sizeof(buf));
}
#endif
