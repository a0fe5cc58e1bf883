// bf16 MFMA GEMM, phase-pipelined 256x256 tile (vg_gemm tile_cfg 10 / 11): the main loop the large products of the
// training step run on.
//
// Why another main loop.  gemm_dma_kernel keeps ONE 64 KB K tile in flight, requested after the barrier that
// retires the previous one: its request burst, its landing and the 24 fragment reads of the tile all queue behind
// that barrier, and both waves of a SIMD reach their MFMA run together.  Here the K tile is cut into four 16 KB
// half-tile images that stream through an 8-slot ring at ONE image per phase, five phases (80 KB per CU) ahead of
// their first reader, and the two wave groups of the block run half a phase apart, so that on every SIMD one wave
// issues its 16 MFMAs while its partner reads fragments and issues the next image's LDS-DMA.
//
//   block   256 x 256 x 64, 8 waves = 2 (rows) x 4 (columns); wave (wr, wc) owns rows {wr*64 .. +63} of BOTH row halves
//           of the tile and columns {wc*32 .. +31} of BOTH column halves: every phase reads one specific half-tile,
//           which is what lets the images stream (an image is dead two phases after its single reading phase)
//   phases  P1 reads A-half 0 + B-half 0, MFMA a0 x b0;  P2 reads B-half 1, a0 x b1;  P3 reads A-half 1, a1 x b1;
//           P4 reads nothing, a1 x b0 (b0 is still in registers) -- 16 MFMAs (16x16x32) per wave and phase
//   stream  image s = 4 t + h (h: B0, A0, B1, A1 of K tile t) lives in slot (t & 1, h) and is requested in global
//           phase s - 6; its readers run in phase 4 t + {0, 0, 1, 2}.  After each request a wave waits until all but
//           its 8 youngest LDS-DMA instructions (4 images) have landed, which retires the image of the NEXT phase;
//           the barrier between the two makes that true for every wave's pieces (read one phase after the wait).
//   hazards an image is requested >= 2 phases after its slot's last reading phase, so with the half-phase stagger
//           every reader has passed the barrier behind its last fragment read before any wave can request.
// Same operand modes, LDS images, tails by range check (M, N; K must be a multiple of 64 per split) and epilogue
// contract as gemm_dma_kernel.
#include <type_traits>
#include "vg_gemm_tile.h"

namespace {

constexpr int HALF_BYTES = 16384;            // one half-tile image: [128 rows][64 k] row image or [64 k][128] k-major
constexpr int BUF_BYTES = 4 * HALF_BYTES;    // B0 A0 B1 A1 of one K tile

// per-lane source offsets of this wave's two 1-KiB pieces of a half-tile image (same images as dma_tile<..., 128, 8>)
template <bool TR>
VG_DEVICE void piece_offsets(unsigned (&voff)[2][2], long ld_bytes, int rc0, int wave, int lane) {
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = j * 8 + wave;
      if constexpr (!TR) {
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        voff[half][j] = (unsigned)((long)(rc0 + half * 128 + row) * ld_bytes + chunk * 16);
      } else {
        const int krow = piece * 4 + (lane >> 4);
        const int p16 = lane & 15;
        const int gran = (p16 >> 2) ^ (krow & 3);
        const int hf = ((p16 >> 1) & 1) ^ ((krow >> 3) & 1);
        const int col = gran * 32 + hf * 16 + (p16 & 1) * 8;
        voff[half][j] = (unsigned)((long)krow * ld_bytes + (long)(rc0 + half * 128 + col) * 2);
      }
    }
}

// Fragment addressing with the lane-dependent part hoisted: a 16x16x32 operand fragment of a half-tile image is
// base[variant] + compile-time immediate.
//  row image: row = r0 + (lane & 15), chunk (4 s + (lane >> 4)) ^ ((row >> 1) & 7): two bases (s = 0, 1)
//  k-major  : two ds_read_b64_tr_b16 at k-rows ka = 32 s + 8 g + q and ka + 4, columns c0 + 4 p (g = lane >> 4,
//             q = (lane & 15) >> 2, p = lane & 3): the granule swizzle depends on (c0 >> 5) & 3 and (c0 >> 4) & 1
template <bool TR, int NV>
struct FragBase {
  int b[NV];
};
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

// ABL (lab only): 1 = no LDS-DMA inside the main loop, 2 = no MFMA, 3 = no fragment reads
template <bool A_TR, bool B_TR, bool STAGGER, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_ph_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware remap + bands of group_m row-tiles (see gemm_dma_kernel)
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = ntn * ntm;
  const int orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
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
  const int nkt = (kend - kbeg) / BK;          // whole K tiles only (the host checks)

  // Operand windows.  The descriptor base advances with the K tile and its size shrinks by the same amount, so the
  // hardware range check stays exact (rows past M / N and k-rows past K read zeros) and the per-lane offsets
  // never change.
  const long lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  const long a_bytes = A_TR ? (long)(p.K - 1) * lda_b + (long)p.M * 2 : (long)(p.M - 1) * lda_b + (long)p.K * 2;
  const long b_bytes = B_TR ? (long)(p.K - 1) * ldb_b + (long)p.N * 2 : (long)(p.N - 1) * ldb_b + (long)p.K * 2;
  const long a_step = A_TR ? (long)BK * lda_b : (long)BK * 2, b_step = B_TR ? (long)BK * ldb_b : (long)BK * 2;
  const long a_first = A_TR ? (long)kbeg * lda_b : (long)kbeg * 2, b_first = B_TR ? (long)kbeg * ldb_b : (long)kbeg * 2;
  unsigned va[2][2], vb[2][2];
  piece_offsets<A_TR>(va, lda_b, m0, wave, lane);
  piece_offsets<B_TR>(vb, ldb_b, n0, wave, lane);

  // image s = 4 t + h of the stream: h = 0 B half 0, 1 A half 0, 2 B half 1, 3 A half 1
  auto request = [&](int t, int h) {
    char* slot = smem + (t & 1) * BUF_BYTES + h * HALF_BYTES + wave * 1024;
    const bool is_a = h & 1;
    const int half = h >> 1;
    const long adv = (is_a ? a_first : b_first) + (long)t * (is_a ? a_step : b_step);
    const long left = (is_a ? a_bytes : b_bytes) - adv;
    const char* base = reinterpret_cast<const char*>(is_a ? p.A : p.B) + adv;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0,
                                                                  (int)max(0L, min(left, 0x7fffffffL)), 0x00020000);
    const unsigned o0 = is_a ? va[half][0] : vb[half][0], o1 = is_a ? va[half][1] : vb[half][1];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, slot), 16, o0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, slot + 8 * 1024), 16, o1, 0, 0, 0);
  };

  BlockReader<A_TR, 4> rda;
  BlockReader<B_TR, 2> rdb;
  rda.init(wr * 64, lane);
  rdb.init(wc * 32, lane);

  f32x4 acc[8][4];                 // [a * 4 + i][b * 2 + j]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];

  // Requests past the last K tile are made all the same: their window is empty, so the range check turns them
  // into zero fills of slots nobody reads any more, and every phase keeps the same counted wait (no tail code).
  // ---- prologue: images 0..5 (K tile 0 and the first two of K tile 1), then images 0 and 1 must have landed
#pragma unroll
  for (int s = 0; s < 6; ++s) request(s >> 2, s & 3);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (STAGGER && wr == 1) {        // second wave group runs half a phase behind the first
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }

  // one phase: reads, request of image g + 6, counted wait, barrier, 16 MFMAs, barrier
  auto phase = [&](auto phc, int t, const char* buf) {
    constexpr int PH = decltype(phc)::value;
    if constexpr (ABL == 3) {
      if (t == 0 && PH == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int s = 0; s < 2; ++s) fb1[j][s] = fb0[j][s] = rdb.get(buf + 0 * HALF_BYTES, j, s);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(buf + 1 * HALF_BYTES, i, s);
      }
    } else if constexpr (PH == 0) {
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
    // image g + 6: K tile t + 1 (h = PH + 2) for PH < 2, K tile t + 2 (h = PH - 2) otherwise
    if constexpr (ABL != 1) {
      request(PH < 2 ? t + 1 : t + 2, PH < 2 ? PH + 2 : PH - 2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    constexpr int A0 = (PH >= 2) ? 4 : 0, B0 = (PH == 1 || PH == 2) ? 2 : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 b = (B0 == 0) ? fb0[j][s] : fb1[j][s];
          if constexpr (ABL == 2) {
            asm volatile("" ::"v"(fa[i][s]), "v"(b));
          } else {
            acc[A0 + i][B0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][s], b, acc[A0 + i][B0 + j], 0, 0, 0);
          }
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < nkt; ++t) {
    const char* buf = smem + (t & 1) * BUF_BYTES;
    phase(std::integral_constant<int, 0>{}, t, buf);
    phase(std::integral_constant<int, 1>{}, t, buf);
    phase(std::integral_constant<int, 2>{}, t, buf);
    phase(std::integral_constant<int, 3>{}, t, buf);
  }
  if (STAGGER && wr == 0) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero fills past the end must not land in the strips

  static_assert(8 * 16 * (4 * 16 + 4) * 4 <= 2 * BUF_BYTES, "epilogue strips must fit in the image slots");
  tile_epilogue<BM, BN, 2, 4, true>(p, acc, smem, m0, n0, wg, nwg);
}

// ---------------------------------------------------------------------------------------------------------
// Ring variant (tile_cfg 11): the same block, images and wave ownership, but TWO phases of 32 MFMAs per K tile
// (four barriers instead of eight: a barrier pair costs ~60 cycles beside a 256-cycle MFMA run) and a 10-slot image
// ring (all 160 KB of LDS) so that the images still travel 2-3 phases ahead:
//   phase A  reads b0, b1, a0 (16 fragments), MFMA a0 x b0, a0 x b1;   phase B  reads a1 (8), MFMA a1 x b1, a1 x b0
//   image s = 4 t + h sits in slot s mod 10; phase g requests images 2 g + 8 and 2 g + 9 at the START of its MFMA
//   segment -- one phase after the last fragment read of the images they replace (2 g - 2, 2 g - 1), which every wave
//   has completed before the barrier in front of that segment -- and each read segment ends with a counted wait
//   that retires what the NEXT phase reads (vmcnt(8) after phase A, vmcnt(6) after phase B).
template <bool A_TR, bool B_TR>
__global__ __launch_bounds__(512) void gemm_ring_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, NSLOT = 10;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = ntn * ntm;
  const int orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
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
  const int nkt = (kend - kbeg) / BK;

  const long lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  const long a_bytes = A_TR ? (long)(p.K - 1) * lda_b + (long)p.M * 2 : (long)(p.M - 1) * lda_b + (long)p.K * 2;
  const long b_bytes = B_TR ? (long)(p.K - 1) * ldb_b + (long)p.N * 2 : (long)(p.N - 1) * ldb_b + (long)p.K * 2;
  const long a_step = A_TR ? (long)BK * lda_b : (long)BK * 2, b_step = B_TR ? (long)BK * ldb_b : (long)BK * 2;
  const long a_first = A_TR ? (long)kbeg * lda_b : (long)kbeg * 2, b_first = B_TR ? (long)kbeg * ldb_b : (long)kbeg * 2;
  unsigned va[2][2], vb[2][2];
  piece_offsets<A_TR>(va, lda_b, m0, wave, lane);
  piece_offsets<B_TR>(vb, ldb_b, n0, wave, lane);

  // image (t, h) into ring slot `slot`: h = 0 B half 0, 1 A half 0, 2 B half 1, 3 A half 1
  auto request = [&](int t, int h, int slot) {
    char* dst = smem + slot * HALF_BYTES + wave * 1024;
    const bool is_a = h & 1;
    const int half = h >> 1;
    const long adv = (is_a ? a_first : b_first) + (long)t * (is_a ? a_step : b_step);
    const long left = (is_a ? a_bytes : b_bytes) - adv;
    const char* base = reinterpret_cast<const char*>(is_a ? p.A : p.B) + adv;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0,
                                                                  (int)max(0L, min(left, 0x7fffffffL)), 0x00020000);
    const unsigned o0 = is_a ? va[half][0] : vb[half][0], o1 = is_a ? va[half][1] : vb[half][1];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst), 16, o0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, dst + 8 * 1024), 16, o1, 0, 0, 0);
  };
  auto wrap = [](int s) { return s >= NSLOT ? s - NSLOT : s; };

  BlockReader<A_TR, 4> rda;
  BlockReader<B_TR, 2> rdb;
  rda.init(wr * 64, lane);
  rdb.init(wc * 32, lane);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];

  // ---- prologue: images 0..7 (K tiles 0 and 1); phase A of tile 0 reads images 0, 1, 2
#pragma unroll
  for (int s = 0; s < 8; ++s) request(s >> 2, s & 3, s);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) {                   // second wave group runs half a phase behind the first
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }

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
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
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
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- phase B (g = 2 t + 1)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[i][s] = rda.get(smem + s3 * HALF_BYTES, i, s);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
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
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    rs0 = wrap(rs0 + 4);
    ws0 = wrap(ws0 + 4);
  }
  if (wr == 0) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero fills past the end must not land in the strips

  tile_epilogue<BM, BN, 2, 4, true>(p, acc, smem, m0, n0, wg, nwg);
}

template <bool A_TR, bool B_TR>
int launch_ring(const GemmParams& p, int splits, hipStream_t stream) {
  constexpr size_t lds = 10 * HALF_BYTES;
  auto k = gemm_ring_kernel<A_TR, B_TR>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
  hipLaunchKernelGGL(k, dim3(ntn * ntm, 1, splits), dim3(512), lds, stream, p);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Complementary variant (tile_cfg 12): ONE barrier per phase.  The two wave groups run the same phase in the same
// barrier interval but in opposite order: group X (wr = 0) issues MFMA_k first and then reads the fragments of
// phase k + 1, group Y (wr = 1) reads the fragments of phase k first and then issues MFMA_k -- on every SIMD the
// matrix pipe goes from the X wave to the Y wave in mid-interval and the fragment reads / LDS-DMA requests of one
// wave always sit beside the MFMAs of the other.  Same images, slots and ownership as gemm_ph_kernel.
//   interval k: both groups request image k + 7 in their read segment and wait (vmcnt(8)) before the closing
//   barrier for what is read in interval k + 1: X's fragments of phase k + 2 and Y's of phase k + 1, i.e. images
//   <= k + 3.  An image is requested >= 1 interval after the last interval that reads its slot (Y's read of phase k
//   is in interval k; the request comes after the next barrier).
// STAMP (lab build only, tile_cfg 13): per-segment s_memtime sums of every wave go to p.split_ws (no other use of it)
template <bool A_TR, bool B_TR, bool STAMP = false>
__global__ __launch_bounds__(512) void gemm_px_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256;
  unsigned long long tsum[4] = {0, 0, 0, 0}, tlast = 0, tmark[4] = {0, 0, 0, 0}, rmark[2] = {0, 0};
  if constexpr (STAMP) { tmark[0] = __builtin_amdgcn_s_memtime(); rmark[0] = __builtin_amdgcn_s_memrealtime(); }
  auto stamp = [&](int seg) {
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      if (seg >= 0) tsum[seg] += now - tlast;
      tlast = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = ntn * ntm;
  const int orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
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
  const int nkt = (kend - kbeg) / BK;

  const long lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  const long a_bytes = A_TR ? (long)(p.K - 1) * lda_b + (long)p.M * 2 : (long)(p.M - 1) * lda_b + (long)p.K * 2;
  const long b_bytes = B_TR ? (long)(p.K - 1) * ldb_b + (long)p.N * 2 : (long)(p.N - 1) * ldb_b + (long)p.K * 2;
  const long a_step = A_TR ? (long)BK * lda_b : (long)BK * 2, b_step = B_TR ? (long)BK * ldb_b : (long)BK * 2;
  const long a_first = A_TR ? (long)kbeg * lda_b : (long)kbeg * 2, b_first = B_TR ? (long)kbeg * ldb_b : (long)kbeg * 2;
  unsigned va[2][2], vb[2][2];
  piece_offsets<A_TR>(va, lda_b, m0, wave, lane);
  piece_offsets<B_TR>(vb, ldb_b, n0, wave, lane);

  auto request = [&](int t, int h) {
    char* slot = smem + (t & 1) * BUF_BYTES + h * HALF_BYTES + wave * 1024;
    const bool is_a = h & 1;
    const int half = h >> 1;
    const long adv = (is_a ? a_first : b_first) + (long)t * (is_a ? a_step : b_step);
    const long left = (is_a ? a_bytes : b_bytes) - adv;
    const char* base = reinterpret_cast<const char*>(is_a ? p.A : p.B) + adv;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0,
                                                                  (int)max(0L, min(left, 0x7fffffffL)), 0x00020000);
    const unsigned o0 = is_a ? va[half][0] : vb[half][0], o1 = is_a ? va[half][1] : vb[half][1];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, slot), 16, o0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(void, slot + 8 * 1024), 16, o1, 0, 0, 0);
  };

  BlockReader<A_TR, 4> rda;
  BlockReader<B_TR, 2> rdb;
  rda.init(wr * 64, lane);
  rdb.init(wc * 32, lane);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];

  // fragments of phase PH of the K tile in `buf`
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
  // request of interval k = 4 t + PH: image k + 7 = (K tile t + 1, h = PH + 3) for PH = 0, (t + 2, h = PH - 1) else
  auto req = [&](auto phc, int t) {
    constexpr int PH = decltype(phc)::value;
    if constexpr (PH == 0) request(t + 1, 3);
    else request(t + 2, PH - 1);
  };
  auto close = [&]() {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    stamp(2);                                   // segment 2: the counted wait
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp(3);                                   // segment 3: the barrier
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;

  // ---- prologue: images 0..6; images 0 and 1 visible at the first barrier, image 2 at the second
#pragma unroll
  for (int s = 0; s < 7; ++s) request(s >> 2, s & 3);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 0) {
    // ------------------------------------------------ group X: MFMA_k, then the fragments of phase k + 1
    reads(P0{}, smem);
    close();
    stamp(-1);
    if constexpr (STAMP) { tsum[2] = 0; tsum[3] = 0; tmark[1] = tlast; }
    for (int t = 0; t < nkt; ++t) {
      const char* buf = smem + (t & 1) * BUF_BYTES;
      const char* nxt = smem + ((t + 1) & 1) * BUF_BYTES;
      mfmas(P0{}); stamp(0); __builtin_amdgcn_sched_barrier(0); reads(P1{}, buf); req(P0{}, t); stamp(1); close();
      mfmas(P1{}); stamp(0); __builtin_amdgcn_sched_barrier(0); reads(P2{}, buf); req(P1{}, t); stamp(1); close();
      mfmas(P2{}); stamp(0); __builtin_amdgcn_sched_barrier(0); req(P2{}, t); stamp(1); close();
      mfmas(P3{}); stamp(0); __builtin_amdgcn_sched_barrier(0); reads(P0{}, nxt); req(P3{}, t); stamp(1); close();
    }
  } else {
    // ------------------------------------------------ group Y: the fragments of phase k, then MFMA_k
    close();
    stamp(-1);
    if constexpr (STAMP) { tsum[2] = 0; tsum[3] = 0; tmark[1] = tlast; }
    for (int t = 0; t < nkt; ++t) {
      const char* buf = smem + (t & 1) * BUF_BYTES;
      reads(P0{}, buf); req(P0{}, t); stamp(1); __builtin_amdgcn_sched_barrier(0); mfmas(P0{}); stamp(0); close();
      reads(P1{}, buf); req(P1{}, t); stamp(1); __builtin_amdgcn_sched_barrier(0); mfmas(P1{}); stamp(0); close();
      reads(P2{}, buf); req(P2{}, t); stamp(1); __builtin_amdgcn_sched_barrier(0); mfmas(P2{}); stamp(0); close();
      req(P3{}, t); stamp(1); __builtin_amdgcn_sched_barrier(0); mfmas(P3{}); stamp(0); close();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero fills past the end must not land in the strips
  if constexpr (STAMP) tmark[2] = __builtin_amdgcn_s_memtime();

  tile_epilogue<BM, BN, 2, 4, true>(p, acc, smem, m0, n0, wg, nwg);
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tmark[3] = __builtin_amdgcn_s_memtime();
    rmark[1] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && p.split_ws != nullptr) {
      float* o = p.split_ws + ((long)blockIdx.x * 8 + wave) * 16;
      for (int i = 0; i < 4; ++i) o[i] = (float)tsum[i];
      o[4] = (float)(tmark[1] - tmark[0]);      // entry -> first interval
      o[5] = (float)(tmark[2] - tmark[1]);      // K loop
      o[6] = (float)(tmark[3] - tmark[2]);      // epilogue incl. store drain
      o[7] = (float)(rmark[1] - rmark[0]);      // whole kernel in 100 MHz ticks
      o[8] = (float)(tmark[3] - tmark[0]);
    }
  }
}

template <bool A_TR, bool B_TR, bool STAMP = false>
int launch_px(const GemmParams& p, int splits, hipStream_t stream) {
  constexpr size_t lds = 2 * BUF_BYTES;
  auto k = gemm_px_kernel<A_TR, B_TR, STAMP>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
  hipLaunchKernelGGL(k, dim3(ntn * ntm, 1, splits), dim3(512), lds, stream, p);
  return 0;
}

template <bool A_TR, bool B_TR, bool STAGGER, int ABL = 0>
int launch_ph(const GemmParams& p, int splits, hipStream_t stream) {
  constexpr size_t lds = 2 * BUF_BYTES;
  auto k = gemm_ph_kernel<A_TR, B_TR, STAGGER, ABL>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
  hipLaunchKernelGGL(k, dim3(ntn * ntm, 1, splits), dim3(512), lds, stream, p);
  return 0;
}

}  // namespace

namespace vg_host {
// phase-pipelined 256x256 tile; returns 0 if launched, -1 if this variant does not apply (caller falls back)
int gemm_ph_launch(const GemmParams& p, int a_tr, int b_tr, int cfg, int splits, hipStream_t stream) {
  if (a_tr && !b_tr) return -1;
  if (p.k_per_split % BK != 0 || p.K % BK != 0) return -1;       // whole K tiles in every split
  if (cfg == 11) {
    if (!a_tr && !b_tr) return launch_ring<false, false>(p, splits, stream);
    if (!a_tr && b_tr) return launch_ring<false, true>(p, splits, stream);
    return launch_ring<true, true>(p, splits, stream);
  }
  if (cfg == 13 && !a_tr && !b_tr) return launch_px<false, false, true>(p, splits, stream);   // lab: stamped build
  if (cfg == 12) {
    if (!a_tr && !b_tr) return launch_px<false, false>(p, splits, stream);
    if (!a_tr && b_tr) return launch_px<false, true>(p, splits, stream);
    return launch_px<true, true>(p, splits, stream);
  }
  if (!a_tr && !b_tr) return launch_ph<false, false, true>(p, splits, stream);
  if (!a_tr && b_tr) return launch_ph<false, true, true>(p, splits, stream);
  return launch_ph<true, true, true>(p, splits, stream);
}
}  // namespace vg_host
