// Causal multi-head self-attention with in-kernel ALiBi (SURVEY.md K1 + K5),
// head_dim = 64, for gfx950.  Replaces the (B,H,T,T) mask/bias materialisation
// and F.scaled_dot_product_attention of modules/attention/attention.py:60-77.
//
// Orientation (all three kernels): the time index that softmax statistics
// belong to sits on the MFMA *lane*, so row max / row sum / rescale are
// lane-local (one cross-half shuffle) and the probability tile in the
// accumulator registers is directly the B operand of the next product
// (no LDS round trip for P):
//   forward / dQ : S^T[key][query] = K Q^T ; O^T[d][query] = V^T P^T ; dQ^T = K^T dS^T
//   dK/dV        : S[query][key]   = Q K^T ; dV^T[d][key] = dO^T P   ; dK^T = Q^T dS
// K/V (or Q/dO) tiles are staged once per block through LDS in two images:
// a K-contiguous RowTile (operand of the score products) and a TrTile read
// with ds_read_b64_tr_b16 (operand of the products that sum over time).
// Scores live in the log2 domain: s2 = (q.k / 8 - slope (i - j)) * log2(e).
#include <type_traits>
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

using namespace vg;

namespace {

#ifndef VG_LAB_ATTN
#define VG_LAB_ATTN 0      // lab builds (tools/lab/variant.sh ... -DVG_LAB_ATTN=bits), results wrong, timing only.  attn2_fwd:
#endif                     // 1 no exp2, 2 no PV products, 4 no barrier, 8 no DMA requests, 16 no S products, 32 no maxima,
                           // 64 empty tile body; backward kernels: 256 empty tile body, 512 no DMA requests
constexpr int DH = 64;
constexpr int QB = 128;      // time rows owned by a block (4 waves x 32)
constexpr int TB = 64;       // time rows staged per LDS tile
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float LN2 = 0.69314718055994530942f;
constexpr float SCALE = 0.125f;   // 1 / sqrt(64)
// resident blocks per CU the bf16 kernels are register-budgeted for (measured: 128-query forward 3; backward 2 until
// round 5, 3 since: dQ 75.9 -> 71.7 us, dK/dV 93.9 -> 86.6 us at B = 16, T = 1000; 4 spills)
#ifndef VG_ATTN_PAIR
#define VG_ATTN_PAIR 1     // forward: a wave owns the 32-query groups w and 7 - w of its block (0: 2w and 2w + 1, lab only)
#endif
#ifndef VG_ATTN_TILESKIP
#define VG_ATTN_TILESKIP 1  // backward: per-32x32-block test that drops negligible blocks inside the window (0: lab, branch-free tile body)
#endif
#ifndef VG_ATTN_HEADMIX
#define VG_ATTN_HEADMIX 1   // heads x and 15 - x share an XCD (0: x and x + 8, lab)
#endif
#ifndef VG_ATTN_PRIO
#define VG_ATTN_PRIO 0      // lab: static wave priority by block parity (bit 0: forward, blockIdx & 1; bit 1: backward, blockIdx % 3)
#endif
#ifndef VG_ATTN_BSW
#define VG_ATTN_BSW 1       // swizzle of the bf16 backward kernels' row images (see row_swz): 1 = the dual-use one (round 6), 0 = RowTile's (lab)
#endif
constexpr int BSW = VG_ATTN_BSW;
#ifndef VG_ATTN_OCC_FWD
#define VG_ATTN_OCC_FWD 3
#endif
#ifndef VG_ATTN_OCC
#define VG_ATTN_OCC 3     // round 5: with the transposed images gone from the bf16 stages three blocks fit a CU (168 VGPRs, no spill)
#endif

// Block index -> ((batch, head) pair, rank of the tile inside the pair: 0 = longest sweep).
// sched 0 (default): launch-wide longest-first order (LPT): rank-major, all pairs' longest tiles first.
// sched 1 (VG_ATTN_SCHED=1, measured and NOT kept): all tiles of a pair dispatched back to back onto one XCD (blocks i
// and i + 8 share an XCD under the round-robin placement) and swept from the same end, so that co-resident blocks of a
// pair request the same K/V (Q/dO) tile at about the same time and all but one hit the XCD's L2.  It was built because
// with LPT the blocks resident together belong to different pairs (L2 hit rate of the forward: 25 %), but the stream
// is not what bounds these kernels (every request pointed at one L2-resident tile: 58.1 vs 59.6 us) and the lost
// balance costs more than the hits return: forward 88 vs 65 us, backward 226 vs 210 us at B = 16, T = 1000.
VG_DEVICE void pair_and_rank(int bid, int ntiles, int npairs, int sched, int& hb, int& rank) {
  // sched & 4 (lab, round 6): the launch order scattered by a multiplicative permutation that keeps bid mod 8 (the XCD
  // class of a pair) -- blocks of every sweep length start together instead of rank by rank.  Built to test whether the
  // memory-bound prologue / epilogue phases and the compute-bound sweeps of equal-rank blocks run in step chip-wide
  if ((sched & 4) && ((ntiles * npairs) & 7) == 0) {
    const unsigned total = (unsigned)(ntiles * npairs);
    bid = (int)(((unsigned long long)(unsigned)bid * 1001ull) % total);     // 1001 = 8 * 125 + 1; odd: a permutation when total is a power of two times ...
  }
  // sched & 8 (lab, round 6): rank-major like the default, but the ranks leave in the order packed into bits 8.. (three
  // bits per position, up to 8 tiles per pair): e.g. 0, 7, 1, 6 ... mixes long and short sweeps in the first round
  if ((sched & 8) && ntiles <= 8) {
    hb = bid % npairs;
    rank = (sched >> (8 + 3 * (bid / npairs))) & 7;
    if (rank >= ntiles) rank = ntiles - 1;
    return;
  }
  sched &= 1;
  if (sched == 1 && (npairs & 7) == 0) {
    const int j = bid >> 3;
    hb = (j / ntiles) * 8 + (bid & 7);
    rank = j % ntiles;
  } else if (sched == 1) {
    hb = bid / ntiles;
    rank = bid % ntiles;
  } else {
    hb = bid % npairs;
    rank = bid / npairs;
  }
}

// Which head a block of pair slot `hb` works on.  Blocks are dealt round-robin over the 8 XCDs (blocks i and i + 8 share
// one) and H * B is a multiple of 8, so with h = hb % H an XCD would work on heads x and x + 8 of every 16 only -- and
// under ALiBi the cost of a head grows with its index (the steep heads' far tiles are dropped / outside the window):
// XCD 7 (heads 7, 15) carried half as much again as XCD 0 (heads 0, 8) and the launch ended on it.  Slots 8..15 of
// every 16 take the heads in descending order, so that an XCD gets heads x and 15 - x (round 5).
VG_DEVICE int head_of_slot(int hb, int H) {
  const int j = hb % H;
  if ((H & 15) != 0 || !VG_ATTN_HEADMIX) return j;
  const int j16 = j & 15;
  return (j & ~15) | (j16 < 8 ? j16 : 23 - j16);
}

template <typename T> struct NVec { static constexpr int v = TB * DH / Traits<T>::VEC / 256; };  // 2 bf16 / 4 f32

// ---- stage a [64 time][64 d] slab (one head) global -> registers -> LDS images
template <typename T>
VG_DEVICE void slab_load(uint4 (&r)[NVec<T>::v], const T* __restrict__ base, long row_stride, int t0, int t_lim,
                         int tid) {
  constexpr int VEC = Traits<T>::VEC, CPR = DH / VEC;
#pragma unroll
  for (int it = 0; it < NVec<T>::v; ++it) {
    const int v = tid + 256 * it;
    const int row = v / CPR, c16 = v % CPR;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (t0 + row < t_lim) val = *reinterpret_cast<const uint4*>(base + (long)(t0 + row) * row_stride + c16 * VEC);
    r[it] = val;
  }
}
template <typename T, bool ROW, bool TR>
VG_DEVICE void slab_store(const uint4 (&r)[NVec<T>::v], char* row_img, char* tr_img, int tid) {
  constexpr int VEC = Traits<T>::VEC, CPR = DH / VEC;
#pragma unroll
  for (int it = 0; it < NVec<T>::v; ++it) {
    const int v = tid + 256 * it;
    const int row = v / CPR, c16 = v % CPR;
    if constexpr (ROW) RowTile<T, DH>::store_vec(row_img, row, c16, r[it]);
    if constexpr (TR) TrTile<T, DH>::store_vec(tr_img, row, c16, r[it]);
  }
}

// ---- bf16 fast path: the same two LDS images written by LDS-DMA (`buffer_load ... lds`, 1 KiB per
// wave-instruction, no VGPR hop).  The DMA writes lane-linear, so each lane's SOURCE address carries the
// image's swizzle; rows past the end of the sequence are zero-filled by the buffer range check.
// 4 waves: each issues pieces wave, wave + 4 of the 8 pieces (8 rows x 128 B) of an image.
VG_DEVICE __amdgpu_buffer_rsrc_t slab_rsrc(const bf16_t* base, long row_stride, int Tn) {
  const long bytes = (long)(Tn - 1) * row_stride * 2 + DH * 2;
  // provably wave-uniform inputs: the descriptor must live in SGPRs (it is an "s" operand of the asm DMA)
  const uintptr_t a = (uintptr_t)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int nb = __builtin_amdgcn_readfirstlane((int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16_t*>(((uintptr_t)hi << 32) | lo), 0, nb, 0x00020000);
}
// One LDS-DMA piece (1 KiB per wave-instruction) issued from inline asm, so that it is NOT part of hipcc's s_waitcnt
// bookkeeping: with the builtin form the compiler treats every later LDS read that may alias the destination as
// dependent on the DMA and drains vmcnt in the middle of the tile (s_waitcnt vmcnt(0) in front of the transposed
// reads of the PV / dV / dK products: the NEXT tile's transfer, requested one barrier earlier, was waited for before
// the current tile's second half could start -- its latency was exposed once per tile).  The kernels order DMA
// against the reads themselves: a counted s_waitcnt vmcnt + s_barrier before the first read of a stage.
// `lds_addr` must be wave-uniform; M0 is saved and restored inside the statement (cdna_hip_programming.md 5.7).
VG_DEVICE void dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned voff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}
VG_DEVICE unsigned lds_addr_of(const char* p) { return (unsigned)(uintptr_t)LDS_PTR(const char, p); }

// SW (round 6): which 16-byte-chunk swizzle a ROW image carries.  0: chunk ^ ((row >> 1) & 7) (RowTile: conflict-free
// ds_read_b128 row reads; the transposed reads of the backward kernels -- four rows x 64 bytes per half-wave -- put rows q
// and q + 2 on the same banks: 2-way, 25 % of the LDS-active cycles by the counters, rounds 3 - 5).  1: the same XOR with
// bit 2 flipped for odd row pairs, x(row) = ((row >> 1) & 7) ^ (((row >> 1) & 1) << 2): still a bijection of (row >> 1) & 7,
// so the row reads stay conflict-free, and rows q + 2, q + 3 of a transposed read move to the other 64-byte half of
// their 128-byte row -- the half-wave's four rows x 64 bytes tile one 256-byte bank row exactly.  Used by the images of
// the bf16 backward kernels (both kinds of read on one image).
VG_DEVICE int row_swz(int row, int sw) {
  const int v = (row >> 1) & 7;
  return sw ? (v ^ ((v & 1) << 2)) : v;
}
template <int SW>
VG_DEVICE bf16x8 row_frag_sw(const char* base, int row, int s, int lane) {       // RowTile<bf16, 64>::frag with swizzle SW
  return *reinterpret_cast<const bf16x8*>(base + row * 128 + (((2 * s + (lane >> 5)) ^ row_swz(row, SW)) << 4));
}
template <int SW>
VG_DEVICE bf16x8 tr_frag_sw(const char* base, int k0, int col0, int s, int lane) {   // RowTile<bf16, 64>::tr_frag<true>
  // k0 % 16 == 0 and col0 % 32 == 0 (both are multiples of 32 at every call site): the swizzle term of row k0 + 16 s + r
  // is that of r, the second row (+8) has bit 2 of it flipped, and col0 only flips bit 2 of the chunk -- one lane-constant
  // offset, XORs and adds of immediates per read.
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = g >> 1;
  const int r = 4 * h + q;
  const int lane_off = r * 128 + (((2 * (g & 1) + (p >> 1)) ^ row_swz(r, SW)) << 4) + ((p & 1) << 3);
  const int sel = (col0 >> 5) & 1;
  const int oa = (sel ? (lane_off ^ 64) : lane_off) + (k0 + 16 * s) * 128;
  const int ob = (sel ? lane_off : (lane_off ^ 64)) + (k0 + 16 * s) * 128 + 1024;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + oa));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + ob));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <bool TR, int SW = 0>
VG_DEVICE void slab_dma(__amdgpu_buffer_rsrc_t rsrc, char* img, long row_stride, int t0, int wave, int lane) {
  const unsigned img_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(img));
  const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int piece = wv + 4 * j;
    const int row = piece * 8 + (lane >> 3);
    const int pos = lane & 7;               // 16-byte position inside the 128-byte LDS row
    int col;
    if constexpr (!TR) col = (pos ^ row_swz(row, SW)) * 8;
    else col = (((pos >> 2) ^ ((row >> 1) & 1)) * 32) + (pos & 3) * 8;
    const unsigned voff = (unsigned)(((long)(t0 + row) * row_stride + col) * 2);
    dma16(rsrc, img_lds + piece * 1024, voff);
  }
}
VG_DEVICE void dma_wait_and_publish() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces have landed ...
  __builtin_amdgcn_s_barrier();                      // ... and so have everyone else's; the other stage is free
}

// ---- per-lane operand fragments of one time row held in registers (B operand, k = d)
template <typename T> struct RowRegs;
template <> struct RowRegs<bf16_t> {
  bf16x8 f[4];
  VG_DEVICE void load(const bf16_t* __restrict__ row, int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const bf16x8*>(row + 16 * s + 8 * (lane >> 5));
  }
};
template <> struct RowRegs<float> {
  float f[32];
  VG_DEVICE void load(const float* __restrict__ row, int lane) {
#pragma unroll
    for (int s = 0; s < 32; ++s) f[s] = row[2 * s + (lane >> 5)];
  }
};

template <typename T, int SW = 0>
VG_DEVICE f32x16 mma_row_regs(const char* row_img, int row, const RowRegs<T>& b, int lane, f32x16 acc) {
  constexpr int STEPS = DH / Traits<T>::KSTEP;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    if constexpr (sizeof(T) == 2 && SW != 0) acc = Traits<T>::mfma(row_frag_sw<SW>(row_img, row, s, lane), b.f[s], acc);
    else acc = Traits<T>::mfma(RowTile<T, DH>::frag(row_img, row, s, lane), b.f[s], acc);
  }
  return acc;
}

// the same product with the transposed fragments read out of a ROW image (bf16 backward kernels: K, Q and dO are
// staged once instead of twice -- a third less LDS-DMA traffic for dQ, half for dK / dV)
template <int SW = 0>
VG_DEVICE f32x16 mma_tr_acc_rowimg(const char* row_img, int k0, int db, const f32x16& x, int lane, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < AccOperand<bf16_t>::STEPS; ++s)
    acc = Traits<bf16_t>::mfma(tr_frag_sw<SW>(row_img, k0, db * 32, s, lane), AccOperand<bf16_t>::get(x, s), acc);
  return acc;
}

// acc[d-block db] += (tr image)^T[d][time k0..k0+31] . X[time][lane]
template <typename T>
VG_DEVICE f32x16 mma_tr_acc(const char* tr_img, int k0, int db, const f32x16& x, int lane, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < AccOperand<T>::STEPS; ++s)
    acc = Traits<T>::mfma(TrTile<T, DH>::template frag<true>(tr_img, k0, db * 32, s, lane), AccOperand<T>::get(x, s),
                          acc);
  return acc;
}

// store a transposed accumulator pair O^T[2][d][time = lane] as rows [time][64 d] scaled by `mul`
template <typename T>
VG_DEVICE void store_rows_T(T* __restrict__ dst_row, const f32x16 (&o)[2], float mul, int lane) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = db * 32 + 8 * g + 4 * (lane >> 5);
      if constexpr (sizeof(T) == 2) {
        bf16x4 v = {(bf16_t)(o[db][4 * g] * mul), (bf16_t)(o[db][4 * g + 1] * mul), (bf16_t)(o[db][4 * g + 2] * mul),
                    (bf16_t)(o[db][4 * g + 3] * mul)};
        *reinterpret_cast<bf16x4*>(dst_row + d0) = v;
      } else {
        f32x4 v = {o[db][4 * g] * mul, o[db][4 * g + 1] * mul, o[db][4 * g + 2] * mul, o[db][4 * g + 3] * mul};
        *reinterpret_cast<f32x4*>(dst_row + d0) = v;
      }
    }
}

// bf16: the same rows through a wave-private 4 KB LDS area, leaving as whole 128-byte rows (16-byte stores, 8 rows per
// wave-instruction) instead of 8-byte pieces at a row stride (32 rows x 16 bytes per wave-instruction: the backward
// kernels' epilogues were bound by store issue, the forward's O has left this way since round 3).  `area`: 32 rows x
// 128 bytes this wave owns; the caller has made sure no other wave still reads it.  Rows >= nrows are not stored.
VG_DEVICE void store_rows_T_lds(bf16_t* __restrict__ dst, long row_stride, int nrows, const f32x16 (&o)[2], float mul,
                                char* area, int lane) {
  const int row = lane & 31;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = db * 32 + 8 * g + 4 * (lane >> 5);
      const bf16x4 v = {(bf16_t)(o[db][4 * g] * mul), (bf16_t)(o[db][4 * g + 1] * mul), (bf16_t)(o[db][4 * g + 2] * mul),
                        (bf16_t)(o[db][4 * g + 3] * mul)};
      *reinterpret_cast<bf16x4*>(area + row * 128 + ((((d0 >> 3) ^ (row & 7))) << 4) + (d0 & 7) * 2) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave's LDS accesses stay in order; the data must have landed)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = it * 8 + (lane >> 3), c16 = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(area + r * 128 + ((c16 ^ (r & 7)) << 4));
    if (r < nrows) *reinterpret_cast<uint4*>(dst + (long)r * row_stride + c16 * 8) = v;
  }
}

// lanes l and l + 32 hold the same accumulator column: combine the two halves with one v_permlane32_swap (VALU)
// instead of a ds_bpermute round trip through the LDS crossbar.  With both operands = v the instruction returns
// {v[l & 31], v[(l & 31) + 32]} in every lane.
VG_DEVICE float xhalf_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
VG_DEVICE float xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v_max3_f32 without the canonicalising v_max the compiler puts in front of fmaxf on MFMA results
VG_DEVICE float max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <typename T> struct LdsPlan {
  static constexpr int ROW_BYTES = sizeof(T) == 2 ? TB * 128 : TB * (DH + 1) * 4;
  static constexpr int TR_BYTES = sizeof(T) == 2 ? TB * 128 : TB * DH * 4;
};

// exp2 of a non-positive argument: raw v_exp_f32 for the bf16 path, accurate exp2f for fp32 parity
template <typename T> VG_DEVICE float fexp2(float x) {
  if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f(x);
  else return exp2f(x);
}

VG_DEVICE float max16(const f32x16& s) {
  const float a = max3(s[0], s[1], s[2]), b = max3(s[3], s[4], s[5]), c = max3(s[6], s[7], s[8]);
  const float d = max3(s[9], s[10], s[11]), e = max3(s[12], s[13], s[14]);
  return max3(max3(a, b, c), max3(d, e, s[15]), s[15]);
}

// 16 per-row constants (rows of the accumulator map) from an LDS float array
VG_DEVICE f32x16 rows16(const float* arr, int row0, int lane) {
  f32x16 r;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(arr + row0 + 8 * g + 4 * (lane >> 5));
    r[4 * g] = v[0]; r[4 * g + 1] = v[1]; r[4 * g + 2] = v[2]; r[4 * g + 3] = v[3];
  }
  return r;
}

// ---- per-(batch, head) statistics the bf16 forward leaves for the backward's ALiBi window (round 5).
// stats: fp32 [B*H][attn_stats_floats(Tn)]: [0] = max |k_j|^2 over the sequence's keys, written by the one wave that sees
// every K tile of the pair ([1..3] pad); [4 + g] = max |q_i|^2 and [4 + 4 n128 + g] = max -L_i (L_i: the row's
// log-sum-exp in the log2 domain) over the valid queries of 32-query group g of the sequence (n128 = 128-query blocks,
// four groups each; written by both bf16 forward kernels).  Plain stores, one writer per entry: no initialisation, no
// atomics, bitwise repeatable.  With p_ij = 2^(s_ij - L_i) and
// s_ij <= c2 |q_i| |k_j| - slope2 (i - j), every probability further than
//     W = (c2 max|q| max|k| + max(-L) + thr) / slope2
// from the diagonal is below 2^-thr: the backward kernels do not even STREAM those tiles (the per-tile test they had
// dropped the products but still paid DMA + barrier + the S product of every tile; at T = 1000 the steep half of the
// 16 ALiBi heads needs 1-4 of its up to 16 tiles).
VG_DEVICE int attn_stats_floats(int Tn) { return 4 + 8 * ((Tn + QB - 1) / QB); }        // [0] max |k|^2, [1..3] pad (the arrays stay 16-byte aligned)
// The window in frames from the statistics of a pair.  Every address is wave-uniform and the buffer is read-only in the
// backward kernels, so these are SCALAR loads (s_load_dwordx4) and a few v_max on uniform values: the first version --
// one entry per lane, two DPP wave maxima -- cost 1.4 us per block, which at four rounds of blocks per launch ate what the
// window saved on the steep heads.  q-side entries [e0, e0 + n) of both arrays, n a multiple of 4.
VG_DEVICE float attn_window(const float* __restrict__ st, int n128, int e0, int n, float slope2, float c2, float thr) {
  float q2 = 0.f, nl = -INFINITY;
  const f32x4* __restrict__ q4 = reinterpret_cast<const f32x4*>(st + 4 + e0);
  const f32x4* __restrict__ n4 = reinterpret_cast<const f32x4*>(st + 4 + 4 * n128 + e0);
  for (int i = 0; i < n / 4; ++i) {
    const f32x4 a = q4[i], c = n4[i];
    q2 = fmaxf(fmaxf(q2, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
    nl = fmaxf(fmaxf(nl, fmaxf(c[0], c[1])), fmaxf(c[2], c[3]));
  }
  const float bound = c2 * sqrtf(q2 * st[0]) * 1.01f + nl + thr;          // log2 units; 1 % for the rounding of the norms
  const float w = bound * __builtin_amdgcn_rcpf(slope2) * 1.0001f;        // (approximate reciprocal: rounded up)
  return w >= 0.f ? w : (w < 0.f ? 0.f : INFINITY);                       // NaN (inf - inf, 0 * inf) -> no window
}


// =====================================================================================
// forward
//
// Scores are kept in raw q.k units inside the accumulators; the ALiBi term is
// split into (a) a loop-invariant per-register constant 8*slope*row (the MFMA's
// initial accumulator -> free), (b) a per-tile scalar slope2*(tile start - wave's
// first query) added inside the exp2 argument, (c) a per-lane constant that
// softmax is invariant to (restored in the stored LSE).  Positions are taken
// relative to the wave's first query so the large distances only appear where
// the probability underflows anyway.  Causal compares run on diagonal tiles only.
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? VG_ATTN_OCC_FWD : 1) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                       float* __restrict__ lse, const float* __restrict__ slopes,
                                                       int Tn, int H, const int* __restrict__ lengths, int sched,
                                                       const int* __restrict__ cu, int Mtot, float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_row = smem;
  char* v_tr = smem + LdsPlan<T>::ROW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, longest sweeps first over the whole launch (LPT order); the tiles of one (b, h)
  // are H*B apart, i.e. on the same XCD whenever H*B is a multiple of 8, and share K/V in its L2
  const int nqt = (Tn + QB - 1) / QB, HB = gridDim.x / nqt;
  int hb, rank;
  pair_and_rank(blockIdx.x, nqt, HB, sched, hb, rank);
  const int qt = nqt - 1 - rank, h = head_of_slot(hb, H), b = hb / H;
  const int D = H * DH;
  const long rs = 3L * D;
  const int soff = cu ? cu[b] : b * Tn;                 // first row of the sequence in the [rows][...] tensors
  const int Tr = cu ? cu[b + 1] - soff : Tn;            // rows the sequence owns (packed layout: its own length)
  const int len = lengths ? min(lengths[b], Tr) : Tr;
  const int q0 = qt * QB;
  const int qw0 = q0 + wave * 32;
  const int query = qw0 + (lane & 31);
  const T* __restrict__ base = qkv + (long)soff * rs + h * DH;
  T* __restrict__ obase = out + (long)soff * D + h * DH;

  // statistics of the backward's ALiBi window (bf16 launches with a workspace; layout: attn_stats_floats)
  const int n128 = (Tn + QB - 1) / QB;
  float* __restrict__ st_pair = (sizeof(T) == 2 && stats) ? stats + (long)(b * H + h) * attn_stats_floats(Tn) : nullptr;
  if (q0 >= len) {   // fully padded tile: zero rows (attention.py:80 re-mask)
    f32x16 z[2] = {zero16(), zero16()};
    if (query < Tr) store_rows_T<T>(obase + (long)query * D, z, 0.f, lane);
    if (st_pair && lane == 0) {
      st_pair[4 + 4 * qt + wave] = 0.f;
      st_pair[4 + 4 * n128 + 4 * qt + wave] = -INFINITY;
    }
    return;
  }
  const int qend = min(q0 + QB, len);
  const int nkt = (qend + TB - 1) / TB;

  RowRegs<T> qf;
  qf.load(base + (long)min(query, Tr - 1) * rs, lane);
  float qmax2 = 0.f, kmax2 = 0.f;
  // the wave that holds the sequence's last valid query computes every K tile of the pair: it takes max |k|^2 along
  const bool kstat = st_pair != nullptr && len - 1 < q0 + QB && wave == ((len - 1 - q0) >> 5);
  if constexpr (sizeof(T) == 2) {
    if (st_pair) {
      float part = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const bf16x2 v = {qf.f[st][j], qf.f[st][j + 1]};
          part = __builtin_amdgcn_fdot2_f32_bf16(v, v, part, false);
        }
      qmax2 = wave_max(query < len ? xhalf_sum(part) : 0.f);
    }
  }
  const float slope = slopes[h];
  const float slope2 = slope * LOG2E, c2 = SCALE * LOG2E;
  f32x16 kinit;
#pragma unroll
  for (int i = 0; i < 16; ++i) kinit[i] = slope * 8.0f * (float)acc_row(i, lane);

  f32x16 o[2] = {zero16(), zero16()};
  float m = -INFINITY, l = 0.f;

  // bf16: two LDS stages filled by LDS-DMA, one barrier per tile (tile kt+1 lands while tile kt is
  // consumed); fp32 parity path: register-staged single stage
  constexpr bool DMA = sizeof(T) == 2;
  constexpr int STAGE = LdsPlan<T>::ROW_BYTES + LdsPlan<T>::TR_BYTES;
  uint4 rk[DMA ? 1 : NVec<T>::v], rv[DMA ? 1 : NVec<T>::v];
  __amdgpu_buffer_rsrc_t rsk, rsv;
  if constexpr (DMA) {
    rsk = slab_rsrc(reinterpret_cast<const bf16_t*>(base + D), rs, Tr);
    rsv = slab_rsrc(reinterpret_cast<const bf16_t*>(base + 2 * D), rs, Tr);
    slab_dma<false>(rsk, smem, rs, 0, wave, lane);
    slab_dma<true>(rsv, smem + LdsPlan<T>::ROW_BYTES, rs, 0, wave, lane);
  } else {
    slab_load<T>(rk, base + D, rs, 0, Tr, tid);
    slab_load<T>(rv, base + 2 * D, rs, 0, Tr, tid);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const int kv0 = kt * TB;
    if constexpr (DMA) {
      dma_wait_and_publish();
      if (kt + 1 < nkt) {
        char* nx = smem + ((kt + 1) & 1) * STAGE;
        slab_dma<false>(rsk, nx, rs, kv0 + TB, wave, lane);
        slab_dma<true>(rsv, nx + LdsPlan<T>::ROW_BYTES, rs, kv0 + TB, wave, lane);
      }
      k_row = smem + (kt & 1) * STAGE;
      v_tr = k_row + LdsPlan<T>::ROW_BYTES;
    } else {
      __syncthreads();
      slab_store<T, true, false>(rk, k_row, nullptr, tid);
      slab_store<T, false, true>(rv, nullptr, v_tr, tid);
      __syncthreads();
      if (kt + 1 < nkt) {
        slab_load<T>(rk, base + D, rs, kv0 + TB, Tr, tid);
        slab_load<T>(rv, base + 2 * D, rs, kv0 + TB, Tr, tid);
      }
    }
    if (qw0 + 31 < kv0) continue;   // this wave's queries all precede the tile (causal)
    if constexpr (sizeof(T) == 2) {
      if (kstat) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          float part = 0.f;
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            const bf16x8 kf = RowTile<T, DH>::frag(k_row, kb * 32 + (lane & 31), st, lane);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              const bf16x2 v = {kf[j], kf[j + 1]};
              part = __builtin_amdgcn_fdot2_f32_bf16(v, v, part, false);
            }
          }
          kmax2 = fmaxf(kmax2, xhalf_sum(part));
        }
      }
    }
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) s[kb] = mma_row_regs<T>(k_row, kb * 32 + (lane & 31), qf, lane, kinit);
    if (kv0 + TB - 1 > qw0) {       // diagonal tile: mask key > query
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int lim = query - kv0 - kb * 32;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kb][i] = acc_row(i, lane) <= lim ? s[kb][i] : -INFINITY;
      }
    }
    const float b0 = slope2 * (float)(kv0 - qw0), b1 = slope2 * (float)(kv0 + 32 - qw0);
    const float mx = xhalf_max(fmaxf(fmaf(max16(s[0]), c2, b0), fmaf(max16(s[1]), c2, b1)));
    const float m_new = fmaxf(m, mx);
    const float alpha = fexp2<T>(m - m_new);
    const float off0 = b0 - m_new, off1 = b1 - m_new;
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p0 = fexp2<T>(fmaf(s[0][i], c2, off0));
      const float p1 = fexp2<T>(fmaf(s[1][i], c2, off1));
      s[0][i] = p0;
      s[1][i] = p1;
      ps += p0 + p1;
    }
    l = l * alpha + xhalf_sum(ps);
    m = m_new;
    if (__any(alpha != 1.f)) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) o[db] = mma_tr_acc<T>(v_tr, kb * 32, db, s[kb], lane, o[db]);
  }
  const float L2 = m + log2f(l) - slope2 * (float)(query - qw0);            // the row's log-sum-exp, log2 domain
  if constexpr (sizeof(T) == 2) {
    // bf16: O leaves as whole 128-byte rows through this wave's 4 KB of the freed stages (round 5, as in the backward)
    __syncthreads();
    const bool valid = query < len;
    store_rows_T_lds(obase + (long)qw0 * D, D, min(32, Tr - qw0), o, valid ? 1.f / l : 0.f, smem + wave * 4096, lane);
    if (valid && lane < 32) lse[(long)h * Mtot + soff + query] = L2 * LN2;
  } else if (query < Tr) {
    const bool valid = query < len;
    store_rows_T<T>(obase + (long)query * D, o, valid ? 1.f / l : 0.f, lane);
    if (valid && lane < 32) lse[(long)h * Mtot + soff + query] = L2 * LN2;
  }
  if (st_pair) {
    const float nl = wave_max(query < len ? -L2 : -INFINITY), km = wave_max(kmax2);
    if (lane == 0) {
      st_pair[4 + 4 * qt + wave] = qmax2;
      st_pair[4 + 4 * n128 + 4 * qt + wave] = nl;
      if (kstat) st_pair[0] = km;
    }
  }
}

// =====================================================================================
// forward, bf16, round 3 (`attn2_fwd_kernel`): 64 queries per wave.
//
// Same orientation and arithmetic as attn_fwd_kernel (key on the accumulator rows, query on the lane, ALiBi row term
// as the initial accumulator, log2-domain online softmax), re-shaped around what bounded that kernel on MI355X:
//   * a wave owns TWO 32-query column sets, so every K fragment (ds_read_b128) and every V^T fragment (two
//     ds_read_b64_tr_b16) feeds two MFMAs, a block of 4 waves owns 256 queries and one K/V tile (16 KB) is staged
//     once per 256 queries: half the LDS reads, LDS-DMA requests and barriers per MFMA;
//   * the two query sets give the scheduler two independent chains per tile: the softmax VALU work of set 0 sits
//     beside the S MFMAs of set 1, that of set 1 beside the PV MFMAs of set 0;
//   * K/V tiles stream through a THREE-stage ring, requested two tiles ahead by inline-asm LDS-DMA and retired by
//     a counted `s_waitcnt vmcnt(4)` + one barrier per tile (one tile always stays in flight across the barrier);
//   * the row sum stays per lane half until the end (no cross-half exchange per tile), the cross-half maximum is one
//     v_permlane32_swap, the 16-register maxima are v_max3 chains;
//   * O leaves through LDS as whole 128-byte rows (16-byte stores) instead of 8-byte pieces at a row stride.
// Two blocks (8 waves) per CU: <= 256 VGPRs.
// =====================================================================================
constexpr int QW2 = 64;             // queries per wave (two 32-query groups)
constexpr int QB2 = 4 * QW2;        // queries per block
constexpr int STAGE2 = 2 * TB * 128;   // K row image + V transposed-read image of one 64-key tile
constexpr int NSTAGE2 = 3;

__global__ __launch_bounds__(256, 2) void attn2_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           float* __restrict__ lse, const float* __restrict__ slopes,
                                                           int Tn, int H, const int* __restrict__ lengths, float skip_thr,
                                                           int sched, const int* __restrict__ cu, int Mtot,
                                                           float* __restrict__ stats) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((VG_ATTN_PRIO & 1) && (blockIdx.x & 1)) __builtin_amdgcn_s_setprio(1);
  // 1-D grid, longest sweeps first over the whole launch (LPT order); the tiles of one (b, h) are H*B apart
  const int nqt = (Tn + QB2 - 1) / QB2, HB = gridDim.x / nqt;
  int hb, rank;
  pair_and_rank(blockIdx.x, nqt, HB, sched, hb, rank);
  const int qt = nqt - 1 - rank, h = head_of_slot(hb, H), b = hb / H;
  const int D = H * DH;
  const long rs = 3L * D;
  const int soff = cu ? cu[b] : b * Tn;                 // first row of the sequence in the [rows][...] tensors
  const int Tr = cu ? cu[b + 1] - soff : Tn;            // rows the sequence owns (packed layout: its own length)
  const int len = lengths ? min(lengths[b], Tr) : Tr;
  const int q0 = qt * QB2;
  // Causal balance (round 5): a wave owns the 32-query groups w and 7 - w of the block's eight, not two neighbours.  Of
  // the block's four diagonal tiles group g needs g / 2 + 1, so every wave computes 5 of the 8 (group, tile) pairs there
  // (neighbouring groups: 2, 4, 6 and 8 of them -- the block ran at the pace of its last wave: 15 % of the wave-time).
#if VG_ATTN_PAIR
  const int qg0[2] = {q0 + 32 * wave, q0 + 32 * (7 - wave)};          // first query of the wave's two groups
#else
  const int qg0[2] = {q0 + 64 * wave, q0 + 64 * wave + 32};           // (lab build: neighbouring groups, the round-4 map)
#endif
  const T* __restrict__ base = qkv + (long)soff * rs + h * DH;
  T* __restrict__ obase = out + (long)soff * D + h * DH;
  float* __restrict__ st_pair = stats ? stats + (long)(b * H + h) * attn_stats_floats(Tn) : nullptr;
  const int n128 = (Tn + QB - 1) / QB;

  if (q0 >= len) {   // fully padded tile: zero rows (attention.py:80 re-mask)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r64 = it * 8 + (lane >> 3);
      const int row = qg0[r64 >> 5] + (r64 & 31);
      if (row < Tr) *reinterpret_cast<uint4*>(obase + (long)row * D + (lane & 7) * 8) = make_uint4(0, 0, 0, 0);
    }
    if (st_pair && lane < 2) {                       // this wave's two groups: no valid query
      const int g = lane == 0 ? wave : 7 - wave;
      if (8 * qt + g < 4 * n128) {
        st_pair[4 + 8 * qt + g] = 0.f;
        st_pair[4 + 4 * n128 + 8 * qt + g] = -INFINITY;
      }
    }
    return;
  }
  const int qend = min(q0 + QB2, len);
  const int nkt = (qend + TB - 1) / TB;
  // a group's diagonal tile = the first it computes (a group of padding past the sequence's end starts with the block:
  // its rows are computed like the others and zeroed at the end)
  const int kd[2] = {min((qg0[0] + 31) / TB, nkt - 1), min((qg0[1] + 31) / TB, nkt - 1)};

  // Q fragments, pre-scaled by log2(e) / sqrt(d): the S products come out in the log2 domain
  RowRegs<T> qf[2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) qf[qs].load(base + (long)min(qg0[qs] + (lane & 31), Tr - 1) * rs, lane);
  const float slope = slopes[h];
  const float slope2 = slope * LOG2E, c2 = SCALE * LOG2E;
  f32x16 o[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};
  // r: the reference the exponentials of a query are taken against (log2 domain).  It is NOT the running maximum: it
  // is set to the first tile's maximum and afterwards only raised when a tile's maximum exceeds it by more than
  // RESCALE_THR, so that most tiles skip the O / row-sum rescale and its exp2 (probabilities stay <= 2^THR; they
  // are floating point, so their relative precision does not depend on the reference).
  float r[2] = {0.f, 0.f}, lp[2] = {0.f, 0.f};     // lp: row sum of THIS lane half only
  constexpr float RESCALE_THR = 8.0f;
  // Everything that is added to q.k enters the S product itself as a FIFTH k-step, so the score needs no VALU
  // instruction between the MFMA and the exp2 (it was one fma per element -- a quarter of the VALU time of a loop that
  // is VALU-bound at head dimension 64) and the chain starts from a zero accumulator (no 16-register ALiBi vector):
  //   k slots 0, 1: K side 1, 1; Q side the per-(query, tile) constant slope2 * (tile start - first query) - r, hi + lo
  //   k slots 2, 3: K side the ALiBi row term slope2 * (key - first key of the 32-key block), hi + lo; Q side 1, 1
  // (two bf16 per constant: 16 mantissa bits, < 1e-3 absolute in the log2 domain).  Lanes 32..63 (k slots 8..15) carry zeros on the Q side.
  bf16x8 kext;
  {
    const float rt = slope2 * (float)(lane & 31);
    const bf16_t rhi = (bf16_t)rt, rlo = (bf16_t)(rt - (float)rhi);
#pragma unroll
    for (int j = 0; j < 8; ++j) kext[j] = (bf16_t)0.0f;
    kext[0] = (bf16_t)1.0f; kext[1] = (bf16_t)1.0f; kext[2] = rhi; kext[3] = rlo;
  }
  const bool lowhalf = lane < 32;
  const bool vq[2] = {qg0[0] + (lane & 31) < len, qg0[1] + (lane & 31) < len};     // this lane's two queries are real rows

  // ---- addressing, hoisted: every LDS read is a loop-invariant per-lane base + the stage offset and every DMA
  // request is a loop-invariant per-lane source offset + the tile's byte offset.  Left to the per-tile helpers this
  // was ~70 address instructions per tile in a loop that is bound by VALU issue.
  int kbase[4];             // K fragment (kb, k-step s): stage + kb * 4096 + kbase[s]               (RowTile::frag)
#pragma unroll
  for (int st = 0; st < 4; ++st) kbase[st] = (lane & 31) * 128 + ((((2 * st + (lane >> 5)) ^ (((lane & 31) >> 1) & 7))) << 4);
  int vbase[2];             // V^T fragment (kb, st, db): stage + 8192 + (32 kb + 16 st) * 128 + vbase[db], + 1024 for its second half (TrTile::frag<true>)
  {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db)
      vbase[db] = (4 * (g >> 1) + q) * 128 + ((db ^ ((q >> 1) & 1)) << 6) + ((16 * (g & 1) + 4 * pp) << 1);
  }
  unsigned dsrc[2][2];      // DMA source offset of this wave's piece j of the K (0) / V (1) image, without the tile offset
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wave + 4 * j) * 8 + (lane >> 3), pos = lane & 7;
    dsrc[0][j] = (unsigned)(((long)row * rs + (pos ^ ((row >> 1) & 7)) * 8) * 2);
    dsrc[1][j] = (unsigned)(((long)row * rs + (((pos >> 2) ^ ((row >> 1) & 1)) * 32) + (pos & 3) * 8) * 2);
  }
  const unsigned tile_bytes = (unsigned)(TB * rs * 2);
  const unsigned smem0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
  const __amdgpu_buffer_rsrc_t rsk = slab_rsrc(base + D, rs, Tr), rsv = slab_rsrc(base + 2 * D, rs, Tr);
  auto issue = [&](int kt, int stage) {
    // (sched & 2: lab switch -- every request reads tile 0, an L2-resident stream; results are wrong)
    const unsigned tb = (sched & 2) ? 0u : (unsigned)kt * tile_bytes, dst = smem0 + stage * STAGE2 + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dma16(rsk, dst + j * 4096, dsrc[0][j] + tb);
      dma16(rsv, dst + 8192 + j * 4096, dsrc[1][j] + tb);
    }
  };
  // Key tiles are swept from the block's last tile DOWN to tile 0: under ALiBi the nearest keys carry the largest
  // bias, so the first tile a group computes (its diagonal tile) fixes the reference and later tiles almost never
  // raise it, and a tile all of whose scores lie more than `skip_thr` below the reference of every query of a group
  // (every probability < 2^-skip_thr of a row sum that is >= 1) is dropped for that group after its S products: no
  // exp2, no PV.  Tile kt sits in ring stage (nkt - 1 - kt) % 3.
  const int t0 = q0 / TB;                       // ring slot of tile kt: (t0 - kt) mod 3 -- the block's first diagonal tile in slot 0
  auto slot_of = [&](int kt) { return (t0 + 3 - kt) % 3; };
  issue(nkt - 1, slot_of(nkt - 1));
  if (nkt > 1) issue(nkt - 2, slot_of(nkt - 2));
  // The Q fragments are ordinary loads the compiler counts: left alone, it waits for them at their first use INSIDE
  // the loop with vmcnt(8..1), which on every later iteration drains the (uncounted) DMA ring.  Retire them here.
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0].f[0]), "+v"(qf[0].f[1]), "+v"(qf[0].f[2]), "+v"(qf[0].f[3]),
               "+v"(qf[1].f[0]), "+v"(qf[1].f[1]), "+v"(qf[1].f[2]), "+v"(qf[1].f[3]) :: "memory");
  // statistics for the backward's window: max |q|^2 over this wave's valid queries (before the pre-scaling)
  float qmax2[2] = {0.f, 0.f};
  if (st_pair) {
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      float part = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const bf16x2 v = {qf[qs].f[st][j], qf[qs].f[st][j + 1]};
          part = __builtin_amdgcn_fdot2_f32_bf16(v, v, part, false);
        }
      const float whole = xhalf_sum(part);
      qmax2[qs] = wave_max(qg0[qs] + (lane & 31) < len ? whole : 0.f);
    }
  }
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[qs].f[st][j] = (bf16_t)((float)qf[qs].f[st][j] * c2);
  // the wave that holds the sequence's last valid query computes every K tile of the pair: it takes max |k|^2 along
  const int g_last = (len - 1 - q0) >> 5;
#if VG_ATTN_PAIR
  const bool kstat = st_pair != nullptr && len - 1 < q0 + QB2 && wave == (g_last < 4 ? g_last : 7 - g_last);
#else
  const bool kstat = st_pair != nullptr && len - 1 < q0 + QB2 && wave == (g_last >> 1);
#endif
  float kmax2 = 0.f;

  // One tile of the sweep: MASK = which of the wave's two groups take part (3: both; 2: the upper group alone, between
  // the two diagonals), J = ring slot -- a compile-time constant in the hot loop, which is unrolled over the ring so that
  // every LDS address of a body is a per-lane base + an immediate; a register in the few tiles of the upper group alone.
  // (Both kinds of body inside ONE loop cost 60 spilled registers, reloaded in the middle of the tile behind counted
  // waits that drain the DMA ring: they live in loops of their own.)
  auto body = [&](auto mc, auto jc, const int kt) __attribute__((always_inline)) {
    constexpr int MASK = decltype(mc)::value;
    const int J = jc;
    const int kv0 = kt * TB;
    if (VG_LAB_ATTN & 64) return;
    const char* k_row = smem + J * STAGE2;
    const char* v_tr = k_row + TB * 128;
    constexpr int Q0 = MASK == 3 ? 0 : 1;         // first group of the loops below

    bf16x8 kf[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < 4; ++st) kf[kb][st] = *reinterpret_cast<const bf16x8*>(k_row + kb * 4096 + kbase[st]);
    f32x16 s[2][2];
#pragma unroll
    for (int qs = Q0; qs < 2; ++qs) {
      const float cb = slope2 * (float)(kv0 - qg0[qs]) - r[qs];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const float cst = cb + (float)(kb * 32) * slope2;
        const bf16_t hi = (bf16_t)cst;
        const bf16_t lo = (bf16_t)(cst - (float)hi);
        bf16x8 qext;
#pragma unroll
        for (int j = 0; j < 8; ++j) qext[j] = (bf16_t)0.0f;
        qext[0] = lowhalf ? hi : (bf16_t)0.0f;
        qext[1] = lowhalf ? lo : (bf16_t)0.0f;
        qext[2] = qext[3] = lowhalf ? (bf16_t)1.0f : (bf16_t)0.0f;
        f32x16 a = Traits<T>::mfma(kext, qext, zero16());
        if (!(VG_LAB_ATTN & 16)) {
#pragma unroll
          for (int st = 0; st < 4; ++st) a = Traits<T>::mfma(kf[kb][st], qf[qs].f[st], a);
        } else {
#pragma unroll
          for (int st = 0; st < 4; ++st) asm volatile("" :: "v"(kf[kb][st]), "v"(qf[qs].f[st]));
        }
        s[qs][kb] = a;
      }
    }
    if (kstat) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float part = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const bf16x2 v = {kf[kb][st][j], kf[kb][st][j + 1]};
            part = __builtin_amdgcn_fdot2_f32_bf16(v, v, part, false);
          }
        kmax2 = fmaxf(kmax2, xhalf_sum(part));
      }
    }
    // a group's diagonal tile (the first it computes): mask key > query.  Only the first group of the body can be on
    // its diagonal: the upper group's lies above the lower group's
    const bool first = kt == kd[Q0];
    if (first) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        // acc_row(i, lane) <= query - key block start, with the lane's part of acc_row moved to the right-hand side
        // (one per-lane value compared against 16 immediates, not 16 hoisted per-lane values)
        const int lim = qg0[Q0] + (lane & 31) - kv0 - kb * 32 - 4 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < 16; ++i) s[Q0][kb][i] = ((i & 3) + 8 * (i >> 2)) <= lim ? s[Q0][kb][i] : -INFINITY;
      }
    }
    // tile maxima relative to the references
    float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int qs = Q0; qs < 2; ++qs)
      mx[qs] = (VG_LAB_ATTN & 32) ? s[qs][0][0] : xhalf_max(fmaxf(max16(s[qs][0]), max16(s[qs][1])));
    // negligible for every VALID query of the wave (padding rows hold other bytes in the packed layout than in the padded
    // one: left in, they made the decision -- hence the valid rows' last bits -- depend on the layout)
    if (!first && !__any((vq[0] && mx[0] > -skip_thr) || (vq[1] && mx[1] > -skip_thr))) return;
    bf16x8 vf[2][2][2];      // [kb][k-step of 16 keys][d block]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* at = v_tr + (32 * kb + 16 * st) * 128 + vbase[db];
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, at));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, at + 1024));
          vf[kb][st][db] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
    for (int qs = Q0; qs < 2; ++qs) {
      const bool fst = kt == kd[qs];          // (two groups of padding past the sequence's end share a first tile)
      // the first tile sets the reference, later ones raise it rarely
      if (fst || __any(mx[qs] > RESCALE_THR)) {
        const float delta = (fst || mx[qs] > RESCALE_THR) ? mx[qs] : 0.f;
        // first tile: O and the row sum are still zero and the maximum may be far below zero (2^-delta overflows)
        const float sc = fst ? 1.0f : __builtin_amdgcn_exp2f(-delta);
        r[qs] += delta;
        lp[qs] *= sc;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[qs][db][i] *= sc;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) s[qs][kb][i] -= delta;
      }
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p0 = (VG_LAB_ATTN & 1) ? s[qs][0][i] : __builtin_amdgcn_exp2f(s[qs][0][i]);
        const float p1 = (VG_LAB_ATTN & 1) ? s[qs][1][i] : __builtin_amdgcn_exp2f(s[qs][1][i]);
        s[qs][0][i] = p0;
        s[qs][1][i] = p1;
        ps0 += p0;
        ps1 += p1;
      }
      lp[qs] += ps0 + ps1;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x8 pb = AccOperand<T>::get(s[qs][kb], st);
          if (!(VG_LAB_ATTN & 2)) {
#pragma unroll
            for (int db = 0; db < 2; ++db) o[qs][db] = Traits<T>::mfma(vf[kb][st][db], pb, o[qs][db]);
          } else {
#pragma unroll
            for (int db = 0; db < 2; ++db) asm volatile("" :: "v"(vf[kb][st][db]), "v"(pb));
          }
        }
    }
  };
  // tile kt has landed (this wave's four pieces; the barrier adds everyone else's) while the next one stays in flight;
  // then the tile after next is requested into the stage of the previous tile, which every wave has finished reading
  // tile kt has landed (this wave's four pieces; the barrier adds everyone else's) while the next one stays in flight;
  // then the tile after next is requested into the slot of the previous tile, which every wave has finished reading
  auto arrive = [&](const int J, const int kt) __attribute__((always_inline)) {
    if (kt > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(VG_LAB_ATTN & 4)) __builtin_amdgcn_s_barrier();
    if (!(VG_LAB_ATTN & 8) && kt >= 2) issue(kt - 2, J >= 1 ? J - 1 : 2);
  };
  // (1) above the lower group's diagonal: nothing, or the upper group alone; (2) the hot loop over both groups starts
  // in slot 0 (waves 0, 1: the block's first diagonal tile) or slot 2 (waves 2, 3: one tile higher; a peeled copy)
  int kt = nkt - 1;
  for (int Jr = slot_of(kt); kt > kd[0]; --kt, Jr = Jr == 2 ? 0 : Jr + 1) {
    arrive(Jr, kt);
    if (kt <= kd[1]) body(std::integral_constant<int, 2>{}, Jr, kt);
  }
  if (slot_of(kt) == 2) {
    arrive(2, kt);
    body(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}, kt);
    --kt;
  }
  if (kt >= 0) {
    for (;;) {
      arrive(0, kt);
      body(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, kt);
      if (--kt < 0) break;
      arrive(1, kt);
      body(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, kt);
      if (--kt < 0) break;
      arrive(2, kt);
      body(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}, kt);
      if (--kt < 0) break;
    }
  }

  // ---- epilogue: O^T -> rows through LDS.  The last tile (tile 0) sits in stage (nkt - 1) % 3; the other two stages
  // hold no tile any wave still reads and nothing is in flight: 2 x 16 KB = 8 KB per wave.
  char* ow = smem + ((t0 + 1 + (wave >> 1)) % NSTAGE2) * STAGE2 + (wave & 1) * 8192;
  float nlmax[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    const float l = xhalf_sum(lp[qs]);
    const int query = qg0[qs] + (lane & 31);
    const float mul = query < len ? 1.f / l : 0.f;
    const int row = qs * 32 + (lane & 31);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + 8 * g + 4 * (lane >> 5);
        bf16x4 v = {(bf16_t)(o[qs][db][4 * g] * mul), (bf16_t)(o[qs][db][4 * g + 1] * mul),
                    (bf16_t)(o[qs][db][4 * g + 2] * mul), (bf16_t)(o[qs][db][4 * g + 3] * mul)};
        if (query >= len) v = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};   // (a padded row may hold inf * 0)
        *reinterpret_cast<bf16x4*>(ow + row * 128 + ((((d0 >> 3) ^ (row & 7))) << 4) + (d0 & 7) * 2) = v;
      }
    const float L2 = r[qs] + log2f(l) - slope2 * (float)(lane & 31);        // log-sum-exp of the row, log2 domain
    if (query < len) {
      if (lane < 32) lse[(long)h * Mtot + soff + query] = L2 * LN2;
      nlmax[qs] = -L2;
    }
  }
  if (st_pair) {
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      const float nl = wave_max(nlmax[qs]);
      const int g = 8 * qt + (qs == 0 ? wave : 7 - wave);       // 32-query group of the sequence
      if (lane == 0 && g < 4 * n128) {
        st_pair[4 + g] = qmax2[qs];
        st_pair[4 + 4 * n128 + g] = nl;
      }
    }
    if (kstat) {
      const float km = wave_max(kmax2);
      if (lane == 0) st_pair[0] = km;
    }
  }
  // the rows of a wave are written and read by that wave only: no block barrier (LDS accesses of one wave stay in order)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3), c16 = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(ow + row * 128 + ((c16 ^ (row & 7)) << 4));
    const int query = qg0[row >> 5] + (row & 31);
    if (query < Tr) *reinterpret_cast<uint4*>(obase + (long)query * D + c16 * 8) = v;
  }
}

// =====================================================================================
// backward: delta = rowsum(dO * O) per (b, h, t)
// =====================================================================================
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta,
                                  int Mtot, int H) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gid & 7;
  const long mh = gid >> 3;
  const int h = mh % H;
  const long m = mh / H;
  float acc = 0.f;
  if (m < (long)Mtot) {
    const long off = m * (long)H * DH + h * DH + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += to_f32<T>(o[off + e]) * to_f32<T>(dout[off + e]);
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  if (c == 0 && m < (long)Mtot) delta[(long)h * Mtot + m] = acc;       // [head][row], like the LSE
}

// =====================================================================================
// backward: dQ  (block owns 128 queries, sweeps key tiles; no atomics)
// p = exp2(raw*c2 + slope2*(key - qw0) - [lse2 + slope2*(query - qw0)]);  dS = p * (dP - delta)
// (the 1/sqrt(d) factor of dS is applied once to the final dQ)
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? VG_ATTN_OCC : 1) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                          const float* __restrict__ lse,
                                                          float* __restrict__ delta, const T* __restrict__ out_o,
                                                          const float* __restrict__ slopes, T* __restrict__ dqkv,
                                                          int Tn, int H, const int* __restrict__ lengths, float skip_thr, int sched,
                                                          const int* __restrict__ cu, int Mtot, const float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_row = smem;
  char* v_row = smem + LdsPlan<T>::ROW_BYTES;
  char* k_tr = smem + 2 * LdsPlan<T>::ROW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (VG_ATTN_PRIO & 2) { const int pr = blockIdx.x % 3; if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); }
  // 1-D grid, longest sweeps first over the whole launch (LPT order); the tiles of one (b, h)
  // are H*B apart, i.e. on the same XCD whenever H*B is a multiple of 8, and share K/V in its L2
  const int nqt = (Tn + QB - 1) / QB, HB = gridDim.x / nqt;
  int hb, rank;
  pair_and_rank(blockIdx.x, nqt, HB, sched, hb, rank);
  const int qt = nqt - 1 - rank, h = head_of_slot(hb, H), b = hb / H;
  const int D = H * DH;
  const long rs = 3L * D;
  const int soff = cu ? cu[b] : b * Tn;                 // first row of the sequence in the [rows][...] tensors
  const int Tr = cu ? cu[b + 1] - soff : Tn;            // rows the sequence owns (packed layout: its own length)
  const int len = lengths ? min(lengths[b], Tr) : Tr;
  const int q0 = qt * QB, qw0 = q0 + wave * 32, query = qw0 + (lane & 31);
  const T* __restrict__ base = qkv + (long)soff * rs + h * DH;
  T* __restrict__ dqbase = dqkv + (long)soff * rs + h * DH;
  if (q0 >= len) {
    f32x16 z[2] = {zero16(), zero16()};
    if (query < Tr) store_rows_T<T>(dqbase + (long)query * rs, z, 0.f, lane);
    if (query < Tr && lane < 32) delta[(long)h * Mtot + soff + query] = 0.f;     // the dK/dV pass reads every row's delta
    return;
  }
  const int qend = min(q0 + QB, len);
  const int nkt = (qend + TB - 1) / TB;
  const bool qvalid = query < len;
  const int qc = min(query, Tr - 1);
  constexpr bool DMA = sizeof(T) == 2;
  constexpr int STAGE = 2 * LdsPlan<T>::ROW_BYTES + (DMA ? 0 : LdsPlan<T>::TR_BYTES);   // bf16: K^T fragments come out of the K row image
  uint4 rk[DMA ? 1 : NVec<T>::v], rv[DMA ? 1 : NVec<T>::v];
  __amdgpu_buffer_rsrc_t rsk, rsv;
  auto issue = [&](int t0, char* st) {
    slab_dma<false, BSW>(rsk, st, rs, t0, wave, lane);
    slab_dma<false, BSW>(rsv, st + LdsPlan<T>::ROW_BYTES, rs, t0, wave, lane);      // (K^T fragments come out of the K row image)
  };
  // Round 5: key tiles are swept from the block's diagonal DOWN, and the first tile is requested before anything else is
  // loaded: the row fragments, the statistics of the window and the first K / V tile share ONE memory round trip per
  // block (they were three in a row: delta's operands, then the window's statistics, then the first tile).
  if constexpr (DMA) {
    rsk = slab_rsrc(reinterpret_cast<const bf16_t*>(base + D), rs, Tr);
    rsv = slab_rsrc(reinterpret_cast<const bf16_t*>(base + 2 * D), rs, Tr);
    issue((nkt - 1) * TB, smem + ((nkt - 1) & 1) * STAGE);
  } else {
    slab_load<T>(rk, base + D, rs, (nkt - 1) * TB, Tr, tid);
    slab_load<T>(rv, base + 2 * D, rs, (nkt - 1) * TB, Tr, tid);
  }
  // every load of the prologue is requested before the first value is used (one round trip: the lse value used to be
  // waited for -- with everything in front of it -- before the O rows were even requested)
  RowRegs<T> qf, dof, of;
  qf.load(base + (long)qc * rs, lane);
  dof.load(dout + ((long)soff + qc) * D + h * DH, lane);
  of.load(out_o + ((long)soff + qc) * D + h * DH, lane);
  const float lse_raw = lse[(long)h * Mtot + soff + qc];
  // ALiBi window (round 5, see attn_stats_floats): keys further than W frames below the block's first query carry
  // probabilities < 2^-skip_thr for every query of the block -- their tiles are not streamed at all
  const float slope = slopes[h];
  const float slope2 = slope * LOG2E, c2 = SCALE * LOG2E;
  int kt_lo = 0;
  if (DMA && stats != nullptr && skip_thr < INFINITY) {
    const int n128 = (Tn + QB - 1) / QB;          // this block's own four 32-query groups
    const float W = attn_window(stats + (long)(b * H + h) * attn_stats_floats(Tn), n128, 4 * (q0 / QB), 4, slope2, c2, skip_thr);
    const float lo = (float)q0 - W;
    if (lo > 0.f) kt_lo = min((int)(lo * (1.0f / TB)), q0 / TB);
  }
  // delta = rowsum(dO * O) of this wave's 32 queries, from the dO row fragments it holds anyway and the matching O row
  // fragments (round 4: the stand-alone delta kernel -- a 15 us launch per layer that re-read O and dO -- is gone);
  // written out for the dK/dV pass, which runs after this launch.  Lanes l and l + 32 hold the two halves of a row.
  float dl;
  {
    float part = 0.f;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) part += (float)dof.f[s][e] * (float)of.f[s][e];
    } else {
#pragma unroll
      for (int s = 0; s < 32; ++s) part += dof.f[s] * of.f[s];
    }
    const float whole = xhalf_sum(part);       // cross-lane: every lane takes part
    dl = qvalid ? whole : 0.f;
    if (query < Tr && lane < 32) delta[(long)h * Mtot + soff + query] = dl;
  }
  // +inf for padded queries -> p = exp2(-inf) = 0 without a compare
  const float Lq = qvalid ? lse_raw * LOG2E + slope2 * (float)(query - qw0) : INFINITY;
  f32x16 kinit, dinit;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    kinit[i] = slope * 8.0f * (float)acc_row(i, lane);
    dinit[i] = -dl;
  }
  // Round 6 (bf16): everything that was added to or multiplied into the products per element now rides in the products
  // themselves, as in the forward (attn2_fwd_kernel).  Q is pre-scaled by log2(e) / sqrt(d) -- the very rounding the
  // forward applied, so the recomputed scores are the forward's -- and a FIFTH k-step carries the additive terms:
  //   S':  K side (1, 1, row term hi, lo), Q side (c hi, c lo, 1, 1) with c = slope2 (tile start - first query) - L_q
  //   dP': K side the same registers,      dO side (-delta hi, -delta lo, 0, 0) -- loop-invariant
  // (two bf16 per constant: 16 significant bits).  Per 32 x 32 block that is 2 more MFMAs on a pipe that was 18 % busy and
  // 16 fma + 16 v_mov (the compiler rebuilt the 16-register -delta vector for every block) fewer on the issue port that
  // bounds the kernel: the census (tools/isa_census.py) went from 146 to 110 vector instructions per block.
  [[maybe_unused]] bf16x8 kext, dext;
  [[maybe_unused]] float Lfin = 0.f;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf.f[st][j] = (bf16_t)((float)qf.f[st][j] * c2);
    const float rt = slope2 * (float)(lane & 31);
    const bf16_t rhi = (bf16_t)rt, rlo = (bf16_t)(rt - (float)rhi);
    const float nd = -dl;
    const bf16_t dhi = (bf16_t)nd, dlo = (bf16_t)(nd - (float)dhi);
#pragma unroll
    for (int j = 0; j < 8; ++j) { kext[j] = (bf16_t)0.0f; dext[j] = (bf16_t)0.0f; }
    kext[0] = (bf16_t)1.0f; kext[1] = (bf16_t)1.0f; kext[2] = rhi; kext[3] = rlo;
    if (lane < 32) { dext[0] = dhi; dext[1] = dlo; }
    Lfin = qvalid ? Lq : 1.0e30f;          // (finite: the hi / lo split of an infinity is a NaN; 2^-1e30 is 0 all the same)
  }

  f32x16 dq[2] = {zero16(), zero16()};
  for (int kt = nkt - 1; kt >= kt_lo; --kt) {
    const int kv0 = kt * TB;
    if constexpr (DMA) {
      dma_wait_and_publish();
      if (!(VG_LAB_ATTN & 512) && kt > kt_lo) issue(kv0 - TB, smem + ((kt - 1) & 1) * STAGE);
      k_row = smem + (kt & 1) * STAGE;
      v_row = k_row + LdsPlan<T>::ROW_BYTES;
      k_tr = k_row + 2 * LdsPlan<T>::ROW_BYTES;
    } else {
      __syncthreads();
      slab_store<T, true, true>(rk, k_row, k_tr, tid);
      slab_store<T, true, false>(rv, v_row, nullptr, tid);
      __syncthreads();
      if (kt > kt_lo) {
        slab_load<T>(rk, base + D, rs, kv0 - TB, Tr, tid);
        slab_load<T>(rv, base + 2 * D, rs, kv0 - TB, Tr, tid);
      }
    }
    if (qw0 + 31 < kv0) continue;
    if (VG_LAB_ATTN & 256) continue;
    const bool diag = kv0 + TB - 1 > qw0;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
      if constexpr (sizeof(T) == 2) {
        const float cst = slope2 * (float)(kv0 + kb * 32 - qw0) - Lfin;
        const bf16_t hi = (bf16_t)cst, lo = (bf16_t)(cst - (float)hi);
        bf16x8 qext;
#pragma unroll
        for (int j = 0; j < 8; ++j) qext[j] = (bf16_t)0.0f;
        if (lane < 32) { qext[0] = hi; qext[1] = lo; qext[2] = (bf16_t)1.0f; qext[3] = (bf16_t)1.0f; }
        s = mma_row_regs<T, BSW>(k_row, kb * 32 + (lane & 31), qf, lane, Traits<T>::mfma(kext, qext, zero16()));
      } else {
        s = mma_row_regs<T>(k_row, kb * 32 + (lane & 31), qf, lane, kinit);
      }
      if (diag) {
        const int lim = query - kv0 - kb * 32;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = acc_row(i, lane) <= lim ? s[i] : -INFINITY;
      }
      const float off = slope2 * (float)(kv0 + kb * 32 - qw0) - Lq;
      // every probability of this 32-key block below 2^-skip_thr (of a row that sums to 1) for all 32 queries of the
      // wave: its dS is dropped -- no dP product, no exp2, no dQ product (ALiBi: the far keys of the steep heads)
      if constexpr (sizeof(T) == 2) {
        if (VG_ATTN_TILESKIP && !__any(max16(s) > -skip_thr)) continue;
        dp = mma_row_regs<T, BSW>(v_row, kb * 32 + (lane & 31), dof, lane, Traits<T>::mfma(kext, dext, zero16()));
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = fexp2<T>(s[i]) * dp[i];
      } else {
        if (VG_ATTN_TILESKIP && !__any(fmaf(max16(s), c2, off) > -skip_thr)) continue;
        dp = mma_row_regs<T>(v_row, kb * 32 + (lane & 31), dof, lane, dinit);
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = fexp2<T>(fmaf(s[i], c2, off)) * dp[i];
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        if constexpr (DMA) dq[db] = mma_tr_acc_rowimg<BSW>(k_row, kb * 32, db, s, lane, dq[db]);
        else dq[db] = mma_tr_acc<T>(k_tr, kb * 32, db, s, lane, dq[db]);
      }
    }
  }
  if constexpr (DMA) {
    __syncthreads();                                   // every wave is done with the last K / V tile: the stages are free
    store_rows_T_lds(dqbase + (long)qw0 * rs, rs, min(32, Tr - qw0), dq, SCALE, smem + wave * 4096, lane);
  } else {
    if (query < Tr) store_rows_T<T>(dqbase + (long)query * rs, dq, SCALE, lane);
  }
}

// =====================================================================================
// backward: dK, dV (block owns 128 keys, sweeps query tiles; no atomics)
// rows of the accumulator = queries: the per-query constants -(lse2 + slope2*(q - k0))/c2 and
// -delta are the INITIAL accumulators of the S and dP products (staged in LDS per query tile;
// -inf for padded queries), the per-key constant slope2*(key - k0) sits on the lane.
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? VG_ATTN_OCC : 1) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta,
                                                           const float* __restrict__ slopes, T* __restrict__ dqkv,
                                                           int Tn, int H, const int* __restrict__ lengths, float skip_thr, int sched,
                                                           const int* __restrict__ cu, int Mtot, const float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* q_row = smem;
  char* do_row = smem + LdsPlan<T>::ROW_BYTES;
  char* q_tr = smem + 2 * LdsPlan<T>::ROW_BYTES;
  char* do_tr = q_tr + LdsPlan<T>::TR_BYTES;
  float* st = reinterpret_cast<float*>(do_tr + LdsPlan<T>::TR_BYTES);   // [2][64]: S init, dP init
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (VG_ATTN_PRIO & 2) { const int pr = blockIdx.x % 3; if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); }
  const int nkb = (Tn + QB - 1) / QB, HB = gridDim.x / nkb;   // low key tiles sweep the most query tiles: first
  int hb, ktile;
  pair_and_rank(blockIdx.x, nkb, HB, sched, hb, ktile);
  const int h = head_of_slot(hb, H), b = hb / H;
  const int D = H * DH;
  const long rs = 3L * D;
  const int soff = cu ? cu[b] : b * Tn;                 // first row of the sequence in the [rows][...] tensors
  const int Tr = cu ? cu[b + 1] - soff : Tn;            // rows the sequence owns (packed layout: its own length)
  const int len = lengths ? min(lengths[b], Tr) : Tr;
  const int k0 = ktile * QB, kw0 = k0 + wave * 32, key = kw0 + (lane & 31);
  const T* __restrict__ base = qkv + (long)soff * rs + h * DH;
  const T* __restrict__ dobase = dout + (long)soff * D + h * DH;
  T* __restrict__ dkbase = dqkv + (long)soff * rs + D + h * DH;
  T* __restrict__ dvbase = dqkv + (long)soff * rs + 2 * D + h * DH;
  f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
  if (k0 >= len) {
    if (key < Tr) {
      store_rows_T<T>(dkbase + (long)key * rs, dk, 0.f, lane);
      store_rows_T<T>(dvbase + (long)key * rs, dv, 0.f, lane);
    }
    return;
  }
  const int kc = min(key, Tr - 1);
  const float* __restrict__ lse_bh = lse + (long)h * Mtot + soff;
  const float* __restrict__ dl_bh = delta + (long)h * Mtot + soff;
  // Round 5: query tiles are swept from the block's diagonal UP (they were swept down from the sequence's last tile so
  // that the key blocks of a pair started on one tile -- L2 hits that never bounded these kernels), so that the first
  // tile is known before the window is: it is requested first, and the K / V row fragments, the first two tiles'
  // lse / delta and the window's statistics all share its round trip (they were three in a row).
  const int qt_beg = k0 / TB, qt_end = (len + TB - 1) / TB;
  constexpr bool DMA = sizeof(T) == 2;
  constexpr int IMG = 2 * LdsPlan<T>::ROW_BYTES + (DMA ? 0 : 2 * LdsPlan<T>::TR_BYTES);   // bf16: Q^T / dO^T fragments come out of the row images
  constexpr int STAGE = IMG + 2 * 64 * (int)sizeof(float);     // + the per-query constants of the tile
  uint4 rq[DMA ? 1 : NVec<T>::v], rd[DMA ? 1 : NVec<T>::v];
  __amdgpu_buffer_rsrc_t rsq, rsd;
  auto issue = [&](int t0, char* sg) {
    slab_dma<false, BSW>(rsq, sg, rs, t0, wave, lane);
    slab_dma<false, BSW>(rsd, sg + LdsPlan<T>::ROW_BYTES, D, t0, wave, lane);       // (Q^T / dO^T fragments come out of the row images)
  };
  if constexpr (DMA) {
    rsq = slab_rsrc(reinterpret_cast<const bf16_t*>(base), rs, Tr);
    rsd = slab_rsrc(reinterpret_cast<const bf16_t*>(dobase), D, Tr);
    issue(qt_beg * TB, smem);
  } else {
    slab_load<T>(rq, base, rs, qt_beg * TB, Tr, tid);
    slab_load<T>(rd, dobase, D, qt_beg * TB, Tr, tid);
  }
  RowRegs<T> kf, vf;
  kf.load(base + D + (long)kc * rs, lane);
  vf.load(base + 2 * D + (long)kc * rs, lane);
  const float slope = slopes[h];
  const float slope2 = slope * LOG2E, c2 = SCALE * LOG2E;
  const float kl = slope2 * (float)(key - k0);

  // per-query constants of a tile: S init = -(lse2 + slope2 (q - k0)) / c2 (-inf for padded queries), dP init = -delta
  auto st_values = [&](int qs, float& a, float& d) {
    const int qq = qs + tid;
    const bool ok = tid < 64 && qq < len;
    a = ok ? -(lse_bh[qq] * LOG2E + slope2 * (float)(qq - k0)) / c2 : -INFINITY;
    d = ok ? -dl_bh[qq] : 0.f;
  };
  // bf16 / LDS-DMA path: wave 0 fetches the RAW lse / delta of a tile two iterations ahead and turns them into the
  // constants only when it stores them (one iteration later, behind the loop's own vmcnt(0)).  Any arithmetic on the
  // loaded value in the iteration that issued the load makes the compiler wait for it right there -- behind the
  // eight LDS-DMA requests of the next tile, i.e. wave 0 sat out the whole transfer once per tile.
  auto st_raw = [&](int qs, float& a, float& d) {
    const int qq = min(qs + tid, Tr - 1);
    a = lse_bh[qq];
    d = dl_bh[qq];
  };
  auto st_store = [&](int qs, float a_raw, float d_raw, float* dst) {      // (the bf16 / LDS-DMA path only)
    const int qq = qs + tid;
    const bool ok = qq < len;
    // round 6: K is pre-scaled by c2 (below), so the row constant is in log2 units already
    dst[tid] = ok ? -(a_raw * LOG2E + slope2 * (float)(qq - k0)) : -INFINITY;
    dst[64 + tid] = ok ? -d_raw : 0.f;
  };
  float st_a = 0.f, st_d = 0.f;    // wave 0: raw lse / delta of the NEXT tile, fetched one iteration ahead
  float st_a0 = 0.f, st_d0 = 0.f;  // ... and of the first tile
  if (DMA && tid < 64) {
    st_raw(qt_beg * TB, st_a0, st_d0);
    st_raw((qt_beg + 1) * TB, st_a, st_d);          // (clamped to the sequence's rows: harmless when there is no second tile)
  }
  // ALiBi window (round 5, see attn_stats_floats): queries further than W frames above the block's last key see its
  // keys with probabilities < 2^-skip_thr -- their tiles are not streamed at all
  int qt_first = qt_end - 1;         // the LAST tile of the upward sweep
  if (DMA && stats != nullptr && skip_thr < INFINITY) {
    const int n128 = (Tn + QB - 1) / QB;
    const float W = attn_window(stats + (long)(b * H + h) * attn_stats_floats(Tn), n128, 0, 4 * n128, slope2, c2, skip_thr);
    const float hi = (float)(k0 + QB - 1) + W;
    if (hi < (float)(qt_end * TB)) qt_first = min(qt_first, (int)(hi * (1.0f / TB)));
  }
  if constexpr (DMA) {
    // retire the counted prologue loads (K / V fragments, the first constants) here: left to the compiler, their
    // waits land at the first use inside the loop as vmcnt(7..0) and drain the uncounted DMA ring on every iteration
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(kf.f[0]), "+v"(kf.f[1]), "+v"(kf.f[2]), "+v"(kf.f[3]), "+v"(vf.f[0]),
                 "+v"(vf.f[1]), "+v"(vf.f[2]), "+v"(vf.f[3]), "+v"(st_a), "+v"(st_d), "+v"(st_a0), "+v"(st_d0) :: "memory");
    if (tid < 64) st_store(qt_beg * TB, st_a0, st_d0, reinterpret_cast<float*>(smem + IMG));
  }
  // Round 6 (bf16), as in the dQ kernel: K pre-scaled by log2(e) / sqrt(d) and the per-key ALiBi term slope2 (key - k0) in a
  // fifth k-step of the S product (query side 1, 1; key side hi, lo -- both loop-invariant), so that the score leaves the
  // MFMA chain ready for exp2: 16 fma per 32 x 32 block fewer (84 -> 68 vector instructions next to 17 MFMAs).
  // (the all-lanes constant (1, 1, 0 ...) operand is rebuilt from immediates at each use -- four v_mov -- instead of held:
  // with it the kernel needs 170 registers, two more than three blocks per CU allow, and a spilled register is reloaded
  // through a counted vmcnt wait that drains the LDS-DMA ring)
  auto one_ext_now = [&]() {
    unsigned a, z0, z1, z2;
    asm volatile("v_mov_b32 %0, 0x3f803f80\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(a), "=v"(z0), "=v"(z1), "=v"(z2));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 w = {a, z0, z1, z2};
    return __builtin_bit_cast(bf16x8, w);
  };
  [[maybe_unused]] bf16x8 kl_ext;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int st_ = 0; st_ < 4; ++st_)
#pragma unroll
      for (int j = 0; j < 8; ++j) kf.f[st_][j] = (bf16_t)((float)kf.f[st_][j] * c2);
    const bf16_t khi = (bf16_t)kl, klo = (bf16_t)(kl - (float)khi);
#pragma unroll
    for (int j = 0; j < 8; ++j) kl_ext[j] = (bf16_t)0.0f;
    if (lane < 32) { kl_ext[0] = khi; kl_ext[1] = klo; }
  }
  const int nq = qt_first - qt_beg + 1;
  for (int it = 0; it < nq; ++it) {
    const int qt = qt_beg + it;
    const int qs0 = qt * TB;
    if constexpr (DMA) {
      dma_wait_and_publish();
      const int sl = it & 1;
      if (it + 1 < nq) {
        char* nx = smem + (sl ^ 1) * STAGE;
        if (tid < 64)                // registers were filled one iteration ago: no wait behind the DMA below
          st_store(qs0 + TB, st_a, st_d, reinterpret_cast<float*>(nx + IMG));
        if (!(VG_LAB_ATTN & 512)) issue(qs0 + TB, nx);
        if (tid < 64 && it + 2 < nq) st_raw(qs0 + 2 * TB, st_a, st_d);
      }
      q_row = smem + sl * STAGE;
      do_row = q_row + LdsPlan<T>::ROW_BYTES;
      st = reinterpret_cast<float*>(q_row + IMG);
    } else {
      __syncthreads();
      slab_store<T, true, true>(rq, q_row, q_tr, tid);
      slab_store<T, true, true>(rd, do_row, do_tr, tid);
      if (tid < 64) st_values(qs0, st[tid], st[64 + tid]);
      __syncthreads();
      if (it + 1 < nq) {
        slab_load<T>(rq, base, rs, qs0 + TB, Tr, tid);
        slab_load<T>(rd, dobase, D, qs0 + TB, Tr, tid);
      }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qb0 = qs0 + qb * 32;
      if (qb0 + 31 < kw0) continue;   // all queries precede this wave's keys
      if (VG_LAB_ATTN & 256) continue;
      f32x16 s;
      if constexpr (sizeof(T) == 2)
        s = mma_row_regs<T, BSW>(q_row, qb * 32 + (lane & 31), kf, lane, Traits<T>::mfma(one_ext_now(), kl_ext, rows16(st, qb * 32, lane)));
      else
        s = mma_row_regs<T>(q_row, qb * 32 + (lane & 31), kf, lane, rows16(st, qb * 32, lane));
      if (qb0 < kw0 + 31) {            // diagonal block: mask query < key
        const int lim = key - qb0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = acc_row(i, lane) >= lim ? s[i] : -INFINITY;
      }
      // all 32 x 32 probabilities of this block below 2^-skip_thr: no dP product, no exp2, no dV / dK products
      if constexpr (sizeof(T) == 2) {
        if (VG_ATTN_TILESKIP && !__any(max16(s) > -skip_thr)) continue;
      } else {
        if (VG_ATTN_TILESKIP && !__any(fmaf(max16(s), c2, kl) > -skip_thr)) continue;
      }
      f32x16 dp = mma_row_regs<T, (sizeof(T) == 2 ? BSW : 0)>(do_row, qb * 32 + (lane & 31), vf, lane, rows16(st + 64, qb * 32, lane));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if constexpr (sizeof(T) == 2) s[i] = fexp2<T>(s[i]);
        else s[i] = fexp2<T>(fmaf(s[i], c2, kl));
        dp[i] *= s[i];
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        if constexpr (DMA) {
          dv[db] = mma_tr_acc_rowimg<BSW>(do_row, qb * 32, db, s, lane, dv[db]);
          dk[db] = mma_tr_acc_rowimg<BSW>(q_row, qb * 32, db, dp, lane, dk[db]);
        } else {
          dv[db] = mma_tr_acc<T>(do_tr, qb * 32, db, s, lane, dv[db]);
          dk[db] = mma_tr_acc<T>(q_tr, qb * 32, db, dp, lane, dk[db]);
        }
      }
    }
  }
  if constexpr (DMA) {
    __syncthreads();                                   // every wave is done with the last Q / dO tile: the stages are free
    store_rows_T_lds(dkbase + (long)kw0 * rs, rs, min(32, Tr - kw0), dk, SCALE, smem + wave * 8192, lane);
    store_rows_T_lds(dvbase + (long)kw0 * rs, rs, min(32, Tr - kw0), dv, 1.f, smem + wave * 8192 + 4096, lane);
  } else if (key < Tr) {
    store_rows_T<T>(dkbase + (long)key * rs, dk, SCALE, lane);
    store_rows_T<T>(dvbase + (long)key * rs, dv, 1.f, lane);
  }
}

// =====================================================================================
// decode: one query row per (b, h) against a pre-allocated cache (HBM-bound)
// one wave per (b, h); lanes stride over cache rows, 64-d dot per lane.
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(64) void attn_decode_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                         const T* __restrict__ vc, T* __restrict__ out,
                                                         const float* __restrict__ slopes,
                                                         const int* __restrict__ pos, int Tmax, int H) {
  __shared__ float red[64][DH + 1];
  const int lane = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
  const int D = H * DH;
  const int n = pos[b];
  float qv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) qv[d] = to_f32<T>(q[(long)b * D + h * DH + d]);
  const float slope = slopes[h];
  float m = -INFINITY, l = 0.f, acc[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) acc[d] = 0.f;
  for (int j = lane; j < n; j += 64) {
    const T* kr = kc + ((long)b * Tmax + j) * D + h * DH;
    const T* vr = vc + ((long)b * Tmax + j) * D + h * DH;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) s += qv[d] * to_f32<T>(kr[d]);
    s = s * SCALE - slope * (float)(n - 1 - j);
    const float mn = fmaxf(m, s);
    const float a = expf(m - mn), p = expf(s - mn);
    l = l * a + p;
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = acc[d] * a + p * to_f32<T>(vr[d]);
    m = mn;
  }
  const float mg = wave_max(m);
  const float w = (m == -INFINITY) ? 0.f : expf(m - mg);
  const float lg = wave_sum(l * w);
#pragma unroll
  for (int d = 0; d < DH; ++d) red[lane][d] = acc[d] * w;
  __syncthreads();
  float o = 0.f;
  for (int r = 0; r < 64; ++r) o += red[r][lane];
  out[(long)b * D + h * DH + lane] = from_f32<T>(o / lg);
}

// integer switches for A/B measurements, read on every launch (cheap: a getenv per launch is ~50 ns)
static int attn_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// Which bf16 forward kernel: attn2_fwd_kernel works on blocks of 256 queries, the round-1/2 kernel on blocks of 128.  When
// the last 256-query block of a sequence would be at most half full, or when 256-query blocks do not give every CU two
// blocks, the 128-query kernel is the faster one (B = 16, H = 16; us per launch, 256 / 128: T = 250 12.4 / 9.8, 384
// 21.4 / 15.7, 640 37.6 / 32.7, 896 57.3 / 52.3, 1150 81.9 / 76.3 -- and 500 23.8 / 24.7, 1000 59.4 / 63.2, 1280
// 85.4 / 94.6 the other way).  VG_ATTN_V1=1 / 0 forces one of them (A/B measurements).
static bool attn_v1(int Tn, long bh) {
  static const int v = [] { const char* e = getenv("VG_ATTN_V1"); return e ? atoi(e) : -1; }();
  if (v >= 0) return v != 0;
  const int w256 = (Tn + 255) / 256 * 256 - Tn, w128 = (Tn + 127) / 128 * 128 - Tn;
  return w256 - w128 >= 128 || (long)((Tn + 255) / 256) * bh < 512;
}

// Tiles whose every probability is below 2^-thr of its row's sum are dropped by the bf16 kernels (ALiBi makes the far
// key tiles of the steep heads negligible).  VG_ATTN_SKIP=<thr> changes the threshold, VG_ATTN_SKIP=0 keeps every tile.
static float attn_skip_thr() {
  static const float v = [] {
    const char* e = getenv("VG_ATTN_SKIP");
    const float t = e ? (float)atof(e) : 20.0f;
    return t > 0.f ? t : INFINITY;
  }();
  return v;
}

// `cu` (packed rows, vg_attn_*_varlen): cu[b] = first row of sequence b in the [rows][...] tensors, cu[B] = their total;
// Mtot = rows of those tensors (= B * Tn in the padded layout); lse / delta are [H][Mtot].
// host-side mirror of attn_stats_floats (the public vg_attn_stats_floats)
static int stats_floats_host(int Tn) { return 4 + 8 * ((Tn + QB - 1) / QB); }

template <typename T>
int launch_fwd(const void* qkv, void* out, float* lse, const float* slopes, int B, int Tn, int H,
               const int32_t* lengths, const int32_t* cu, int Mtot, float* stats, hipStream_t stream) {
  // algorithmic work: causal-exact QK^T + PV, 2*2*64 FLOP per (query, key <= query) pair
  const int tok = vg_host::prof_begin(VG_PROF_ATTN_FWD, 256.0 * B * H * (0.5 * Tn * (Tn + 1.0)), stream);
  const int sched = attn_env("VG_ATTN_SCHED", 0);
  if constexpr (sizeof(T) == 2) {
    if (!attn_v1(Tn, (long)B * H)) {
      dim3 grid2(((Tn + QB2 - 1) / QB2) * H * B);
      // lab: VG_ATTN_LDS_EXTRA=<bytes> of unused dynamic LDS per block lowers the blocks a CU holds (occupancy A/B runs)
      static const int extra = attn_env("VG_ATTN_LDS_EXTRA", 0);
      static bool attr2 = false;
      if (extra > 0 && !attr2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn2_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  NSTAGE2 * STAGE2 + extra);
        attr2 = true;
      }
      VG_LAUNCH(attn2_fwd_kernel, grid2, dim3(256), NSTAGE2 * STAGE2 + extra, stream, (const bf16_t*)qkv, (bf16_t*)out,
                         lse, slopes, Tn, H, lengths, attn_skip_thr(), sched, cu, Mtot, stats);
      vg_host::prof_end(tok, stream);
      return vg_host::check_launch("vg_attn_fwd");
    }
  }
  const size_t lds = (sizeof(T) == 2 ? 2 : 1) * (LdsPlan<T>::ROW_BYTES + LdsPlan<T>::TR_BYTES);
  dim3 grid(((Tn + QB - 1) / QB) * H * B);
  VG_LAUNCH(attn_fwd_kernel<T>, grid, dim3(256), lds, stream, (const T*)qkv, (T*)out, lse, slopes, Tn, H,
                     lengths, sched, cu, Mtot, sizeof(T) == 2 ? stats : nullptr);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_attn_fwd");
}

template <typename T>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const float* slopes,
               void* dqkv, float* delta, int B, int Tn, int H, const int32_t* lengths, const int32_t* cu, int Mtot,
               const float* stats, hipStream_t stream) {
  // algorithmic work of the backward: 2 x forward (training = 3 x forward, no credit for the scores the
  // backward recomputes; the kernels execute 5 products -- S, dP, dV, dK, dQ -- i.e. 2.5 x)
  const int tok = vg_host::prof_begin(VG_PROF_ATTN_BWD, 512.0 * B * H * (0.5 * Tn * (Tn + 1.0)), stream);
  // (the stand-alone attn_delta_kernel is no longer launched: the dQ pass computes and stores delta)
  dim3 grid(((Tn + QB - 1) / QB) * H * B);
  const int sched = attn_env("VG_ATTN_SCHED", 0);
  if (attn_env("VG_ATTN_WINDOW", 1) == 0) stats = nullptr;      // A/B switch: the round-4 sweep over every tile
  static const int extra = attn_env("VG_ATTN_LDS_EXTRA", 0);      // lab: see launch_fwd
  const size_t lds_q = (sizeof(T) == 2 ? 2 * (2 * LdsPlan<T>::ROW_BYTES) : 2 * LdsPlan<T>::ROW_BYTES + LdsPlan<T>::TR_BYTES) + extra;
  static bool attrq[2] = {false, false};
  if (extra > 0 && !attrq[sizeof(T) == 2]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    attrq[sizeof(T) == 2] = true;
  }
  const float skip = sizeof(T) == 2 ? attn_skip_thr() : INFINITY;     // the fp32 parity path keeps every tile
  VG_LAUNCH(attn_bwd_dq_kernel<T>, grid, dim3(256), lds_q, stream, (const T*)qkv, (const T*)dout, lse,
                     delta, (const T*)out, slopes, (T*)dqkv, Tn, H, lengths, skip, sched, cu, Mtot, stats);
  const size_t lds_k = (sizeof(T) == 2 ? 2 * (2 * LdsPlan<T>::ROW_BYTES + 2 * 64 * sizeof(float))
                                       : 2 * LdsPlan<T>::ROW_BYTES + 2 * LdsPlan<T>::TR_BYTES + 2 * 64 * sizeof(float)) + extra;
  static bool attr[2] = {false, false};
  if (!attr[sizeof(T) == 2]) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<T>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
    attr[sizeof(T) == 2] = true;
  }
  VG_LAUNCH(attn_bwd_dkv_kernel<T>, grid, dim3(256), lds_k, stream, (const T*)qkv, (const T*)dout, lse,
                     delta, slopes, (T*)dqkv, Tn, H, lengths, skip, sched, cu, Mtot, stats);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_attn_bwd");
}

}  // namespace

extern "C" int vg_attn_stats_floats(int B, int T, int H) { return B * H * stats_floats_host(T); }

extern "C" int vg_attn_fwd_stats(const void* qkv, void* out, float* lse, const float* slopes, int B, int T, int H,
                                 const int32_t* lengths, const int32_t* cu_rows, int rows, float* stats, int dtype,
                                 hipStream_t stream) {
  VG_REQUIRE(B > 0 && T > 0 && H > 0, "vg_attn_fwd: empty problem");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_attn_fwd: bad dtype %d", dtype);
  VG_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0, "vg_attn_fwd: unaligned");
  VG_REQUIRE(cu_rows == nullptr || (lengths != nullptr && rows > 0), "vg_attn_fwd: packed rows need lengths[B], cu_rows[B + 1] and rows");
  const int Mtot = cu_rows ? rows : B * T;
  VG_REQUIRE((long)Mtot < 0x7fffffffL / 4, "vg_attn_fwd: too many rows");
  if (dtype == VG_BF16) return launch_fwd<bf16_t>(qkv, out, lse, slopes, B, T, H, lengths, cu_rows, Mtot, stats, stream);
  return launch_fwd<float>(qkv, out, lse, slopes, B, T, H, lengths, cu_rows, Mtot, nullptr, stream);
}

extern "C" int vg_attn_bwd_stats(const void* qkv, const void* out, const void* dout, const float* lse,
                                 const float* slopes, void* dqkv, float* delta, int B, int T, int H,
                                 const int32_t* lengths, const int32_t* cu_rows, int rows, const float* stats, int dtype,
                                 hipStream_t stream) {
  VG_REQUIRE(B > 0 && T > 0 && H > 0, "vg_attn_bwd: empty problem");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_attn_bwd: bad dtype %d", dtype);
  VG_REQUIRE(cu_rows == nullptr || (lengths != nullptr && rows > 0), "vg_attn_bwd: packed rows need lengths[B], cu_rows[B + 1] and rows");
  const int Mtot = cu_rows ? rows : B * T;
  if (dtype == VG_BF16)
    return launch_bwd<bf16_t>(qkv, out, dout, lse, slopes, dqkv, delta, B, T, H, lengths, cu_rows, Mtot, stats, stream);
  return launch_bwd<float>(qkv, out, dout, lse, slopes, dqkv, delta, B, T, H, lengths, cu_rows, Mtot, nullptr, stream);
}

extern "C" int vg_attn_fwd(const void* qkv, void* out, float* lse, const float* slopes, int B, int T, int H,
                           const int32_t* lengths, int dtype, hipStream_t stream) {
  return vg_attn_fwd_stats(qkv, out, lse, slopes, B, T, H, lengths, nullptr, 0, nullptr, dtype, stream);
}

extern "C" int vg_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                           const float* slopes, void* dqkv, float* delta, int B, int T, int H,
                           const int32_t* lengths, int dtype, hipStream_t stream) {
  return vg_attn_bwd_stats(qkv, out, dout, lse, slopes, dqkv, delta, B, T, H, lengths, nullptr, 0, nullptr, dtype, stream);
}

extern "C" int vg_attn_fwd_varlen(const void* qkv, void* out, float* lse, const float* slopes, int B, int Tmax, int H,
                                  const int32_t* lengths, const int32_t* cu_rows, int rows, int dtype, hipStream_t stream) {
  VG_REQUIRE(lengths != nullptr && cu_rows != nullptr && rows > 0, "vg_attn_fwd_varlen: lengths[B], cu_rows[B + 1] and rows are required");
  return vg_attn_fwd_stats(qkv, out, lse, slopes, B, Tmax, H, lengths, cu_rows, rows, nullptr, dtype, stream);
}

extern "C" int vg_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse,
                                  const float* slopes, void* dqkv, float* delta, int B, int Tmax, int H,
                                  const int32_t* lengths, const int32_t* cu_rows, int rows, int dtype, hipStream_t stream) {
  VG_REQUIRE(lengths != nullptr && cu_rows != nullptr && rows > 0, "vg_attn_bwd_varlen: lengths[B], cu_rows[B + 1] and rows are required");
  return vg_attn_bwd_stats(qkv, out, dout, lse, slopes, dqkv, delta, B, Tmax, H, lengths, cu_rows, rows, nullptr, dtype, stream);
}

extern "C" int vg_attn_decode(const void* q, const void* kcache, const void* vcache, void* out,
                              const float* slopes, const int32_t* pos, int B, int Tmax, int H, int dtype,
                              hipStream_t stream) {
  VG_REQUIRE(B > 0 && Tmax > 0 && H > 0, "vg_attn_decode: empty problem");
  dim3 grid(H, B);
  if (dtype == VG_BF16)
    VG_LAUNCH(attn_decode_kernel<bf16_t>, grid, dim3(64), 0, stream, (const bf16_t*)q,
                       (const bf16_t*)kcache, (const bf16_t*)vcache, (bf16_t*)out, slopes, pos, Tmax, H);
  else
    VG_LAUNCH(attn_decode_kernel<float>, grid, dim3(64), 0, stream, (const float*)q, (const float*)kcache,
                       (const float*)vcache, (float*)out, slopes, pos, Tmax, H);
  return vg_host::check_launch("vg_attn_decode");
}
