// Building blocks shared by the bf16 LDS-DMA GEMM kernels (vg_gemm_dma.hip, vg_gemm_ph.hip): operand tile images,
// their LDS-DMA fill, fragment reads and the common epilogue.  Included inside an anonymous namespace.
#pragma once
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"
#include "vg_gemm_params.h"

using namespace vg;

namespace {

constexpr int BK = 64;

// Results and epilogue operands are streamed once: non-temporal stores / loads keep them from evicting the operand panels
// the other blocks of the XCD are still reading out of L2 (round 4, measured in one call on cold operands: QKV forward
// 105-110 -> 97-101 us, N = K = 1024 34.4 -> 31.4 (NN 41 -> 34.5), N = 2048 / K = 512 45 -> 39, FFN-in 138-142 -> 134-137; in the
// training step 485.6 -> 497.7 k tokens/s).  -DVG_EPI_TEMPORAL restores the plain forms for A/B builds.  Per operand
// (one call, two runs each): the GELU launch's result, which the next GEMM reads as its A operand, with the default
// policy and its stored derivative streamed: 29.01-29.17 against 29.04-29.11 ms per step (no difference); the plain
// launches' results with the default policy: 29.48-29.52 ms (worse).  Everything streamed stays.
#ifndef VG_EPI_TEMPORAL
#define VG_EPI_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define VG_EPI_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define VG_EPI_STORE(ptr, val) (*(ptr) = (val))
#define VG_EPI_LOAD(ptr) (*(ptr))
#endif

// ---- issue the LDS-DMA loads of one operand tile
//  ROW image: [R rows][128 B], chunk position p of row r holds source chunk p ^ ((r >> 1) & 7)
//  TR  image: R/128 sub-images of [64 krows][256 B]; 64-B granule position g of krow k holds
//             source granule g ^ (k & 3), its 32-B halves swapped when bit 3 of k is set
template <bool TR, int R, int NW>
VG_DEVICE void dma_tile(__amdgpu_buffer_rsrc_t rsrc, char* tile, long ld_bytes, int rc0, int k0, int wave, int lane,
                        int klim) {
  constexpr int PER_WAVE = (R / 8) / NW;
  static_assert(PER_WAVE >= 1, "tile too small for the wave count");
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int piece = j * NW + wave;           // 1-KiB piece index inside the tile
    unsigned voff;
    if constexpr (!TR) {
      const int row = piece * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      voff = (unsigned)((long)(rc0 + row) * ld_bytes + (long)(k0 + chunk * 8) * 2);
      // K tail of a k-contiguous operand: chunks past the end of the row would read the next row, so they
      // are pointed past the end of the buffer instead (the range check writes zeros into LDS)
      if (k0 + chunk * 8 >= klim) voff = 0x7ffffff0u;
    } else {
      const int sub = piece >> 4;              // 128-column sub-image
      const int krow = (piece & 15) * 4 + (lane >> 4);
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
VG_DEVICE bf16x8 frag_of(const char* tile, int rc, int s, int lane) {
  // rc = first row/col of this wave's 16-wide MFMA tile inside the block tile; k-step s is 32 deep
  if constexpr (!TR) return RowTile<bf16_t, 64>::frag16(tile, rc, s, lane);
  else return TrTile<bf16_t, 128>::frag16(tile + (rc >> 7) * (64 * 256), 0, rc & 127, s, lane);
}

// ---- per-lane epilogue for 8 consecutive columns of one output row
VG_DEVICE void epilogue_emit(const GemmParams& p, bool split, int m, int n, float (&v)[8]) {
  const long idx = (long)m * p.ldc + n;
  if (split) {   // partial sums of a split-K wgrad: raw fp32 accumulation
    float* c = reinterpret_cast<float*>(p.C) + idx;
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(c + e, v[e]);
    return;
  }
  if (p.bias) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
  }
  if (p.pre_add) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.pre_add) + idx);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
  }
  const int act = p.act & 15;
  if (p.act & VG_ACT_SAVE_DERIV) {
    // the activation and its derivative share their transcendental; the derivative goes to aux_out
    bf16x8 o;
    u32x2_t w8 = {0u, 0u};        // VG_ACT_DERIV_U8 (GELU only, checked by the host side): one byte per derivative
    if (act == VG_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        f32x2_t cdf, px;
        gelu_parts_pk(f32x2_t{v[e], v[e + 1]}, cdf, px);
        o[e] = (bf16_t)(cdf[0] + px[0]);
        o[e + 1] = (bf16_t)(cdf[1] + px[1]);
        if (p.aux_u8) {
          if (e < 4) { w8.x = deriv_u8_put(cdf[0] + px[0], e, w8.x); w8.x = deriv_u8_put(cdf[1] + px[1], e + 1, w8.x); }
          else { w8.y = deriv_u8_put(cdf[0] + px[0], e - 4, w8.y); w8.y = deriv_u8_put(cdf[1] + px[1], e - 3, w8.y); }
        }
        v[e] *= cdf[0];
        v[e + 1] *= cdf[1];
      }
    } else if (act == VG_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = __frcp_rn(1.0f + __expf(-v[e]));
        o[e] = (bf16_t)(sg * (1.0f + v[e] * (1.0f - sg)));
        v[e] *= sg;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (bf16_t)((act != VG_ACT_RELU || v[e] > 0.f) ? 1.0f : 0.0f);
        if (act == VG_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
      }
    }
    if (p.aux_out && p.aux_u8) VG_EPI_STORE(reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned char*>(p.aux_out) + idx), w8);
    else if (p.aux_out) VG_EPI_STORE(reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.aux_out) + idx), o);
  } else {
    if (p.aux_out) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      VG_EPI_STORE(reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.aux_out) + idx), o);
    }
    if (act == VG_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (act == VG_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        f32x2_t cdf, px;
        gelu_parts_pk(f32x2_t{v[e], v[e + 1]}, cdf, px);
        v[e] *= cdf[0];
        v[e + 1] *= cdf[1];
      }
    } else if (act == VG_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = silu(v[e]);
    }
  }
  if (p.dact != VG_ACT_NONE && p.aux_u8) {    // x the 8-bit stored derivative
    const u32x2_t w = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned char*>(p.aux_in) + idx);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] *= deriv_u8_get(w.x, e);
      v[4 + e] *= deriv_u8_get(w.y, e);
    }
  } else if (p.dact != VG_ACT_NONE) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.aux_in) + idx);
    if (p.dact == VG_ACT_STORED) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= (float)a[e];
    } else if (p.dact == VG_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((float)a[e] > 0.f) ? v[e] : 0.f;
    } else if (p.dact == VG_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_fast((float)a[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= silu_grad((float)a[e]);
    }
  }
  if (p.residual) {
    const bf16x8 r = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.residual) + idx);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
  }
  if (!row_valid(p.lengths, p.T, p.m_base + m)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
  }
  if (p.out_f32) {
    f32x4* c = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + idx);
    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
    if (p.accumulate) { o0 += c[0]; o1 += c[1]; }
    c[0] = o0;
    c[1] = o1;
  } else {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    VG_EPI_STORE(reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.C) + idx), o);
  }
}

// ---- epilogue shared by the tile kernels (same contract as vg_gemm.hip)
// Accumulator map of the 16x16 tiles: col = lane & 15, row = 4 * (lane >> 4) + reg.  Each wave
// transposes one 16-row band at a time through a private LDS strip ([16 rows][TN*16 cols]) so that
// global accesses are row-contiguous: 16-byte vectors for the normal epilogue, two 128-byte row
// segments per wave-instruction for the split-K fp32 atomics (the shape the memory-side atomic
// units take at full rate).  `smem` must offer NW * 16 * (TN*16 + 4) floats that no DMA is writing.
// ILV: the wave's rows / columns are two interleaved blocks (gemm_ph_kernel): 16-row band i sits at tile row
// (i / (TM/2)) * BM/2 + wm * (BM/WM/2) + (i % (TM/2)) * 16, strip column c at tile column
// (c / (WCOLS/2)) * BN/2 + wn * (WCOLS/2) + c % (WCOLS/2); otherwise one contiguous (BM/WM) x (BN/WN) block.
template <int BM, int BN, int WM, int WN, bool ILV = false, bool ILVC = ILV>
VG_DEVICE void tile_epilogue(const GemmParams& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, int m0, int n0,
                             int wg, int nwg, int nsplit = (int)gridDim.z, int zidx = (int)blockIdx.z) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int WCOLS = TN * 16;
  constexpr int SW = WCOLS + 4;            // strip pitch in floats (16-byte aligned rows)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int arow = wm * (BM / WM), bcol = wn * (BN / WN);
  auto band_row = [&](int i) {      // tile-local first row of the wave's 16-row band i
    if constexpr (ILV) return (i / (TM / 2)) * (BM / 2) + wm * (BM / WM / 2) + (i % (TM / 2)) * 16;
    else return arow + i * 16;
  };
  auto strip_col = [&](int c) {     // tile-local column of strip column c (c a multiple of 8 or of 32)
    if constexpr (ILVC) return (c / (WCOLS / 2)) * (BN / 2) + wn * (WCOLS / 2) + c % (WCOLS / 2);
    else return bcol + c;
  };
  const bool split = nsplit > 1;
  __syncthreads();                         // every wave is done reading the last stage
  float* strip = reinterpret_cast<float*>(smem) + wave * (16 * SW);
  float cp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // column sums of this lane's 8 columns (colpart)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        strip[(4 * (lane >> 4) + rr) * SW + j * 16 + (lane & 15)] = acc[i][j][rr] * p.alpha;
    const int mband = m0 + band_row(i);
    if (split && p.split_ws) {
      // in-launch split-K reduction, step 1: this K-slice's partial tile goes to its fp32 slab with plain
      // 16-byte stores (tile-local [BM][BN] layout, no bounds: the slab is padded)
      // write-through (sc1) stores: the slab is in memory once vmcnt drains, so publishing it needs no
      // release fence (an agent-scope release would write back the XCD's whole L2 once per block)
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      const long slab_floats = (long)nsplit * nwg * (BM * BN);
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.split_ws, 0, (int)min(slab_floats * 4, 0x7fffffffL),
                                                                    0x00020000);
      const long slab = ((long)zidx * nwg + wg) * (BM * BN);
      constexpr int CPR = WCOLS / 8, RPP = 64 / CPR;
      const int crow = lane / CPR, cch = lane % CPR;
      for (int ps = 0; ps < (16 + RPP - 1) / RPP; ++ps) {
        const int rloc = ps * RPP + crow;
        if (rloc >= 16) continue;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(strip + rloc * SW + cch * 8);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(strip + rloc * SW + cch * 8 + 4);
        const unsigned off = (unsigned)((slab + (long)(band_row(i) + rloc) * BN + strip_col(cch * 8)) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + 16, 0, 16);
      }
    } else if (split) {
      float* __restrict__ c = reinterpret_cast<float*>(p.C);
      constexpr int SEGS = WCOLS / 32;     // 32-float segments per strip row
      for (int it = lane >> 5; it < 16 * SEGS; it += 2) {
        const int rloc = it / SEGS, seg = it % SEGS;
        const int m = mband + rloc, n = n0 + strip_col(seg * 32) + (lane & 31);
        if (m < p.M && n < p.N) atomicAdd(c + (long)m * p.ldc + n, strip[rloc * SW + seg * 32 + (lane & 31)]);
      }
    } else {
      constexpr int CPR = WCOLS / 8;       // 8-column chunks per strip row
      constexpr int RPP = 64 / CPR;        // rows covered per pass
      const int crow = lane / CPR, cch = lane % CPR;
      for (int ps = 0; ps < (16 + RPP - 1) / RPP; ++ps) {
        const int rloc = ps * RPP + crow;
        if (rloc >= 16) continue;
        const int m = mband + rloc, n = n0 + strip_col(cch * 8);
        const f32x4 lo = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8 + 4);
        if (m >= p.M || n >= p.N) continue;
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        epilogue_emit(p, false, m, n, v);
        if (p.colpart) {
#pragma unroll
          for (int e = 0; e < 8; ++e) cp[e] += v[e];
        }
      }
    }
  }
  // in-launch split-K reduction, steps 2 and 3 (wait-free; cdna_hip_programming.md section 4, item 2): every
  // slice publishes its slab (all waves drain their stores, barrier, ONE agent-scope release, ticket); the
  // slice that draws the last ticket acquires once and adds the sum of all slabs to C with plain 16-byte
  // accesses -- 1.3 TB/s of memory-side atomics become ~6 TB/s streams, and nobody ever waits.
  if (split && p.split_ws) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* bcast = reinterpret_cast<int*>(smem);
    if (tid == 0)      // slabs were stored write-through and drained: the ticket itself publishes them
      bcast[0] = __hip_atomic_fetch_add(p.split_cnt + wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = bcast[0];
    if (ticket != nsplit - 1) return;
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    constexpr int CPR = WCOLS / 8, RPP = 64 / CPR;
    const int crow = lane / CPR, cch = lane % CPR;
    float* __restrict__ c = reinterpret_cast<float*>(p.C);
    const long slab_stride = (long)nwg * (BM * BN);
    const float* __restrict__ slab0 = p.split_ws + (long)wg * (BM * BN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
      for (int ps = 0; ps < (16 + RPP - 1) / RPP; ++ps) {
        const int rloc = ps * RPP + crow;
        if (rloc >= 16) continue;
        const int m = m0 + band_row(i) + rloc, n = n0 + strip_col(cch * 8);
        if (m >= p.M || n >= p.N) continue;
        const long off = (long)(band_row(i) + rloc) * BN + strip_col(cch * 8);
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < nsplit; ++z) {
          const f32x4* src = reinterpret_cast<const f32x4*>(slab0 + z * slab_stride + off);
          lo += src[0];
          hi += src[1];
        }
        f32x4* dst = reinterpret_cast<f32x4*>(c + (long)m * p.ldc + n);
        dst[0] += lo;
        dst[1] += hi;
      }
    return;
  }
  // colpart: column sums of the tile's stored values -> colpart[m-tile][n].  Lanes that share a column
  // chunk are folded by shuffles, the WM waves of a column panel through LDS (the strips are free now).
  if (p.colpart && !split) {
    constexpr int CPR = WCOLS / 8;
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) cp[e] += __shfl_xor(cp[e], o, 64);
    __syncthreads();
    float* cred = reinterpret_cast<float*>(smem);          // [NW][WCOLS]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cred[wave * WCOLS + lane * 8 + e] = cp[e];
    }
    __syncthreads();
    if (wm == 0 && lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WM; ++w2) t += cred[(w2 * WN + wn) * WCOLS + lane * 8 + e];
        const int n = n0 + strip_col(lane * 8) + e;
        if (n < p.N) p.colpart[(long)(m0 / BM) * p.N + n] = t;
      }
    }
  }
}


// ---- lean epilogues for the three shapes almost every large product of the step ends in.  tile_epilogue above is
// one body for every option of the C ABI, unrolled over the wave's 8 row bands: ~15,000 instructions, of which a
// plain bf16 store walks a few thousand -- measured per 256x256 tile (in-kernel stamps, 4 tiles per CU, nothing else
// in the way): 7-8.6 us for a plain store (the same with the store itself removed: it is instruction issue, not
// memory), 19 us with GELU + stored derivative, 19.5 us when the result is multiplied by a stored derivative (a
// dependent 16-byte load inside every 8-column step).  Here the variant is a compile-time constant, the band's global
// operands (residual / stored derivative) are requested before its LDS transpose, and the row mask is worked out once
// per band (a band's 16 rows span at most two sequences).  Same arithmetic in the same order: results are bitwise
// those of tile_epilogue.
//   EPI_PLAIN      C = mask(relu?(acc + bias) + residual)                  (bias, ReLU, residual, lengths optional)
//   EPI_GELU_SAVE  C = mask(GELU(acc + bias)), aux_out = GELU'(acc + bias) (act = GELU | SAVE_DERIV)
//   EPI_SILU_SAVE  the same with SiLU, and an optional pre_add operand     (act = SILU | SAVE_DERIV: the conv blocks)
//   EPI_DACT       C = mask((acc + bias) * aux_in)                         (dact = STORED; dact = RELU: aux_in > 0 ? . : 0)
// Round 4, measured and rejected: a register-only form (MFMAs with swapped operands, so that a lane holds four consecutive
// columns of a row, one v_permlane16_swap per register to make them eight, 16-byte stores straight from registers: no
// LDS strip, no barrier, ~480 executed vector instructions per wave and tile instead of ~1,200, results bitwise equal).
// Its store instruction covers 16 rows x 64 bytes instead of 8 rows x 128 bytes, and the epilogue is no longer bound by
// instruction issue since the stores went non-temporal: in-kernel stamps 4.2-4.8 us per plain tile for THIS epilogue
// (256 CUs x 128 KB in that time is ~7.4 TB/s: the memory side's rate for a burst in which every CU stores at once)
// against 5.0-6.4 us for the register-only one; x stored derivative 8.0 -> 10.0 us; GELU + stored derivative
// 11.8 -> 10.3 us in the stamps build but no gain in the regular one (166 us per launch either way); training step
// 535-540 k -> 512-514 k tokens/s with all variants switched, 532-538 k with the GELU variant only.  Half-line stores
// cost more than the instructions saved.
// Host-side preconditions (gemm_ph_launch): bf16 C, alpha = 1, one K slice, no pre_add / accumulate / colsum_out,
// N % 8 == 0, T >= 16 when lengths are given.  colpart is supported (the dgrad that also reduces the bias gradient).
//   EPI_GELU_SAVE8 / EPI_DACT8 (round 6): the stored derivative is one byte per element (VG_ACT_DERIV_U8; aux = uint8
//                  [M][ldc]): 8 codes = one 8-byte access per lane and (row, pass) instead of 16 bytes
enum { EPI_GENERIC = 0, EPI_PLAIN = 1, EPI_GELU_SAVE = 2, EPI_DACT = 3, EPI_SILU_SAVE = 4, EPI_GELU_SAVE8 = 5, EPI_DACT8 = 6 };


template <int BM, int BN, int WM, int WN, bool ILV, bool ILVC, int EPI>
VG_DEVICE void tile_epilogue_lean(const GemmParams& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, int m0, int n0,
                                   const char* ring = nullptr, int aux_s0 = -1) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int WCOLS = TN * 16;
  constexpr int SW = WCOLS + 4;
  constexpr int CPR = WCOLS / 8, RPP = 64 / CPR;
  static_assert(RPP * 2 == 16, "two passes of 8 rows per band");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  auto band_row = [&](int i) {
    if constexpr (ILV) return (i / (TM / 2)) * (BM / 2) + wm * (BM / WM / 2) + (i % (TM / 2)) * 16;
    else return wm * (BM / WM) + i * 16;
  };
  auto strip_col = [&](int c) {
    if constexpr (ILVC) return (c / (WCOLS / 2)) * (BN / 2) + wn * (WCOLS / 2) + c % (WCOLS / 2);
    else return wn * (BN / WN) + c;
  };
  float* strip = reinterpret_cast<float*>(smem) + wave * (16 * SW);
  const int crow = lane / CPR, cch = lane % CPR;
  const int n = n0 + strip_col(cch * 8);
  const bool col_ok = n < p.N;
  const bf16_t* __restrict__ res = reinterpret_cast<const bf16_t*>(p.residual);
  const bf16_t* __restrict__ ain = reinterpret_cast<const bf16_t*>(p.aux_in);
  bf16_t* __restrict__ aout = reinterpret_cast<bf16_t*>(p.aux_out);
  bf16_t* __restrict__ cout = reinterpret_cast<bf16_t*>(p.C);
  float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias && col_ok) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias[e] = b0[e]; bias[4 + e] = b1[e]; }
  }
  const float relu_floor = (p.act & 15) == VG_ACT_RELU ? 0.f : -INFINITY;
  float cp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // the tile's global operands (residual / stored derivative: 16 x 16 bytes per lane) are all requested here, ahead
  // of the barrier and of the first LDS transpose: one exposed memory round trip per tile instead of one per band
  bf16x8 g[EPI == EPI_DACT8 ? 1 : TM][2];
  // 8-bit derivative (EPI_GELU_SAVE8 / EPI_DACT8): a lane's 8 codes of a (row, pass) are 8 bytes, and 8-byte accesses run at
  // 0.54 - 0.70 of the 16-byte rate (measured here: x derivative 135.4 us with the bf16 stream, 138.6 us with 8-byte loads
  // of half the bytes).  So the two lanes of a pair (cch even / odd: 16 consecutive columns) share the band's two rows:
  // the even lane moves pass 0's row, the odd lane pass 1's row (8 rows further), 16 codes = ONE 16-byte access per band
  // and lane, and the halves each lane needs from its partner cross with one DPP quad swap per dword.  N % 16 == 0
  // (lean_epilogue_of); the layout in memory stays plain uint8 [M][ldc], so the generic epilogue reads / writes the same bytes.
  const bool odd = (lane & 1) != 0;
  u32x4_t g8[EPI == EPI_DACT8 ? TM : 1];
  // (ring != nullptr: the tile's codes already sit in four ring slots of LDS -- gemm_ph_kernel fetched them through the
  // LDS-DMA ring as the K tile past the end, vg_gemm_ph.hip TileCtx::init_aux -- and each band reads its 8 + 8 bytes there)
  if constexpr (EPI == EPI_DACT8) {
    if (ring == nullptr) {
      const unsigned char* __restrict__ src8 = reinterpret_cast<const unsigned char*>(p.aux_in);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + band_row(i) + (odd ? RPP : 0) + crow;
        if (col_ok && m < p.M) g8[i] = VG_EPI_LOAD(reinterpret_cast<const u32x4_t*>(src8 + (long)m * p.ldc + (odd ? n - 8 : n)));
      }
    }
  }
  if constexpr (EPI == EPI_PLAIN || EPI == EPI_DACT || EPI == EPI_SILU_SAVE) {
    const bf16_t* __restrict__ src = EPI == EPI_PLAIN ? res : EPI == EPI_DACT ? ain : reinterpret_cast<const bf16_t*>(p.pre_add);
    if (src != nullptr) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int m = m0 + band_row(i) + ps * RPP + crow;
          if (col_ok && m < p.M) g[i][ps] = VG_EPI_LOAD(reinterpret_cast<const bf16x8*>(src + (long)m * p.ldc + n));
        }
    }
  }
  // packed rows (T == 1: every row is a one-frame pseudo sequence, lengths[m] > 0 is its predicate -- the layout of
  // hip.packed_rows / hip.packed_step): the row flags of the tile, requested up front like the other operands (round 6:
  // these launches used to fall back to the generic epilogue, ~3.5 us more per tile round)
  const bool rows1 = p.lengths != nullptr && p.T == 1;
  int live[TM][2];
  if (rows1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int m = m0 + band_row(i) + ps * RPP + crow;
        live[i][ps] = m < p.M ? p.lengths[p.m_base + m] : 0;
      }
  }
  __syncthreads();                         // every wave is done reading the last stage
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int mband = m0 + band_row(i);            // wave-uniform
    bool ok[2];
    long idx[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int m = mband + ps * RPP + crow;
      ok[ps] = col_ok && m < p.M;
      idx[ps] = (long)m * p.ldc + n;
    }
    // the band's rows sit in sequence b0 from frame t0 on and, past its end, in sequence b0 + 1 (m_base: this launch is the
    // lower row band of a product split over two launches, vg_gemm.hip: the row mask counts rows of the whole product)
    int t0 = 0, len0 = 0x7fffffff, len1 = 0x7fffffff;
    if (p.lengths != nullptr && !rows1) {
      const int mb = p.m_base + mband;
      const int b0 = mb / p.T;
      t0 = mb - b0 * p.T;
      len0 = (long)b0 * p.T < (long)p.m_base + p.M ? p.lengths[b0] : 0;
      len1 = (long)(b0 + 1) * p.T < (long)p.m_base + p.M ? p.lengths[b0 + 1] : 0;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        strip[(4 * (lane >> 4) + rr) * SW + j * 16 + (lane & 15)] = acc[i][j][rr];
    u32x2_t c8[2] = {{0u, 0u}, {0u, 0u}};      // this lane's 8 codes of pass 0 / pass 1 (read: from the pair's two loads; written: below)
    if constexpr (EPI == EPI_DACT8) {
      if (ring != nullptr) {
        // code bytes of (tile row r, tile byte-column cb): image (r >> 7) * 2 + (cb >> 7) in ring slot (aux_s0 + image) mod 10,
        // row image with the 16-byte chunk XOR of the row (RowTile / dma_tile: position = chunk ^ ((row >> 1) & 7))
        const int cb = strip_col(cch * 8);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int r = band_row(i) + ps * RPP + crow, rr = r & 127;
          int slot = aux_s0 + (r >> 7) * 2 + (cb >> 7);
          slot = slot >= 10 ? slot - 10 : slot;
          const char* at = ring + slot * 16384 + rr * 128 + (((((cb & 127) >> 4) ^ ((rr >> 1) & 7))) << 4) + (cb & 15);
          c8[ps] = *reinterpret_cast<const u32x2_t*>(at);
        }
      } else {                                   // (every lane takes part: the swap reads the partner's registers)
        const u32x4_t own = g8[i];
        const unsigned rx = dpp_quad_swap1(odd ? own[0] : own[2]), ry = dpp_quad_swap1(odd ? own[1] : own[3]);
        c8[0] = odd ? u32x2_t{rx, ry} : u32x2_t{own[0], own[1]};
        c8[1] = odd ? u32x2_t{own[2], own[3]} : u32x2_t{rx, ry};
      }
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int rloc = ps * RPP + crow;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8 + 4);
      if (!ok[ps]) continue;
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias[e];
      if constexpr (EPI == EPI_GELU_SAVE) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          f32x2_t cdf, px;
          gelu_parts_pk(f32x2_t{v[e], v[e + 1]}, cdf, px);
          o[e] = (bf16_t)(cdf[0] + px[0]);
          o[e + 1] = (bf16_t)(cdf[1] + px[1]);
          v[e] *= cdf[0];
          v[e + 1] *= cdf[1];
        }
        VG_EPI_STORE(reinterpret_cast<bf16x8*>(aout + idx[ps]), o);
      } else if constexpr (EPI == EPI_GELU_SAVE8) {
        u32x2_t w = {0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          f32x2_t cdf, px;
          gelu_parts_pk(f32x2_t{v[e], v[e + 1]}, cdf, px);
          if (e < 4) { w.x = deriv_u8_put(cdf[0] + px[0], e, w.x); w.x = deriv_u8_put(cdf[1] + px[1], e + 1, w.x); }
          else { w.y = deriv_u8_put(cdf[0] + px[0], e - 4, w.y); w.y = deriv_u8_put(cdf[1] + px[1], e - 3, w.y); }
          v[e] *= cdf[0];
          v[e + 1] *= cdf[1];
        }
        c8[ps] = w;                              // leaves after the pass loop, 16 codes per lane
      } else if constexpr (EPI == EPI_DACT8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] *= deriv_u8_get(c8[ps][0], e);
          v[4 + e] *= deriv_u8_get(c8[ps][1], e);
        }
      } else if constexpr (EPI == EPI_SILU_SAVE) {
        if (p.pre_add != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)g[i][ps][e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sg = __frcp_rn(1.0f + __expf(-v[e]));
          o[e] = (bf16_t)(sg * (1.0f + v[e] * (1.0f - sg)));
          v[e] *= sg;
        }
        VG_EPI_STORE(reinterpret_cast<bf16x8*>(aout + idx[ps]), o);
      } else if constexpr (EPI == EPI_DACT) {
        if (p.dact == VG_ACT_STORED) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= (float)g[i][ps][e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ((float)g[i][ps][e] > 0.f) ? v[e] : 0.f;
        }
      } else {
        // ReLU as a max against a wave-uniform floor (0 or -inf): one v_max per value whether the launch asks for it or
        // not -- the per-value select the compiler made of `if (act == RELU)` cost two more instructions per value in
        // every plain launch (436 of the epilogue's 1465 vector instructions).  NaN note (accepted): v_max
        // returns the non-NaN operand, so a NaN accumulator leaves a NON-ReLU launch as -inf here where tile_epilogue
        // passes the NaN through: still non-finite (a diverged run stays visible: tests/test_kernels_gpu.py,
        // test_nan_stays_non_finite_through_lean_epilogues), but the "bitwise equal to the generic epilogue" statement
        // holds for non-NaN accumulators only.  A NaN-propagating select costs a second instruction per value.
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], relu_floor);
        if (res != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)g[i][ps][e];
        }
      }
      if (p.lengths != nullptr) {          // wave-uniform: launches without a row mask skip the test and the selects
        const int t = t0 + rloc;
        const bool keep = rows1 ? live[i][ps] > 0 : (t < p.T ? t < len0 : t - p.T < len1);
        if (!keep) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      VG_EPI_STORE(reinterpret_cast<bf16x8*>(cout + idx[ps]), o);
      if (p.colpart) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cp[e] += v[e];
      }
    }
    if constexpr (EPI == EPI_GELU_SAVE8) {      // (all lanes again: the even lane stores pass 0's row of the pair, the odd lane pass 1's)
      const unsigned rx = dpp_quad_swap1(odd ? c8[0][0] : c8[1][0]), ry = dpp_quad_swap1(odd ? c8[0][1] : c8[1][1]);
      const u32x4_t w16 = odd ? u32x4_t{rx, ry, c8[1][0], c8[1][1]} : u32x4_t{c8[0][0], c8[0][1], rx, ry};
      if (odd ? ok[1] : ok[0])
        VG_EPI_STORE(reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(p.aux_out) + (odd ? idx[1] - 8 : idx[0])), w16);
    }
  }
  if (p.colpart) {
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) cp[e] += __shfl_xor(cp[e], o, 64);
    __syncthreads();
    float* cred = reinterpret_cast<float*>(smem);          // [NW][WCOLS]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cred[wave * WCOLS + lane * 8 + e] = cp[e];
    }
    __syncthreads();
    if (wm == 0 && lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WM; ++w2) t += cred[(w2 * WN + wn) * WCOLS + lane * 8 + e];
        const int nn = n0 + strip_col(lane * 8) + e;
        if (nn < p.N) p.colpart[(long)(m0 / BM) * p.N + nn] = t;
      }
    }
  }
}


// ---- lean epilogue of the weight-gradient products (fp32 C, alpha = 1, C += acc): plain 16-byte read-modify-write
// when this block holds the tile's whole K range (a plain store when p.accumulate == 0: C is known to hold zeros),
// fp32 atomics (two 128-byte row segments per wave-instruction) when it holds a piece of it.  Same arithmetic as tile_epilogue's two fp32 branches without its other 15,000 instructions.
template <int BM, int BN, int WM, int WN, bool ILV, bool ILVC>
VG_DEVICE void tile_epilogue_wgrad(const GemmParams& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, int m0, int n0,
                                   bool atomic) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int WCOLS = TN * 16;
  constexpr int SW = WCOLS + 4;
  constexpr int CPR = WCOLS / 8, RPP = 64 / CPR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  auto band_row = [&](int i) {
    if constexpr (ILV) return (i / (TM / 2)) * (BM / 2) + wm * (BM / WM / 2) + (i % (TM / 2)) * 16;
    else return wm * (BM / WM) + i * 16;
  };
  auto strip_col = [&](int c) {
    if constexpr (ILVC) return (c / (WCOLS / 2)) * (BN / 2) + wn * (WCOLS / 2) + c % (WCOLS / 2);
    else return wn * (BN / WN) + c;
  };
  __syncthreads();                         // every wave is done reading the last stage
  float* strip = reinterpret_cast<float*>(smem) + wave * (16 * SW);
  float* __restrict__ c = reinterpret_cast<float*>(p.C);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        strip[(4 * (lane >> 4) + rr) * SW + j * 16 + (lane & 15)] = acc[i][j][rr];
    const int mband = m0 + band_row(i);
    if (atomic) {
      constexpr int SEGS = WCOLS / 32;     // 32-float segments per strip row
      for (int it = lane >> 5; it < 16 * SEGS; it += 2) {
        const int rloc = it / SEGS, seg = it % SEGS;
        const int m = mband + rloc, n = n0 + strip_col(seg * 32) + (lane & 31);
        if (m < p.M && n < p.N) atomicAdd(c + (long)m * p.ldc + n, strip[rloc * SW + seg * 32 + (lane & 31)]);
      }
    } else {
      const int crow = lane / CPR, cch = lane % CPR;
#pragma unroll
      for (int ps = 0; ps < 16 / RPP; ++ps) {
        const int rloc = ps * RPP + crow;
        const int m = mband + rloc, n = n0 + strip_col(cch * 8);
        const f32x4 lo = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(strip + rloc * SW + cch * 8 + 4);
        if (m >= p.M || n >= p.N) continue;
        f32x4* dst = reinterpret_cast<f32x4*>(c + (long)m * p.ldc + n);
        if (p.accumulate) {
          dst[0] += lo;
          dst[1] += hi;
        } else {      // the caller vouches that C holds zeros (first contribution since the optimizer cleared it): no read
          __builtin_nontemporal_store(lo, dst);
          __builtin_nontemporal_store(hi, dst + 1);
        }
      }
    }
  }
}


// which lean epilogue (vg_gemm_tile.h) covers this problem; EPI_GENERIC when none does
inline int lean_epilogue_of(const GemmParams& p, int splits) {
  static const int off = [] { const char* e = getenv("VG_NO_LEAN_EPI"); return e ? atoi(e) : 0; }();
  if (off) return EPI_GENERIC;
  if (p.out_f32 || p.accumulate || splits != 1 || p.alpha != 1.0f || p.split_ws || p.colsum_out) return EPI_GENERIC;
  if (p.N % 8 != 0 || p.ldc % 8 != 0 || (p.lengths != nullptr && p.T < 16 && p.T != 1)) return EPI_GENERIC;
  // the lean epilogues move C / residual / aux / pre_add as bf16x8 and the bias as f32x4: an output view whose first
  // column is not a multiple of 8 (or a bias slice off a 4-float boundary) takes the element-wise generic epilogue
  const uintptr_t ptrs16 = (uintptr_t)p.C | (uintptr_t)p.residual | (p.aux_u8 ? 0 : ((uintptr_t)p.aux_in | (uintptr_t)p.aux_out)) |
                           (uintptr_t)p.pre_add | (uintptr_t)p.bias;
  if (ptrs16 & 15) return EPI_GENERIC;
  const int act = p.act & 15;
  const bool save = (p.act & VG_ACT_SAVE_DERIV) != 0;
  if (p.pre_add && !(act == VG_ACT_SILU && save)) return EPI_GENERIC;
  if ((act == VG_ACT_NONE || act == VG_ACT_RELU) && !save && !p.aux_out && p.dact == VG_ACT_NONE) return EPI_PLAIN;
  // (the 8-bit forms move 16 codes per lane pair: whole 16-column groups, 16-byte aligned rows)
  const bool u8_ok = p.aux_u8 && p.N % 16 == 0 && p.ldc % 16 == 0 && (((uintptr_t)p.aux_in | (uintptr_t)p.aux_out) & 15) == 0;
  if (act == VG_ACT_GELU && save && p.aux_out && !p.residual && p.dact == VG_ACT_NONE && (!p.aux_u8 || u8_ok))
    return p.aux_u8 ? EPI_GELU_SAVE8 : EPI_GELU_SAVE;
  if (u8_ok && p.dact == VG_ACT_STORED && act == VG_ACT_NONE && !save && !p.aux_out && p.aux_in && !p.residual) return EPI_DACT8;
  if (p.aux_u8) return EPI_GENERIC;
  if (act == VG_ACT_SILU && save && p.aux_out && !p.residual && p.dact == VG_ACT_NONE) return EPI_SILU_SAVE;
  if (act == VG_ACT_NONE && !save && !p.aux_out && (p.dact == VG_ACT_STORED || p.dact == VG_ACT_RELU) && p.aux_in && !p.residual)
    return EPI_DACT;
  return EPI_GENERIC;
}


}  // namespace
