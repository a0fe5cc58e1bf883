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
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"
#include "vg_gemm_params.h"

using namespace vg;

namespace {

constexpr int BK = 64;

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
    if (act == VG_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        f32x2_t cdf, px;
        gelu_parts_pk(f32x2_t{v[e], v[e + 1]}, cdf, px);
        o[e] = (bf16_t)(cdf[0] + px[0]);
        o[e + 1] = (bf16_t)(cdf[1] + px[1]);
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
    if (p.aux_out) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.aux_out) + idx) = o;
  } else {
    if (p.aux_out) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.aux_out) + idx) = o;
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
  if (p.dact != VG_ACT_NONE) {
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
  if (!row_valid(p.lengths, p.T, m)) {
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
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.C) + idx) = o;
  }
}

// ---- epilogue shared by the tile kernels (same contract as vg_gemm.hip)
// Accumulator map of the 16x16 tiles: col = lane & 15, row = 4 * (lane >> 4) + reg.  Each wave
// transposes one 16-row band at a time through a private LDS strip ([16 rows][TN*16 cols]) so that
// global accesses are row-contiguous: 16-byte vectors for the normal epilogue, two 128-byte row
// segments per wave-instruction for the split-K fp32 atomics (the shape the memory-side atomic
// units take at full rate).  `smem` must offer NW * 16 * (TN*16 + 4) floats that no DMA is writing.
template <int BM, int BN, int WM, int WN>
VG_DEVICE void tile_epilogue(const GemmParams& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, int m0, int n0,
                             int wg, int nwg) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int WCOLS = TN * 16;
  constexpr int SW = WCOLS + 4;            // strip pitch in floats (16-byte aligned rows)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int arow = wm * (BM / WM), bcol = wn * (BN / WN);
  const bool split = gridDim.z > 1;
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
    const int mband = m0 + arow + i * 16;
    if (split && p.split_ws) {
      // in-launch split-K reduction, step 1: this K-slice's partial tile goes to its fp32 slab with plain
      // 16-byte stores (tile-local [BM][BN] layout, no bounds: the slab is padded)
      // write-through (sc1) stores: the slab is in memory once vmcnt drains, so publishing it needs no
      // release fence (an agent-scope release would write back the XCD's whole L2 once per block)
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      const long slab_floats = (long)gridDim.z * nwg * (BM * BN);
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.split_ws, 0, (int)min(slab_floats * 4, 0x7fffffffL),
                                                                    0x00020000);
      const long slab = ((long)blockIdx.z * nwg + wg) * (BM * BN);
      constexpr int CPR = WCOLS / 8, RPP = 64 / CPR;
      const int crow = lane / CPR, cch = lane % CPR;
      for (int ps = 0; ps < (16 + RPP - 1) / RPP; ++ps) {
        const int rloc = ps * RPP + crow;
        if (rloc >= 16) continue;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(strip + rloc * SW + cch * 8);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(strip + rloc * SW + cch * 8 + 4);
        const unsigned off = (unsigned)((slab + (long)(arow + i * 16 + rloc) * BN + bcol + cch * 8) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(lo, rs, off, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(hi, rs, off + 16, 0, 16);
      }
    } else if (split) {
      float* __restrict__ c = reinterpret_cast<float*>(p.C);
      constexpr int SEGS = WCOLS / 32;     // 32-float segments per strip row
      for (int it = lane >> 5; it < 16 * SEGS; it += 2) {
        const int rloc = it / SEGS, seg = it % SEGS;
        const int m = mband + rloc, n = n0 + bcol + seg * 32 + (lane & 31);
        if (m < p.M && n < p.N) atomicAdd(c + (long)m * p.ldc + n, strip[rloc * SW + seg * 32 + (lane & 31)]);
      }
    } else {
      constexpr int CPR = WCOLS / 8;       // 8-column chunks per strip row
      constexpr int RPP = 64 / CPR;        // rows covered per pass
      const int crow = lane / CPR, cch = lane % CPR;
      for (int ps = 0; ps < (16 + RPP - 1) / RPP; ++ps) {
        const int rloc = ps * RPP + crow;
        if (rloc >= 16) continue;
        const int m = mband + rloc, n = n0 + bcol + cch * 8;
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
    if (ticket != (int)gridDim.z - 1) return;
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
        const int m = m0 + arow + i * 16 + rloc, n = n0 + bcol + cch * 8;
        if (m >= p.M || n >= p.N) continue;
        const long off = (long)(arow + i * 16 + rloc) * BN + bcol + cch * 8;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < (int)gridDim.z; ++z) {
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
        const int n = n0 + bcol + lane * 8 + e;
        if (n < p.N) p.colpart[(long)(m0 / BM) * p.N + n] = t;
      }
    }
  }
}

template <bool A_TR, bool B_TR, int BM, int BN, int WM, int WN, int STAGES, bool COLSUM = false>
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
  tile_epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, wg, nwg);
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

template <bool A_TR, bool B_TR, int BM, int BN, int WM, int WN, int STAGES = 2, bool COLSUM = false>
int launch_cfg(const GemmParams& p, int splits, hipStream_t stream) {
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * BK * 2;
  auto k = gemm_dma_kernel<A_TR, B_TR, BM, BN, WM, WN, STAGES, COLSUM>;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  hipLaunchKernelGGL(k, dim3(ntn * ntm, 1, splits), dim3(WM * WN * 64), lds, stream, p);
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
  hipLaunchKernelGGL(k, dim3(ntn * ntm, 1, splits), dim3(WM * WN * 64), lds, stream, p);
  return 0;
}

template <bool A_TR, bool B_TR>
int launch_mode(const GemmParams& p, int cfg, int splits, hipStream_t stream) {
  if constexpr (A_TR && B_TR) {   // fused bias gradient: 128x128 tiles only (register budget)
    if (p.colsum_out) return launch_cfg<A_TR, B_TR, 128, 128, 2, 2, 2, true>(p, splits, stream);
  }
  switch (cfg) {
    case 1: return launch_cfg<A_TR, B_TR, 128, 128, 2, 2>(p, splits, stream);
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
