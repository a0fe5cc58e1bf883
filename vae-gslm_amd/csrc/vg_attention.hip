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
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

using namespace vg;

namespace {

constexpr int DH = 64;
constexpr int QB = 128;      // time rows owned by a block (4 waves x 32)
constexpr int TB = 64;       // time rows staged per LDS tile
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float LN2 = 0.69314718055994530942f;
constexpr float SCALE = 0.125f;   // 1 / sqrt(64)
// resident blocks per CU the bf16 kernels are register-budgeted for (measured: forward 3, backward 2)
#ifndef VG_ATTN_OCC_FWD
#define VG_ATTN_OCC_FWD 3
#endif
#ifndef VG_ATTN_OCC
#define VG_ATTN_OCC 2
#endif

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
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, (int)bytes, 0x00020000);
}
template <bool TR>
VG_DEVICE void slab_dma(__amdgpu_buffer_rsrc_t rsrc, char* img, long row_stride, int t0, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int piece = wave + 4 * j;
    const int row = piece * 8 + (lane >> 3);
    const int pos = lane & 7;               // 16-byte position inside the 128-byte LDS row
    int col;
    if constexpr (!TR) col = (pos ^ ((row >> 1) & 7)) * 8;
    else col = (((pos >> 2) ^ ((row >> 1) & 1)) * 32) + (pos & 3) * 8;
    const unsigned voff = (unsigned)(((long)(t0 + row) * row_stride + col) * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(void, img + piece * 1024), 16, voff, 0, 0, 0);
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

template <typename T>
VG_DEVICE f32x16 mma_row_regs(const char* row_img, int row, const RowRegs<T>& b, int lane, f32x16 acc) {
  constexpr int STEPS = DH / Traits<T>::KSTEP;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) acc = Traits<T>::mfma(RowTile<T, DH>::frag(row_img, row, s, lane), b.f[s], acc);
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

VG_DEVICE float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
VG_DEVICE float xhalf_sum(float v) { return v + __shfl_xor(v, 32, 64); }

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
  float a = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
  float b = fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]));
  float c = fmaxf(fmaxf(s[8], s[9]), fmaxf(s[10], s[11]));
  float d = fmaxf(fmaxf(s[12], s[13]), fmaxf(s[14], s[15]));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
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
                                                       int Tn, int H, const int* __restrict__ lengths) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_row = smem;
  char* v_tr = smem + LdsPlan<T>::ROW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, longest sweeps first over the whole launch (LPT order); the tiles of one (b, h)
  // are H*B apart, i.e. on the same XCD whenever H*B is a multiple of 8, and share K/V in its L2
  const int nqt = (Tn + QB - 1) / QB, HB = gridDim.x / nqt, hb = blockIdx.x % HB;
  const int qt = nqt - 1 - (int)(blockIdx.x / HB), h = hb % H, b = hb / H;
  const int D = H * DH;
  const long rs = 3L * D;
  const int len = lengths ? min(lengths[b], Tn) : Tn;
  const int q0 = qt * QB;
  const int qw0 = q0 + wave * 32;
  const int query = qw0 + (lane & 31);
  const T* __restrict__ base = qkv + (long)b * Tn * rs + h * DH;
  T* __restrict__ obase = out + (long)b * Tn * D + h * DH;

  if (q0 >= len) {   // fully padded tile: zero rows (attention.py:80 re-mask)
    f32x16 z[2] = {zero16(), zero16()};
    if (query < Tn) store_rows_T<T>(obase + (long)query * D, z, 0.f, lane);
    return;
  }
  const int qend = min(q0 + QB, len);
  const int nkt = (qend + TB - 1) / TB;

  RowRegs<T> qf;
  qf.load(base + (long)min(query, Tn - 1) * rs, lane);
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
    rsk = slab_rsrc(reinterpret_cast<const bf16_t*>(base + D), rs, Tn);
    rsv = slab_rsrc(reinterpret_cast<const bf16_t*>(base + 2 * D), rs, Tn);
    slab_dma<false>(rsk, smem, rs, 0, wave, lane);
    slab_dma<true>(rsv, smem + LdsPlan<T>::ROW_BYTES, rs, 0, wave, lane);
  } else {
    slab_load<T>(rk, base + D, rs, 0, Tn, tid);
    slab_load<T>(rv, base + 2 * D, rs, 0, Tn, tid);
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
        slab_load<T>(rk, base + D, rs, kv0 + TB, Tn, tid);
        slab_load<T>(rv, base + 2 * D, rs, kv0 + TB, Tn, tid);
      }
    }
    if (qw0 + 31 < kv0) continue;   // this wave's queries all precede the tile (causal)
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
  if (query < Tn) {
    const bool valid = query < len;
    store_rows_T<T>(obase + (long)query * D, o, valid ? 1.f / l : 0.f, lane);
    if (valid && lane < 32)
      lse[((long)b * H + h) * Tn + query] = (m + log2f(l) - slope2 * (float)(query - qw0)) * LN2;
  }
}

// =====================================================================================
// backward: delta = rowsum(dO * O) per (b, h, t)
// =====================================================================================
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta,
                                  int B, int Tn, int H) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gid & 7;
  const long mh = gid >> 3;
  const int h = mh % H;
  const long m = mh / H;
  float acc = 0.f;
  if (m < (long)B * Tn) {
    const long off = m * (long)H * DH + h * DH + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += to_f32<T>(o[off + e]) * to_f32<T>(dout[off + e]);
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  if (c == 0 && m < (long)B * Tn) {
    const int b = m / Tn, t = m % Tn;
    delta[((long)b * H + h) * Tn + t] = acc;
  }
}

// =====================================================================================
// backward: dQ  (block owns 128 queries, sweeps key tiles; no atomics)
// p = exp2(raw*c2 + slope2*(key - qw0) - [lse2 + slope2*(query - qw0)]);  dS = p * (dP - delta)
// (the 1/sqrt(d) factor of dS is applied once to the final dQ)
// =====================================================================================
template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? VG_ATTN_OCC : 1) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta,
                                                          const float* __restrict__ slopes, T* __restrict__ dqkv,
                                                          int Tn, int H, const int* __restrict__ lengths) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_row = smem;
  char* v_row = smem + LdsPlan<T>::ROW_BYTES;
  char* k_tr = smem + 2 * LdsPlan<T>::ROW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, longest sweeps first over the whole launch (LPT order); the tiles of one (b, h)
  // are H*B apart, i.e. on the same XCD whenever H*B is a multiple of 8, and share K/V in its L2
  const int nqt = (Tn + QB - 1) / QB, HB = gridDim.x / nqt, hb = blockIdx.x % HB;
  const int qt = nqt - 1 - (int)(blockIdx.x / HB), h = hb % H, b = hb / H;
  const int D = H * DH;
  const long rs = 3L * D;
  const int len = lengths ? min(lengths[b], Tn) : Tn;
  const int q0 = qt * QB, qw0 = q0 + wave * 32, query = qw0 + (lane & 31);
  const T* __restrict__ base = qkv + (long)b * Tn * rs + h * DH;
  T* __restrict__ dqbase = dqkv + (long)b * Tn * rs + h * DH;
  if (q0 >= len) {
    f32x16 z[2] = {zero16(), zero16()};
    if (query < Tn) store_rows_T<T>(dqbase + (long)query * rs, z, 0.f, lane);
    return;
  }
  const int qend = min(q0 + QB, len);
  const int nkt = (qend + TB - 1) / TB;
  const bool qvalid = query < len;
  const int qc = min(query, Tn - 1);
  RowRegs<T> qf, dof;
  qf.load(base + (long)qc * rs, lane);
  dof.load(dout + ((long)b * Tn + qc) * D + h * DH, lane);
  const float slope = slopes[h];
  const float slope2 = slope * LOG2E, c2 = SCALE * LOG2E;
  // +inf for padded queries -> p = exp2(-inf) = 0 without a compare
  const float Lq = qvalid ? lse[((long)b * H + h) * Tn + query] * LOG2E + slope2 * (float)(query - qw0) : INFINITY;
  const float dl = qvalid ? delta[((long)b * H + h) * Tn + query] : 0.f;
  f32x16 kinit, dinit;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    kinit[i] = slope * 8.0f * (float)acc_row(i, lane);
    dinit[i] = -dl;
  }

  f32x16 dq[2] = {zero16(), zero16()};
  constexpr bool DMA = sizeof(T) == 2;
  constexpr int STAGE = 2 * LdsPlan<T>::ROW_BYTES + LdsPlan<T>::TR_BYTES;
  uint4 rk[DMA ? 1 : NVec<T>::v], rv[DMA ? 1 : NVec<T>::v];
  __amdgpu_buffer_rsrc_t rsk, rsv;
  auto issue = [&](int t0, char* st) {
    slab_dma<false>(rsk, st, rs, t0, wave, lane);
    slab_dma<false>(rsv, st + LdsPlan<T>::ROW_BYTES, rs, t0, wave, lane);
    slab_dma<true>(rsk, st + 2 * LdsPlan<T>::ROW_BYTES, rs, t0, wave, lane);
  };
  if constexpr (DMA) {
    rsk = slab_rsrc(reinterpret_cast<const bf16_t*>(base + D), rs, Tn);
    rsv = slab_rsrc(reinterpret_cast<const bf16_t*>(base + 2 * D), rs, Tn);
    issue(0, smem);
  } else {
    slab_load<T>(rk, base + D, rs, 0, Tn, tid);
    slab_load<T>(rv, base + 2 * D, rs, 0, Tn, tid);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    const int kv0 = kt * TB;
    if constexpr (DMA) {
      dma_wait_and_publish();
      if (kt + 1 < nkt) issue(kv0 + TB, smem + ((kt + 1) & 1) * STAGE);
      k_row = smem + (kt & 1) * STAGE;
      v_row = k_row + LdsPlan<T>::ROW_BYTES;
      k_tr = k_row + 2 * LdsPlan<T>::ROW_BYTES;
    } else {
      __syncthreads();
      slab_store<T, true, true>(rk, k_row, k_tr, tid);
      slab_store<T, true, false>(rv, v_row, nullptr, tid);
      __syncthreads();
      if (kt + 1 < nkt) {
        slab_load<T>(rk, base + D, rs, kv0 + TB, Tn, tid);
        slab_load<T>(rv, base + 2 * D, rs, kv0 + TB, Tn, tid);
      }
    }
    if (qw0 + 31 < kv0) continue;
    const bool diag = kv0 + TB - 1 > qw0;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s = mma_row_regs<T>(k_row, kb * 32 + (lane & 31), qf, lane, kinit);
      f32x16 dp = mma_row_regs<T>(v_row, kb * 32 + (lane & 31), dof, lane, dinit);
      if (diag) {
        const int lim = query - kv0 - kb * 32;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = acc_row(i, lane) <= lim ? s[i] : -INFINITY;
      }
      const float off = slope2 * (float)(kv0 + kb * 32 - qw0) - Lq;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = fexp2<T>(fmaf(s[i], c2, off)) * dp[i];
#pragma unroll
      for (int db = 0; db < 2; ++db) dq[db] = mma_tr_acc<T>(k_tr, kb * 32, db, s, lane, dq[db]);
    }
  }
  if (query < Tn) store_rows_T<T>(dqbase + (long)query * rs, dq, SCALE, lane);
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
                                                           int Tn, int H, const int* __restrict__ lengths) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* q_row = smem;
  char* do_row = smem + LdsPlan<T>::ROW_BYTES;
  char* q_tr = smem + 2 * LdsPlan<T>::ROW_BYTES;
  char* do_tr = q_tr + LdsPlan<T>::TR_BYTES;
  float* st = reinterpret_cast<float*>(do_tr + LdsPlan<T>::TR_BYTES);   // [2][64]: S init, dP init
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HB = gridDim.x / ((Tn + QB - 1) / QB), hb = blockIdx.x % HB;   // low key tiles sweep the most query tiles: first
  const int ktile = blockIdx.x / HB, h = hb % H, b = hb / H;
  const int D = H * DH;
  const long rs = 3L * D;
  const int len = lengths ? min(lengths[b], Tn) : Tn;
  const int k0 = ktile * QB, kw0 = k0 + wave * 32, key = kw0 + (lane & 31);
  const T* __restrict__ base = qkv + (long)b * Tn * rs + h * DH;
  const T* __restrict__ dobase = dout + (long)b * Tn * D + h * DH;
  T* __restrict__ dkbase = dqkv + (long)b * Tn * rs + D + h * DH;
  T* __restrict__ dvbase = dqkv + (long)b * Tn * rs + 2 * D + h * DH;
  f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
  if (k0 >= len) {
    if (key < Tn) {
      store_rows_T<T>(dkbase + (long)key * rs, dk, 0.f, lane);
      store_rows_T<T>(dvbase + (long)key * rs, dv, 0.f, lane);
    }
    return;
  }
  const int kc = min(key, Tn - 1);
  RowRegs<T> kf, vf;
  kf.load(base + D + (long)kc * rs, lane);
  vf.load(base + 2 * D + (long)kc * rs, lane);
  const float slope = slopes[h];
  const float slope2 = slope * LOG2E, c2 = SCALE * LOG2E;
  const float kl = slope2 * (float)(key - k0);
  const float* __restrict__ lse_bh = lse + ((long)b * H + h) * Tn;
  const float* __restrict__ dl_bh = delta + ((long)b * H + h) * Tn;

  const int qt_beg = k0 / TB, qt_end = (len + TB - 1) / TB;
  constexpr bool DMA = sizeof(T) == 2;
  constexpr int IMG = 2 * LdsPlan<T>::ROW_BYTES + 2 * LdsPlan<T>::TR_BYTES;
  constexpr int STAGE = IMG + 2 * 64 * (int)sizeof(float);     // + the per-query constants of the tile
  uint4 rq[DMA ? 1 : NVec<T>::v], rd[DMA ? 1 : NVec<T>::v];
  __amdgpu_buffer_rsrc_t rsq, rsd;
  auto issue = [&](int t0, char* sg) {
    slab_dma<false>(rsq, sg, rs, t0, wave, lane);
    slab_dma<false>(rsd, sg + LdsPlan<T>::ROW_BYTES, D, t0, wave, lane);
    slab_dma<true>(rsq, sg + 2 * LdsPlan<T>::ROW_BYTES, rs, t0, wave, lane);
    slab_dma<true>(rsd, sg + 2 * LdsPlan<T>::ROW_BYTES + LdsPlan<T>::TR_BYTES, D, t0, wave, lane);
  };
  // per-query constants of a tile: S init = -(lse2 + slope2 (q - k0)) / c2 (-inf for padded queries), dP init = -delta
  auto st_values = [&](int qs, float& a, float& d) {
    const int qq = qs + tid;
    const bool ok = tid < 64 && qq < len;
    a = ok ? -(lse_bh[qq] * LOG2E + slope2 * (float)(qq - k0)) / c2 : -INFINITY;
    d = ok ? -dl_bh[qq] : 0.f;
  };
  float st_a = 0.f, st_d = 0.f;    // wave 0: constants of the NEXT tile, fetched one iteration ahead
  if constexpr (DMA) {
    rsq = slab_rsrc(reinterpret_cast<const bf16_t*>(base), rs, Tn);
    rsd = slab_rsrc(reinterpret_cast<const bf16_t*>(dobase), D, Tn);
    if (tid < 64) {
      st_values(qt_beg * TB, st_a, st_d);
      float* s0 = reinterpret_cast<float*>(smem + IMG);
      s0[tid] = st_a;
      s0[64 + tid] = st_d;
      if (qt_beg + 1 < qt_end) st_values((qt_beg + 1) * TB, st_a, st_d);
    }
    issue(qt_beg * TB, smem);
  } else {
    slab_load<T>(rq, base, rs, qt_beg * TB, Tn, tid);
    slab_load<T>(rd, dobase, D, qt_beg * TB, Tn, tid);
  }
  for (int qt = qt_beg; qt < qt_end; ++qt) {
    const int qs0 = qt * TB;
    if constexpr (DMA) {
      dma_wait_and_publish();
      const int sl = (qt - qt_beg) & 1;
      if (qt + 1 < qt_end) {
        char* nx = smem + (sl ^ 1) * STAGE;
        if (tid < 64) {              // registers were filled one iteration ago: no wait behind the DMA below
          float* sn = reinterpret_cast<float*>(nx + IMG);
          sn[tid] = st_a;
          sn[64 + tid] = st_d;
        }
        issue(qs0 + TB, nx);
        if (tid < 64 && qt + 2 < qt_end) st_values(qs0 + 2 * TB, st_a, st_d);
      }
      q_row = smem + sl * STAGE;
      do_row = q_row + LdsPlan<T>::ROW_BYTES;
      q_tr = q_row + 2 * LdsPlan<T>::ROW_BYTES;
      do_tr = q_tr + LdsPlan<T>::TR_BYTES;
      st = reinterpret_cast<float*>(q_row + IMG);
    } else {
      __syncthreads();
      slab_store<T, true, true>(rq, q_row, q_tr, tid);
      slab_store<T, true, true>(rd, do_row, do_tr, tid);
      if (tid < 64) st_values(qs0, st[tid], st[64 + tid]);
      __syncthreads();
      if (qt + 1 < qt_end) {
        slab_load<T>(rq, base, rs, qs0 + TB, Tn, tid);
        slab_load<T>(rd, dobase, D, qs0 + TB, Tn, tid);
      }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qb0 = qs0 + qb * 32;
      if (qb0 + 31 < kw0) continue;   // all queries precede this wave's keys
      f32x16 s = mma_row_regs<T>(q_row, qb * 32 + (lane & 31), kf, lane, rows16(st, qb * 32, lane));
      f32x16 dp = mma_row_regs<T>(do_row, qb * 32 + (lane & 31), vf, lane, rows16(st + 64, qb * 32, lane));
      if (qb0 < kw0 + 31) {            // diagonal block: mask query < key
        const int lim = key - qb0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = acc_row(i, lane) >= lim ? s[i] : -INFINITY;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[i] = fexp2<T>(fmaf(s[i], c2, kl));
        dp[i] *= s[i];
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        dv[db] = mma_tr_acc<T>(do_tr, qb * 32, db, s, lane, dv[db]);
        dk[db] = mma_tr_acc<T>(q_tr, qb * 32, db, dp, lane, dk[db]);
      }
    }
  }
  if (key < Tn) {
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

template <typename T>
int launch_fwd(const void* qkv, void* out, float* lse, const float* slopes, int B, int Tn, int H,
               const int32_t* lengths, hipStream_t stream) {
  const size_t lds = (sizeof(T) == 2 ? 2 : 1) * (LdsPlan<T>::ROW_BYTES + LdsPlan<T>::TR_BYTES);
  dim3 grid(((Tn + QB - 1) / QB) * H * B);
  // algorithmic work: causal-exact QK^T + PV, 2*2*64 FLOP per (query, key <= query) pair
  const int tok = vg_host::prof_begin(VG_PROF_ATTN_FWD, 256.0 * B * H * (0.5 * Tn * (Tn + 1.0)), stream);
  hipLaunchKernelGGL(attn_fwd_kernel<T>, grid, dim3(256), lds, stream, (const T*)qkv, (T*)out, lse, slopes, Tn, H,
                     lengths);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_attn_fwd");
}

template <typename T>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const float* slopes,
               void* dqkv, float* delta, int B, int Tn, int H, const int32_t* lengths, hipStream_t stream) {
  const long nthreads = (long)B * Tn * H * 8;
  // algorithmic work of the backward: 5 products (S, dP, dV, dK, dQ) = 2.5 x forward
  const int tok = vg_host::prof_begin(VG_PROF_ATTN_BWD, 640.0 * B * H * (0.5 * Tn * (Tn + 1.0)), stream);
  hipLaunchKernelGGL(attn_delta_kernel<T>, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream,
                     (const T*)out, (const T*)dout, delta, B, Tn, H);
  dim3 grid(((Tn + QB - 1) / QB) * H * B);
  const size_t lds_q = (sizeof(T) == 2 ? 2 : 1) * (2 * LdsPlan<T>::ROW_BYTES + LdsPlan<T>::TR_BYTES);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, grid, dim3(256), lds_q, stream, (const T*)qkv, (const T*)dout, lse,
                     delta, slopes, (T*)dqkv, Tn, H, lengths);
  const size_t lds_k = (sizeof(T) == 2 ? 2 : 1) * (2 * LdsPlan<T>::ROW_BYTES + 2 * LdsPlan<T>::TR_BYTES + 2 * 64 * sizeof(float));
  static bool attr[2] = {false, false};
  if (!attr[sizeof(T) == 2]) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<T>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
    attr[sizeof(T) == 2] = true;
  }
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<T>, grid, dim3(256), lds_k, stream, (const T*)qkv, (const T*)dout, lse,
                     delta, slopes, (T*)dqkv, Tn, H, lengths);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_attn_bwd");
}

}  // namespace

extern "C" int vg_attn_fwd(const void* qkv, void* out, float* lse, const float* slopes, int B, int T, int H,
                           const int32_t* lengths, int dtype, hipStream_t stream) {
  VG_REQUIRE(B > 0 && T > 0 && H > 0, "vg_attn_fwd: empty problem");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_attn_fwd: bad dtype %d", dtype);
  VG_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0, "vg_attn_fwd: unaligned");
  if (dtype == VG_BF16) return launch_fwd<bf16_t>(qkv, out, lse, slopes, B, T, H, lengths, stream);
  return launch_fwd<float>(qkv, out, lse, slopes, B, T, H, lengths, stream);
}

extern "C" int vg_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                           const float* slopes, void* dqkv, float* delta, int B, int T, int H,
                           const int32_t* lengths, int dtype, hipStream_t stream) {
  VG_REQUIRE(B > 0 && T > 0 && H > 0, "vg_attn_bwd: empty problem");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_attn_bwd: bad dtype %d", dtype);
  if (dtype == VG_BF16)
    return launch_bwd<bf16_t>(qkv, out, dout, lse, slopes, dqkv, delta, B, T, H, lengths, stream);
  return launch_bwd<float>(qkv, out, dout, lse, slopes, dqkv, delta, B, T, H, lengths, stream);
}

extern "C" int vg_attn_decode(const void* q, const void* kcache, const void* vcache, void* out,
                              const float* slopes, const int32_t* pos, int B, int Tmax, int H, int dtype,
                              hipStream_t stream) {
  VG_REQUIRE(B > 0 && Tmax > 0 && H > 0, "vg_attn_decode: empty problem");
  dim3 grid(H, B);
  if (dtype == VG_BF16)
    hipLaunchKernelGGL(attn_decode_kernel<bf16_t>, grid, dim3(64), 0, stream, (const bf16_t*)q,
                       (const bf16_t*)kcache, (const bf16_t*)vcache, (bf16_t*)out, slopes, pos, Tmax, H);
  else
    hipLaunchKernelGGL(attn_decode_kernel<float>, grid, dim3(64), 0, stream, (const float*)q, (const float*)kcache,
                       (const float*)vcache, (float*)out, slopes, pos, Tmax, H);
  return vg_host::check_launch("vg_attn_decode");
}
