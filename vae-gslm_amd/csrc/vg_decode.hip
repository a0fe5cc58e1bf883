// Kernels of the autoregressive decode step: one new frame per sequence.
//
//  * vg_gemm_rows: y[M<=64][N] = epi(x[M][K] W[N][K]^T) -- the Linear layers of LVTR.step
//    (reference models/speech/lvtr.py:227-286 -> modules/transformer/layers.py:41-93,
//    modules/attention/attention.py:52,79, modules/linear/layers.py:192) when only a handful of
//    rows exist.  The product is bound by streaming W once from HBM (403 MB of bf16 weights per step at
//    the full config), not by arithmetic, so there is no MFMA and no LDS tile: a block owns 16 output
//    columns, its 16 waves split K, every lane streams 16-byte pieces of one weight row and keeps one
//    fp32 accumulator per input row; partial sums meet in LDS.  Exact fp32 accumulation in both dtypes
//    (the fp32 build of this kernel is the parity path).
//  * vg_attn_decode_append: writes this step's key/value rows into the pre-allocated cache at pos[b]
//    and attends over the pos[b]+1 cached frames (replaces the per-step torch.cat and mask rebuild of
//    modules/attention/attention.py:56-73).
//  * vg_advance: pos[b] += 1 (device-side step counter, so a captured hipGraph can be replayed).
#include <type_traits>
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

using namespace vg;

namespace {

constexpr int RC = 8;           // output columns per block
constexpr int RCHUNK = 512;     // k range of one wave (64 lanes x 8 elements: 1 KiB of a bf16 weight row per load)
constexpr int RMAXW = 8;        // waves per block = min(8, ceil(K / 512)); longer K: a wave walks several ranges
constexpr int RMAXM = 64;       // rows per launch: groups of 8 along grid.y

template <typename T> struct Ld8;     // 8 consecutive elements as floats
template <> struct Ld8<bf16_t> {
  static VG_DEVICE void get(const bf16_t* p, float (&o)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
};
template <> struct Ld8<float> {
  static VG_DEVICE void get(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  }
};

// 8 consecutive elements kept as loaded (4 registers for bf16): kernels that keep dozens of loads in flight convert at use.
// X8<T>: the 8 matching elements of the input vector -- packed bf16 pairs for bf16 weights, so that a dot product of 8 is
// four v_dot2c_f32_bf16 (exact bf16 products, fp32 accumulation) instead of 8 conversions + 8 FMAs: the fused decode
// kernels are bound by VALU issue, not by bytes.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <typename T> struct X8;
template <> struct X8<bf16_t> {
  bf16x2_t p[4];
  VG_DEVICE void set(const float (&x)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = bf16x2_t{(__bf16)x[2 * i], (__bf16)x[2 * i + 1]};
  }
};
template <> struct X8<float> {
  float f[8];
  VG_DEVICE void set(const float (&x)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = x[i];
  }
};
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  bf16x8 v;
  VG_DEVICE void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  VG_DEVICE float dot(const X8<bf16_t>& x, float a) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) a = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{v[2 * i], v[2 * i + 1]}, x.p[i], a, false);
    return a;
  }
  VG_DEVICE void get(float (&o)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
};
template <> struct Raw8<float> {
  f32x4 lo, hi;
  VG_DEVICE void load(const float* p) {
    lo = *reinterpret_cast<const f32x4*>(p);
    hi = *reinterpret_cast<const f32x4*>(p + 4);
  }
  VG_DEVICE float dot(const X8<float>& x, float a) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) a = fmaf(x.f[i], lo[i], a);
#pragma unroll
    for (int i = 0; i < 4; ++i) a = fmaf(x.f[4 + i], hi[i], a);
    return a;
  }
  VG_DEVICE void get(float (&o)[8]) const {
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  }
};

// Block = 8 output columns x all of K; wave w owns k in [512 w, 512 w + 512), lane l the 8 elements at
// 512 w + 8 l: every load is a fully coalesced 16-byte-per-lane row segment, the 8 weight rows of the
// block are requested back to back (8 KiB in flight per wave) and the x rows are read once per lane.
// Each lane then holds MM x 8 partial dot products; a recursive-halving exchange (xor 32, 16, .. 1; 63
// shuffles for 64 values instead of 64 full reductions) leaves lane l with the wave-wide sum of value
// index l (= row l / 8, column l % 8); the waves' sums meet in LDS and wave 0 applies the epilogue.
// TX: type of the input rows and of the residual (T, or float for the fp32 residual stream of the fused decode path:
// vg_gemm_rows_mixed); zero_ptr: an fp32 buffer this launch clears (the next layer's accumulation target).
template <typename TX, typename T, int MM>
__global__ __launch_bounds__(RMAXW * 64) void gemm_rows_kernel(const TX* __restrict__ x, long ldx,
                                                               const T* __restrict__ w, long ldw,
                                                               const float* __restrict__ bias,
                                                               const TX* __restrict__ residual, long ldr,
                                                               void* __restrict__ y, long ldy, int M, int N, int K,
                                                               int act, int out_f32,
                                                               const float* __restrict__ norm_scale, float norm_eps,
                                                               float* __restrict__ zero_ptr, int zero_n) {
  constexpr int V = MM * RC;            // values per lane before the exchange
  constexpr int PER = V / 64;           // values per lane after it (1 for MM = 8, 2 for MM = 16)
  __shared__ float red[RMAXW][V];
  __shared__ float ssq[RMAXW][MM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int n0 = blockIdx.x * RC;
  if (zero_ptr && blockIdx.y == 0)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += gridDim.x * blockDim.x) zero_ptr[i] = 0.f;
  // more than MM rows: blockIdx.y picks a group of MM (the groups read the same weight rows: the second one from L2)
  {
    const int r0 = blockIdx.y * MM;
    x += (long)r0 * ldx;
    if (residual) residual += (long)r0 * ldr;
    y = out_f32 ? (void*)(reinterpret_cast<float*>(y) + (long)r0 * ldy) : (void*)(reinterpret_cast<T*>(y) + (long)r0 * ldy);
    M = min(M - r0, MM);
  }
  float v[V];
#pragma unroll
  for (int i = 0; i < V; ++i) v[i] = 0.f;
  // fused RMSNorm prologue (norm_scale != null): y = act((x * rstd * g) W^T + b).  rstd is a per-row factor,
  // so the products use x * g and rstd multiplies the finished sums; every block recomputes sum(x^2).
  float sq[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) sq[m] = 0.f;
  for (int k0 = wave * RCHUNK + lane * 8; k0 < K; k0 += nwaves * RCHUNK) {
    // every load of the step is requested before the first FMA: the RC weight rows, the MM input rows (row index
    // clamped to M - 1: no per-row branch -- with `if (m < M)` around each row's load the compiler waited for every
    // row separately, eight dependent L2 round trips per launch) and the norm scale: ONE memory round trip
    float wv[RC][8], gv[8];
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      const int n = min(n0 + c, N - 1);
      Ld8<T>::get(w + (long)n * ldw + k0, wv[c]);
    }
    if (norm_scale) Ld8<float>::get(norm_scale + k0, gv);
    if constexpr (MM <= 8) {
      float xv[MM][8];
#pragma unroll
      for (int m = 0; m < MM; ++m) Ld8<TX>::get(x + (long)min(m, M - 1) * ldx + k0, xv[m]);
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        if (norm_scale) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sq[m] = fmaf(xv[m][e], xv[m][e], sq[m]);
            xv[m][e] *= gv[e];
          }
        }
#pragma unroll
        for (int c = 0; c < RC; ++c) {
          float a = v[m * RC + c];
#pragma unroll
          for (int e = 0; e < 8; ++e) a = fmaf(xv[m][e], wv[c][e], a);
          v[m * RC + c] = a;
        }
      }
    } else {        // 16 rows: 128 accumulators leave no registers for more than one input row at a time
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        if (m < M) {
          float xv[8];
          Ld8<TX>::get(x + (long)m * ldx + k0, xv);
          if (norm_scale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              sq[m] = fmaf(xv[e], xv[e], sq[m]);
              xv[e] *= gv[e];
            }
          }
#pragma unroll
          for (int c = 0; c < RC; ++c) {
            float a = v[m * RC + c];
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(xv[e], wv[c][e], a);
            v[m * RC + c] = a;
          }
        }
      }
    }
  }
  // recursive halving: after the stage with offset o a lane keeps the half selected by its bit o
#pragma unroll
  for (int o = 32, cur = V; o >= 1; o >>= 1, cur >>= 1) {
    const int half = cur >> 1;
    const bool up = (lane & o) != 0;
#pragma unroll
    for (int i = 0; i < V / 2; ++i) {
      if (i < half) {
        const float send = up ? v[i] : v[i + half];
        const float keep = up ? v[i + half] : v[i];
        v[i] = keep + __shfl_xor(send, o, 64);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) red[wave][lane * PER + j] = v[j];
  if (norm_scale) {
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      const float t = wave_sum(sq[m]);
      if (lane == 0) ssq[wave][m] = t;
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int idx = lane * PER + j, m = idx / RC, n = n0 + idx % RC;
      if (m < M && n < N) {
        float r = 0.f;
        for (int ww = 0; ww < nwaves; ++ww) r += red[ww][idx];
        if (norm_scale) {
          float ss = 0.f;
          for (int ww = 0; ww < nwaves; ++ww) ss += ssq[ww][m];
          r *= rsqrtf(ss / (float)K + norm_eps);
        }
        if (bias) r += bias[n];
        if (act == VG_ACT_RELU) r = fmaxf(r, 0.f);
        else if (act == VG_ACT_GELU) r = gelu_erf(r);
        else if (act == VG_ACT_SILU) r = silu(r);
        if (residual) r += to_f32<TX>(residual[(long)m * ldr + n]);
        if (out_f32) reinterpret_cast<float*>(y)[(long)m * ldy + n] = r;
        else reinterpret_cast<T*>(y)[(long)m * ldy + n] = from_f32<T>(r);
      }
    }
  }
}

// ---- more than 16 rows (round 5): the same product on the matrix cores.  The 8-row kernel above serves M rows as
// ceil(M / 8) groups along grid.y, i.e. it streams the weights once per group: at the reference's inference batch (64
// sequences, configs/infer/speech/vae-gslm.yaml:27) that is eight passes over the 403 MB of a frame's weights, all but
// the first from L2, and 8 x the VALU work of the dot products.  Here a block owns 16 output columns (16 weight rows)
// and ALL rows: W is read once, straight from HBM into registers as v_mfma_f32_16x16x32_bf16 A fragments (a lane loads
// the 16 bytes k0 + 8 (lane >> 4) .. + 7 of weight row n0 + (lane & 15): four lanes cover 64 contiguous bytes of a
// row), the input rows are the B fragments (bf16 as loaded; fp32 rows of the fused path are rounded to bf16 -- the
// weights are bf16 anyway), one 16 x 16 accumulator per 16 rows.  The block's waves split K (interleaved 32-deep
// steps, four of them in flight per wave), meet in LDS, and the whole block runs the epilogue.  The RMSNorm prologue
// multiplies the input fragments by the norm's scale on the fly and accumulates sum(x^2) from the same registers.
// fp32 weights (the parity path) and M <= 16 stay on the exact-fp32 kernel above.
template <typename TX> struct Frag8;       // 8 consecutive input elements -> (bf16x8 operand, their fp32 values on request)
template <> struct Frag8<bf16_t> {
  bf16x8 v;
  VG_DEVICE void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  VG_DEVICE float sumsq() const {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x2_t t = {v[2 * i], v[2 * i + 1]};
      a = __builtin_amdgcn_fdot2_f32_bf16(t, t, a, false);
    }
    return a;
  }
  VG_DEVICE bf16x8 operand(const float* g) const {
    if (g == nullptr) return v;
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)((float)v[i] * g[i]);
    return r;
  }
};
template <> struct Frag8<float> {
  f32x4 lo, hi;
  VG_DEVICE void load(const float* p) {
    lo = *reinterpret_cast<const f32x4*>(p);
    hi = *reinterpret_cast<const f32x4*>(p + 4);
  }
  VG_DEVICE float sumsq() const {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) a = fmaf(lo[i], lo[i], fmaf(hi[i], hi[i], a));
    return a;
  }
  VG_DEVICE bf16x8 operand(const float* g) const {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[i] = (bf16_t)(g ? lo[i] * g[i] : lo[i]);
      r[4 + i] = (bf16_t)(g ? hi[i] * g[4 + i] : hi[i]);
    }
    return r;
  }
};

constexpr int MFW = 8;          // waves per block of the MFMA rows kernel (K split 8 ways inside the block)
constexpr int MFU = 4;          // 32-deep k-steps in flight per wave

// ACC (round 5, the two N = 1024 products of a layer at 17..64 sequences: 64 blocks each pulling all the input rows
// -- 640 KB at K = 4096 -- through one CU): gridDim.y blocks split K, every block adds its 16 x M tile to a ZEROED fp32 y
// with float atomics (1024 adds per block), the y = 0 slice adds bias and the fp32 residual; no activation, no norm.
template <typename TX, bool ACC = false>
__global__ __launch_bounds__(MFW * 64) void gemm_rows_mfma_kernel(const TX* __restrict__ x, long ldx,
                                                                  const bf16_t* __restrict__ w, long ldw,
                                                                  const float* __restrict__ bias,
                                                                  const void* __restrict__ residual_, long ldr,
                                                                  void* __restrict__ y, long ldy, int M, int N, int K,
                                                                  int act, int out_f32,
                                                                  const float* __restrict__ norm_scale, float norm_eps,
                                                                  float* __restrict__ zero_ptr, int zero_n) {
  __shared__ float red[MFW][16][64 + 1];
  __shared__ float ssq[MFW][64];
  typedef typename std::conditional<ACC, float, TX>::type TR;          // type of the residual rows
  const TR* __restrict__ residual = reinterpret_cast<const TR*>(residual_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16;
  if (zero_ptr && blockIdx.y == 0)
    for (int i = blockIdx.x * blockDim.x + tid; i < zero_n; i += gridDim.x * blockDim.x) zero_ptr[i] = 0.f;
  const bool lead = !ACC || blockIdx.y == 0;                            // the block that adds bias and residual
  const int mt = (M + 15) >> 4;                        // 16-row input tiles (wave-uniform)
  const int kq = (lane >> 4) * 8;
  // the epilogue's operands (bias, residual) are requested NOW: a launch of this kernel is three dependent memory
  // round trips otherwise (operands, epilogue operands, stores), and at 64 rows the step is 69 such launches
  float pre_b[2] = {0.f, 0.f}, pre_r[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * MFW * 64, nn = idx & 15, m = idx >> 4, n = n0 + nn;
    if (m < M && n < N && lead) {
      if (bias) pre_b[i] = bias[n];
      if (residual) pre_r[i] = to_f32<TR>(residual[(long)m * ldr + n]);
    }
  }
  const bf16_t* __restrict__ wp = w + (long)min(n0 + (lane & 15), N - 1) * ldw + kq;
  const TX* xp[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) xp[t] = x + (long)min(16 * t + (lane & 15), M - 1) * ldx + kq;
  f32x4 acc[4];
  float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // k-steps of 32, dealt round-robin to the waves: step j of the block belongs to wave j % MFW (ACC: the block's K slice)
  const int per = ACC ? ((K >> 5) + gridDim.y - 1) / gridDim.y : (K >> 5);
  const int jbeg = ACC ? blockIdx.y * per : 0;
  const int nsteps = ACC ? min(K >> 5, jbeg + per) : (K >> 5);
  for (int j0 = jbeg + wave; j0 < nsteps; j0 += MFW * MFU) {
    bf16x8 wf[MFU];
    Frag8<TX> xf[MFU][4];
    float g[MFU][8];
    // every load of the MFU steps is requested before the first product
#pragma unroll
    for (int u = 0; u < MFU; ++u) {
      const int j = min(j0 + u * MFW, nsteps - 1);      // (past the end: a valid address, the product is dropped below)
      wf[u] = *reinterpret_cast<const bf16x8*>(wp + 32 * j);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < mt) xf[u][t].load(xp[t] + 32 * j);
      if (norm_scale) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(norm_scale + 32 * j + kq);
        const f32x4 c = *reinterpret_cast<const f32x4*>(norm_scale + 32 * j + kq + 4);
        g[u][0] = a[0]; g[u][1] = a[1]; g[u][2] = a[2]; g[u][3] = a[3];
        g[u][4] = c[0]; g[u][5] = c[1]; g[u][6] = c[2]; g[u][7] = c[3];
      }
    }
#pragma unroll
    for (int u = 0; u < MFU; ++u) {
      if (j0 + u * MFW >= nsteps) break;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t >= mt) break;
        if (norm_scale) sq[t] += xf[u][t].sumsq();
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u][t].operand(norm_scale ? g[u] : nullptr), acc[t], 0, 0, 0);
      }
    }
  }
  // accumulator tile t: value (n = 4 (lane >> 4) + i, m = 16 t + (lane & 15)) in acc[t][i]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][4 * (lane >> 4) + i][16 * t + (lane & 15)] = t < mt ? acc[t][i] : 0.f;
  if (norm_scale) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = sq[t];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) ssq[wave][16 * t + lane] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * MFW * 64, nn = idx & 15, m = idx >> 4, n = n0 + nn;
    if (m >= M || n >= N) continue;
    float r = 0.f;
#pragma unroll
    for (int ww = 0; ww < MFW; ++ww) r += red[ww][nn][m];
    if (norm_scale) {
      float ss = 0.f;
#pragma unroll
      for (int ww = 0; ww < MFW; ++ww) ss += ssq[ww][m];
      r *= rsqrtf(ss / (float)K + norm_eps);
    }
    r += pre_b[i];
    if (act == VG_ACT_RELU) r = fmaxf(r, 0.f);
    else if (act == VG_ACT_GELU) r = gelu_erf(r);
    else if (act == VG_ACT_SILU) r = silu(r);
    r += pre_r[i];
    if constexpr (ACC) atomicAdd(reinterpret_cast<float*>(y) + (long)m * ldy + n, r);
    else if (out_f32) reinterpret_cast<float*>(y)[(long)m * ldy + n] = r;
    else reinterpret_cast<bf16_t*>(y)[(long)m * ldy + n] = (bf16_t)r;
  }
}

// one block of 4 waves per (b, h): append this step's k/v rows to the cache at pos[b], then
// softmax(q.k / 8 - slope (n - 1 - j)) v over the n = pos[b] + 1 cached frames.
//  lanes = 8 cached frames x 8 chunks of 8 head channels: one wave-instruction reads 8 whole 128-byte cache rows
//  (16 bytes per lane).  Round 3: ONE pass with the loads of a whole chunk of frames in flight.  The round-2 kernel
//  walked the cache twice in loops with a single dependent 16-byte load per lane and iteration (n / 32 memory round
//  trips per pass, three block barriers, the new row written to the cache and read back): 24 us at 500 cached
//  frames for 90 KB of cache.  Here every lane group (wave, frame slot) owns the frames j = 32 i + 8 wave + slot,
//  requests CH of them at once (K and V rows together: 2 CH loads in flight per lane, the new frame's rows straight
//  from the qkv row), keeps its own running maximum / sum / 8-channel value sum (online softmax per lane group, a
//  rescale only when the chunk raises the maximum) and the 32 lane groups meet once at the end: three shuffle steps
//  inside the wave, one LDS exchange across the four waves, ONE block barrier.
//  static LDS: 4 x (64 + 2) floats
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_append_kernel(const T* __restrict__ qkv, T* __restrict__ kc,
                                                                 T* __restrict__ vc, T* __restrict__ out,
                                                                 const float* __restrict__ slopes,
                                                                 const int* __restrict__ pos, int Tmax, int H) {
  constexpr int DH = 64, CH = 6;
  __shared__ float part[4][DH + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.x, b = blockIdx.y;
  const int D = H * DH;
  const int p0 = min(pos[b], Tmax - 1);
  const T* __restrict__ row = qkv + (long)b * 3 * D + h * DH;
  const long cbase = ((long)b * Tmax) * D + h * DH;
  if (wave == 0) {                                  // the cache rows of this frame, for the steps to come
    kc[cbase + (long)p0 * D + lane] = row[D + lane];
    vc[cbase + (long)p0 * D + lane] = row[2 * D + lane];
  }
  const int n = p0 + 1;
  const int sub = lane >> 3, ch = lane & 7;         // cached frame inside a group of 8, channel chunk
  float q8[8];
  Ld8<T>::get(row + ch * 8, q8);
  const float slope = slopes[h] * 1.44269504088896340736f;
#pragma unroll
  for (int e = 0; e < 8; ++e) q8[e] *= 0.125f * 1.44269504088896340736f;     // log2 domain
  float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j0 = wave * 8 + sub; j0 < n; j0 += 32 * CH) {
    float k8[CH][8], v8[CH][8];
    // all rows of the chunk are requested before the first one is used; frame p0 comes from the qkv row itself
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int j = j0 + 32 * c;
      const int jc = min(j, n - 1);
      const T* kp = jc == p0 ? row + D + ch * 8 : kc + cbase + (long)jc * D + ch * 8;
      const T* vp = jc == p0 ? row + 2 * D + ch * 8 : vc + cbase + (long)jc * D + ch * 8;
      Ld8<T>::get(kp, k8[c]);
      Ld8<T>::get(vp, v8[c]);
    }
    float sc[CH], cmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) t = fmaf(q8[e], k8[c][e], t);
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      const int j = j0 + 32 * c;
      sc[c] = j < n ? t - slope * (float)(n - 1 - j) : -INFINITY;
      cmax = fmaxf(cmax, sc[c]);
    }
    if (cmax > m) {                                  // (the first chunk always has a valid frame: j0 < n)
      const float f = exp2f(m - cmax);
      l *= f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= f;
      m = cmax;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float pj = exp2f(sc[c] - m);
      l += pj;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, v8[c][e], acc[e]);
    }
  }
  // ---- the 8 frame slots of the wave (lanes 8 apart share a channel chunk)
  float mw = m;
  mw = fmaxf(mw, __shfl_xor(mw, 8, 64));
  mw = fmaxf(mw, __shfl_xor(mw, 16, 64));
  mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
  const float f = m == -INFINITY ? 0.f : exp2f(m - mw);      // a slot that owned no frame contributes nothing
  l *= f;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] *= f;
#pragma unroll
  for (int o = 8; o <= 32; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (sub == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[wave][ch * 8 + e] = acc[e];
    if (ch == 0) {
      part[wave][DH] = mw;
      part[wave][DH + 1] = l;
    }
  }
  __syncthreads();
  if (tid < DH) {
    const float m0 = part[0][DH], m1 = part[1][DH], m2 = part[2][DH], m3 = part[3][DH];
    const float mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    const float f0 = m0 == -INFINITY ? 0.f : exp2f(m0 - mx), f1 = m1 == -INFINITY ? 0.f : exp2f(m1 - mx);
    const float f2 = m2 == -INFINITY ? 0.f : exp2f(m2 - mx), f3 = m3 == -INFINITY ? 0.f : exp2f(m3 - mx);
    const float den = part[0][DH + 1] * f0 + part[1][DH + 1] * f1 + part[2][DH + 1] * f2 + part[3][DH + 1] * f3;
    const float num = part[0][tid] * f0 + part[1][tid] * f1 + part[2][tid] * f2 + part[3][tid] * f3;
    out[(long)b * D + h * DH + tid] = from_f32<T>(num / den);
  }
}

// Fused attention sub-layer of the decode step (round 3): one block of 4 waves per (head, sequence) does what
// vg_gemm_rows (norm1 + QKV), vg_attn_decode_append and vg_gemm_rows (out-projection + residual) did in three
// launches -- three graph nodes of ~4.5 us floor each for a few microseconds of work:
//   xn      = x[b] * g1 (the row's rstd multiplies the finished dot products)          RMSNorm, reference norm.py:20-29
//   q, k, v = rows {q, k, v} x {64 h .. 64 h + 63} of Wqkv . xn * rstd + bias           attention.py:52 (this head only)
//   cache[b][pos[b]] <- k, v;   ctx = softmax(q . K^T / 8 - slope * distance) V         attention.py:56-77
//   x1[b]  += Wo[:, 64 h .. 64 h + 63] . ctx  (+ x[b] + bo from the h = 0 block)        attention.py:79, layers.py:52
// The residual stream is fp32 and x1 is an ACCUMULATION target: it must be zero when the launch starts (the H blocks of
// a row add into it with fp32 atomics; the sum order, and with it the last bit, varies from run to run); `zero_buf` is
// the buffer the NEXT layer accumulates into, cleared here.  Weight slices per block: 192 rows of Wqkv (384 KB at
// d = 1024, bf16) + a 64-column band of Wo (128 KB): the B blocks of a head read the same slices (L2 / MALL hits).
#ifdef VG_LAB_DSTAMP
__device__ long long* g_dstamp = nullptr;     // diagnostic build only (tools/lab/variant.sh dstamp vg_decode.hip -DVG_LAB_DSTAMP)
#define DSTAMP(i) do { if (g_dstamp && threadIdx.x == 0) g_dstamp[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif
// Latency, not bandwidth, bounds this kernel (a block's data is ~0.6 MB, but every dependent wait is a 1.5-2 us memory
// round trip): 16 waves, so that each phase is ONE pass with all of its loads in flight -- 12 rows of Wqkv per wave
// (24 loads of 16 bytes per lane, requested BEFORE the row is normalised: weights do not depend on it), 8 rows of the
// Wo band per lane requested before the cache walk (they land while the soft-max runs), 384 cached frames per pass.
// A first version with 4 waves and 8-row groups walked 14 round trips in sequence: 28 us per launch.
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void attn_layer_decode_kernel(const float* __restrict__ x, const float* __restrict__ g1,
                                                                    float eps, const T* __restrict__ wqkv,
                                                                    const float* __restrict__ bqkv, const T* __restrict__ wo,
                                                                    const float* __restrict__ bo, T* __restrict__ kc,
                                                                    T* __restrict__ vc, const float* __restrict__ slopes,
                                                                    const int* __restrict__ pos, int Tmax, int H,
                                                                    float* __restrict__ x1, float* __restrict__ zero_buf) {
  constexpr int DH = 64, DMAX = 1024, NT = NW * 64;
  constexpr int RPW = 3 * DH / NW;                 // rows of the QKV projection per wave
  constexpr int RG = sizeof(T) == 2 ? RPW / 2 : RPW / 4;              // rows per pass: 12 / 6 loads in flight per lane
  // (all 12 rows at once need 96 registers of raw weights on top of the row itself: over the 128 of a 16-wave block)
  constexpr int CH = 384 / (NW * 8) > 0 ? 384 / (NW * 8) : 1;         // cached frames per lane group and pass
  static_assert(3 * DH % NW == 0 && RPW % RG == 0, "rows of the head's projection must divide among the waves");
  __shared__ __attribute__((aligned(16))) float xn[DMAX];
  __shared__ __attribute__((aligned(16))) float ypart[DMAX];
  __shared__ float qkv_s[3 * DH];
  __shared__ float part[NW][DH + 2];
  __shared__ float ctx_s[DH];
  __shared__ float redw[NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.x, b = blockIdx.y;
  const int D = H * DH;
  const float* __restrict__ xr = x + (long)b * D;
  DSTAMP(0);
  const int k0 = lane * 8, k1 = lane * 8 + 512;
  auto wrow = [&](int r) { return wqkv + ((long)(r >> 6) * D + h * DH + (r & 63)) * D; };
  // ---- first pass of this wave's QKV rows: in flight while the row is normalised
  Raw8<T> w0[RG], w1[RG];
#pragma unroll
  for (int c = 0; c < RG; ++c) {
    const T* wr = wrow(wave * RPW + c);
    w0[c].load(wr + min(k0, D - 8));
    w1[c].load(wr + min(k1, D - 8));
  }
  // biases of this wave's rows: lane c holds row c's (a load inside the row loop is a dependent round trip per row and
  // its wait drains every weight load in flight)
  float bias_l = 0.f;
  if (bqkv && lane < RPW) {
    const int r = wave * RPW + lane;
    bias_l = bqkv[(r >> 6) * D + h * DH + (r & 63)];
  }
  // ---- 1. the row, scaled by the norm weight; sum of squares
  float ss = 0.f;
  for (int i = tid * 4; i < D; i += NT * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
    const f32x4 g = *reinterpret_cast<const f32x4*>(g1 + i);
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    *reinterpret_cast<f32x4*>(xn + i) = f32x4{v[0] * g[0], v[1] * g[1], v[2] * g[2], v[3] * g[3]};
  }
  ss = wave_sum(ss);
  if (lane == 0) redw[wave] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) tot += redw[w];
  const float rstd = rsqrtf(tot / (float)D + eps);
  DSTAMP(1);
  // ---- 2. this head's 192 rows of the QKV projection
  {
    X8<T> xv0, xv1;
    {
      float t0[8], t1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        t0[e] = k0 < D ? xn[k0 + e] : 0.f;
        t1[e] = k1 < D ? xn[k1 + e] : 0.f;
      }
      xv0.set(t0);
      xv1.set(t1);
    }
#pragma unroll
    for (int g0 = 0; g0 < RPW; g0 += RG) {
      if (g0 > 0) {
#pragma unroll
        for (int c = 0; c < RG; ++c) {
          const T* wr = wrow(wave * RPW + g0 + c);
          w0[c].load(wr + min(k0, D - 8));
          w1[c].load(wr + min(k1, D - 8));
        }
      }
#pragma unroll
      for (int c = 0; c < RG; ++c) {
        float a = w1[c].dot(xv1, w0[c].dot(xv0, 0.f));
        a = wave_sum(a);
        const int r = wave * RPW + g0 + c;
        const float bias = __shfl(bias_l, g0 + c, 64);
        if (lane == 0) qkv_s[r] = to_f32<T>(from_f32<T>(a * rstd + bias));     // the projection's output dtype
      }
    }
  }
  DSTAMP(2);
  // ---- this head's 64-column band of the out-projection, requested now: 8 lanes per weight row (one 128-byte line),
  //      8 rows per wave-load; the rows land while the cache is walked
  const int sub = lane >> 3, ch = lane & 7;
  constexpr int WU = DMAX / (NW * 8);              // loads per lane that cover DMAX rows
  Raw8<T> wv[WU];
#pragma unroll
  for (int u = 0; u < WU; ++u) {
    const int n = min(u * (NW * 8) + wave * 8 + sub, D - 1);
    wv[u].load(wo + (long)n * D + h * DH + ch * 8);
  }
  // ... and the first pass over the cache (the cached frames do not depend on this frame's projection either)
  const int p0 = min(pos[b], Tmax - 1);
  const long cbase = ((long)b * Tmax) * D + h * DH;
  Raw8<T> kr[CH], vr[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int jc = max(min(wave * 8 + sub + NW * 8 * c, p0 - 1), 0);
    kr[c].load(kc + cbase + (long)jc * D + ch * 8);
    vr[c].load(vc + cbase + (long)jc * D + ch * 8);
  }
  __syncthreads();
  DSTAMP(3);
  // ---- 3. attention over the p0 cached frames (as attn_decode_append_kernel) and the new frame (from LDS)
  constexpr float QSCALE = 0.125f * 1.44269504088896340736f;      // 1 / sqrt(64), log2 domain
  X8<T> q8;
  {
    float t0[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t0[e] = qkv_s[ch * 8 + e];
    q8.set(t0);            // exact: the projection was rounded to T above
  }
  const float slope = slopes[h] * 1.44269504088896340736f;
  float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j0 = wave * 8 + sub; j0 < p0; j0 += NW * 8 * CH) {
    if (j0 >= NW * 8 * CH) {                       // later passes (more than 384 cached frames)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int jc = min(j0 + NW * 8 * c, p0 - 1);
        kr[c].load(kc + cbase + (long)jc * D + ch * 8);
        vr[c].load(vc + cbase + (long)jc * D + ch * 8);
      }
    }
    float v8[CH][8];
#pragma unroll
    for (int c = 0; c < CH; ++c) vr[c].get(v8[c]);
    float sc[CH], cmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float t = kr[c].dot(q8, 0.f);
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      const int j = j0 + NW * 8 * c;
      sc[c] = j < p0 ? t * QSCALE - slope * (float)(p0 - j) : -INFINITY;
      cmax = fmaxf(cmax, sc[c]);
    }
    if (cmax > m) {
      const float f = exp2f(m - cmax);
      l *= f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= f;
      m = cmax;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float pj = exp2f(sc[c] - m);
      l += pj;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, v8[c][e], acc[e]);
    }
  }
  DSTAMP(4);
  if (wave == 0) {                                  // the cache rows of this frame, for the steps to come (row p0: not read above)
    kc[cbase + (long)p0 * D + lane] = from_f32<T>(qkv_s[DH + lane]);
    vc[cbase + (long)p0 * D + lane] = from_f32<T>(qkv_s[2 * DH + lane]);
  }
  {   // the new frame (distance 0): every lane group computes its score, group (wave 0, slot 0) owns it
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) t = fmaf(qkv_s[ch * 8 + e], qkv_s[DH + ch * 8 + e], t);
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    t += __shfl_xor(t, 4, 64);
    t *= QSCALE;
    if (wave == 0 && sub == 0) {
      if (t > m) {
        const float f = exp2f(m - t);
        l *= f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= f;
        m = t;
      }
      const float pj = exp2f(t - m);
      l += pj;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, qkv_s[2 * DH + ch * 8 + e], acc[e]);
    }
  }
  float mw = m;
  mw = fmaxf(mw, __shfl_xor(mw, 8, 64));
  mw = fmaxf(mw, __shfl_xor(mw, 16, 64));
  mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
  const float fw = m == -INFINITY ? 0.f : exp2f(m - mw);
  l *= fw;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] *= fw;
#pragma unroll
  for (int o = 8; o <= 32; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (sub == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[wave][ch * 8 + e] = acc[e];
    if (ch == 0) {
      part[wave][DH] = mw;
      part[wave][DH + 1] = l;
    }
  }
  __syncthreads();
  if (tid < DH) {
    float mx = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) mx = fmaxf(mx, part[w][DH]);
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float mwv = part[w][DH];
      const float f = mwv == -INFINITY ? 0.f : exp2f(mwv - mx);
      den = fmaf(part[w][DH + 1], f, den);
      num = fmaf(part[w][tid], f, num);
    }
    ctx_s[tid] = to_f32<T>(from_f32<T>(num / den));       // the attention output's dtype
  }
  __syncthreads();
  DSTAMP(5);
  // ---- 4. out-projection band x context
  {
    X8<T> c8;
    {
      float t0[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t0[e] = ctx_s[ch * 8 + e];
      c8.set(t0);          // exact: the context was rounded to T above
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      float t = wv[u].dot(c8, 0.f);
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      const int n = u * (NW * 8) + wave * 8 + sub;
      if (ch == 0 && n < D) ypart[n] = t;
    }
  }
  DSTAMP(6);
  float radd[DMAX / NT > 0 ? DMAX / NT : 1];         // the h = 0 block's x + bo, requested before the barrier
#pragma unroll
  for (int i = 0; i < (DMAX / NT > 0 ? DMAX / NT : 1); ++i) {
    const int n = tid + i * NT;
    radd[i] = (h == 0 && n < D) ? xr[n] + (bo ? bo[n] : 0.f) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < (DMAX / NT > 0 ? DMAX / NT : 1); ++i) {
    const int n = tid + i * NT;
    if (n < D) atomicAdd(x1 + (long)b * D + n, ypart[n] + radd[i]);
  }
  DSTAMP(7);
  if (zero_buf && tid < DH) zero_buf[(long)b * D + h * DH + tid] = 0.f;
#ifdef VG_LAB_DSTAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DSTAMP(8);
#endif
}

// frame embedding of the step: out[b][c] = E[id_b][c] + relu(Wf[c][:] . z_b + bf[c])   (one wave per
// sequence, lane-strided over the embedding width; models/speech/lvtr.py:161-168 fuse_inputs)
template <typename T>
__global__ __launch_bounds__(64) void embed_fuse_kernel(const float* __restrict__ frame, int ldf,
                                                        const float* __restrict__ emb, int vocab, int E,
                                                        const float* __restrict__ wf, const float* __restrict__ bf,
                                                        int latent, T* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* fr = frame + (long)b * ldf;
  int id = (int)fr[0];
  id = min(max(id, 0), vocab - 1);
  for (int c = lane; c < E; c += 64) {
    float a = bf ? bf[c] : 0.f;
    for (int j = 0; j < latent; ++j) a = fmaf(wf[(long)c * latent + j], fr[1 + j], a);
    out[(long)b * E + c] = from_f32<T>(emb[(long)id * E + c] + fmaxf(a, 0.f));
  }
}

// token draw of the step: categorical sample from softmax(logits / temperature) by inverse CDF with a
// supplied uniform number per sequence; writes the id (as float) to frame[b][0] and advances pos[b].
// One wave per sequence (models/speech/lvtr.py:276-284: softmax + multinomial + cat).
__global__ __launch_bounds__(64) void sample_token_kernel(const float* __restrict__ logits, int V, float inv_temp,
                                                          const float* __restrict__ uniform, float* __restrict__ frame,
                                                          int ldf, int* __restrict__ pos) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* lg = logits + (long)b * V;
  float mx = -INFINITY;
  for (int i = lane; i < V; i += 64) mx = fmaxf(mx, lg[i] * inv_temp);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < V; i += 64) sum += expf(lg[i] * inv_temp - mx);
  sum = wave_sum(sum);
  const float target = uniform[b] * sum;
  // ids are visited in order in chunks of 64: inclusive prefix sums inside the chunk, running total across chunks
  float run = 0.f;
  int pick = V - 1;
  bool done = false;
  for (int base = 0; base < V && !done; base += 64) {
    const int i = base + lane;
    const float p = i < V ? expf(lg[i] * inv_temp - mx) : 0.f;
    float c = p;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float t = __shfl_up(c, o, 64);
      if (lane >= o) c += t;
    }
    const bool hit = i < V && run + c > target;
    const unsigned long long mask = __ballot(hit);
    if (mask) {
      pick = base + __builtin_ctzll(mask);
      done = true;
    }
    run += __shfl(c, 63, 64);
  }
  if (lane == 0) {
    frame[(long)b * ldf] = (float)pick;
    if (pos) pos[b] += 1;
  }
}

// Random draws of one decode step from a counter-based generator (Philox 4x32-10, Salmon et al. 2011) keyed by
// (seed, sequence, pos[b]): normal[b][0..nn) (Box-Muller on pairs) and uniform[b] in [0, 1) in ONE launch.  A replayed
// hipGraph draws fresh numbers because pos advances on the device; torch's graph-safe generator costs two fill launches
// per replay plus one launch per distribution (4 graph nodes of ~4.7 us per frame).
VG_DEVICE void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__global__ __launch_bounds__(64) void decode_noise_kernel(unsigned long long seed, const int* __restrict__ pos,
                                                          const int* __restrict__ epoch, float* __restrict__ normal, int nn,
                                                          float* __restrict__ uniform, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const unsigned p = (unsigned)pos[b], k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  // draw epoch (device word the caller bumps whenever the frame counter is rewound: a new prompt on the same session):
  // the fourth counter word, so the same (sequence, frame) draws fresh numbers in every epoch
  const unsigned tag = 0x5eedu + (epoch ? (unsigned)epoch[0] * 0x9E3779B9u : 0u);
  auto pair = [&](unsigned w0, unsigned w1, int o) {      // Box-Muller: two words -> normals o, o + 1
    const float u0 = ((float)w0 + 0.5f) * 2.3283064365386963e-10f;          // (0, 1]: 2^-32 (w + 1/2)
    const float u1 = ((float)w1 + 0.5f) * 2.3283064365386963e-10f;
    const float r = sqrtf(-2.f * logf(u0)), a = 6.28318530717958647692f * u1;
    if (o < nn) normal[(long)b * nn + o] = r * cosf(a);
    if (o + 1 < nn) normal[(long)b * nn + o + 1] = r * sinf(a);
  };
  // block 0 of this (sequence, frame): words 0, 1 -> normals 0, 1; word 3 -> the uniform draw (24 bits: never 1.0)
  unsigned c[4] = {p, (unsigned)b, 0u, tag};
  philox4x32_10(c, k0, k1);
  if (uniform) uniform[b] = (float)(c[3] >> 8) * (1.0f / 16777216.0f);
  pair(c[0], c[1], 0);
  for (int k = 1; 4 * k - 2 < nn; ++k) {                  // block k -> normals 4 k - 2 .. 4 k + 1
    unsigned d[4] = {p, (unsigned)b, (unsigned)k, tag};
    philox4x32_10(d, k0, k1);
    pair(d[0], d[1], 4 * k - 2);
    pair(d[2], d[3], 4 * k);
  }
}

// Touch a byte range with few, narrow workgroups so that it sits in the 256 MB Infinity Cache when the kernels that need
// it arrive (decode: layer L + 1's weights while layer L computes).  The loaded words are folded into a value that is
// stored only under a condition that never holds, so the loads cannot be dropped.
__global__ __launch_bounds__(256) void touch_kernel(const f32x4* __restrict__ p, long n16, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = p[i];
    acc ^= __float_as_uint(v[0]) ^ __float_as_uint(v[1]) ^ __float_as_uint(v[2]) ^ __float_as_uint(v[3]);
  }
  if (acc == 0x9E3779B9u && sink != nullptr && n16 < 0) *sink = acc;
}

__global__ void advance_kernel(int* __restrict__ pos, int n, int by) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[i] += by;
}

template <typename TX, typename T>
int launch_rows(const void* x, long ldx, const void* w, long ldw, const float* bias, const void* res, long ldr, void* y,
                long ldy, int M, int N, int K, int act, int out_f32, const float* norm_scale, float norm_eps,
                float* zero_ptr, int zero_n, hipStream_t stream) {
  // bf16 weights: the matrix-core kernel (weights streamed once, 16 output columns per block).  Built for more than 16
  // rows, it is the faster one at EVERY row count (round 5, decode step per frame with the threshold at 17 / 1:
  // B = 1 0.580 / 0.502 ms, B = 8 0.604 / 0.546, B = 16 0.647 / 0.600).  VG_ROWS_MFMA=<M> moves the threshold (rows
  // from which it is used; 0 = never: the 8-row kernel), for A/B runs
  if constexpr (std::is_same<T, bf16_t>::value) {
    static const int from = [] { const char* e = getenv("VG_ROWS_MFMA"); return e ? atoi(e) : 1; }();
    if (from > 0 && M >= from && K % 32 == 0 && ((uintptr_t)x % 16) == 0 && (norm_scale == nullptr || ((uintptr_t)norm_scale % 16) == 0)) {
      dim3 grid((N + 15) / 16), block(MFW * 64);
      gemm_rows_mfma_kernel<TX><<<grid, block, 0, stream>>>((const TX*)x, ldx, (const bf16_t*)w, ldw, bias, (const void*)res, ldr,
                                                            y, ldy, M, N, K, act, out_f32, norm_scale, norm_eps, zero_ptr, zero_n);
      return vg_host::check_launch("vg_gemm_rows");
    }
  }
  int nwaves = (K + RCHUNK - 1) / RCHUNK;
  if (nwaves > RMAXW) nwaves = RMAXW;
  // more than 8 rows run as groups of up to 8 (grid.y) on the 8-row kernel: its 16-row instance keeps one input row
  // in registers at a time (128 accumulators) and walks the rows' loads one after the other -- B = 16 decode 1.16 ms per
  // frame against 0.68 at B = 8 before this
  static const int rows16 = [] { const char* e = getenv("VG_ROWS16"); return e ? atoi(e) : 0; }();
  if (M > 16) VG_REQUIRE(!rows16, "vg_gemm_rows: VG_ROWS16 takes at most 16 rows (M=%d)", M);
  dim3 grid((N + RC - 1) / RC, (M > 8 && !rows16) ? (M + 7) / 8 : 1), block(nwaves * 64);
  if (M <= 8 || !rows16)
    gemm_rows_kernel<TX, T, 8><<<grid, block, 0, stream>>>((const TX*)x, ldx, (const T*)w, ldw, bias, (const TX*)res, ldr,
                                                          y, ldy, M, N, K, act, out_f32, norm_scale, norm_eps, zero_ptr,
                                                          zero_n);
  else
    gemm_rows_kernel<TX, T, 16><<<grid, block, 0, stream>>>((const TX*)x, ldx, (const T*)w, ldw, bias, (const TX*)res, ldr,
                                                           y, ldy, M, N, K, act, out_f32, norm_scale, norm_eps, zero_ptr,
                                                           zero_n);
  return vg_host::check_launch("vg_gemm_rows");
}

}  // namespace

extern "C" int vg_gemm_rows(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                            const void* residual, int64_t ldr, void* y, int64_t ldy, int M, int N, int K, int act,
                            int out_f32, const float* norm_scale, float norm_eps, int dtype, hipStream_t stream) {
  VG_REQUIRE(M >= 1 && M <= RMAXM && N >= 1 && K >= 8, "vg_gemm_rows: M=%d (1..%d) N=%d K=%d", M, RMAXM, N, K);
  VG_REQUIRE(K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0, "vg_gemm_rows: K, ldx, ldw must be multiples of 8");
  VG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, "vg_gemm_rows: x / w must be 16-byte aligned");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_gemm_rows: bad dtype %d", dtype);
  if (dtype == VG_BF16)
    return launch_rows<bf16_t, bf16_t>(x, ldx, w, ldw, bias, residual, ldr, y, ldy, M, N, K, act, out_f32, norm_scale,
                                       norm_eps, nullptr, 0, stream);
  return launch_rows<float, float>(x, ldx, w, ldw, bias, residual, ldr, y, ldy, M, N, K, act, out_f32, norm_scale, norm_eps,
                                   nullptr, 0, stream);
}

extern "C" int vg_gemm_rows_mixed(const float* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                                  const float* residual, int64_t ldr, void* y, int64_t ldy, int M, int N, int K, int act,
                                  int out_f32, const float* norm_scale, float norm_eps, float* zero_buf, int zero_n,
                                  int dtype, hipStream_t stream) {
  VG_REQUIRE(M >= 1 && M <= RMAXM && N >= 1 && K >= 8, "vg_gemm_rows_mixed: M=%d (1..%d) N=%d K=%d", M, RMAXM, N, K);
  VG_REQUIRE(K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0, "vg_gemm_rows_mixed: K, ldx, ldw must be multiples of 8");
  VG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, "vg_gemm_rows_mixed: x / w must be 16-byte aligned");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_gemm_rows_mixed: bad dtype %d", dtype);
  VG_REQUIRE(zero_n >= 0 && (zero_n == 0 || zero_buf != nullptr), "vg_gemm_rows_mixed: zero_n=%d without a buffer", zero_n);
  if (dtype == VG_BF16)
    return launch_rows<float, bf16_t>(x, ldx, w, ldw, bias, residual, ldr, y, ldy, M, N, K, act, out_f32, norm_scale,
                                      norm_eps, zero_buf, zero_n, stream);
  return launch_rows<float, float>(x, ldx, w, ldw, bias, residual, ldr, y, ldy, M, N, K, act, out_f32, norm_scale, norm_eps,
                                   zero_buf, zero_n, stream);
}

extern "C" int vg_gemm_rows_acc(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias,
                                const float* residual, int64_t ldr, float* y, int64_t ldy, int M, int N, int K, int splits,
                                float* zero_buf, int zero_n, hipStream_t stream) {
  VG_REQUIRE(M >= 1 && M <= RMAXM && N >= 1 && K >= 32 && K % 32 == 0, "vg_gemm_rows_acc: M=%d (1..%d) N=%d K=%d (a multiple of 32)", M, RMAXM, N, K);
  VG_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, "vg_gemm_rows_acc: x / w rows must be 16-byte aligned");
  VG_REQUIRE(splits >= 1 && splits <= 16 && y != nullptr, "vg_gemm_rows_acc: splits=%d (1..16)", splits);
  VG_REQUIRE(zero_n >= 0 && (zero_n == 0 || zero_buf != nullptr) && zero_buf != y, "vg_gemm_rows_acc: zero_n=%d without a buffer, or the buffer is y", zero_n);
  dim3 grid((N + 15) / 16, splits), block(MFW * 64);
  gemm_rows_mfma_kernel<bf16_t, true><<<grid, block, 0, stream>>>((const bf16_t*)x, ldx, (const bf16_t*)w, ldw, bias, (const void*)residual, ldr,
                                                                  (void*)y, ldy, M, N, K, VG_ACT_NONE, 1, nullptr, 0.f, zero_buf, zero_n);
  return vg_host::check_launch("vg_gemm_rows_acc");
}

extern "C" int vg_attn_decode_append(const void* qkv, void* kcache, void* vcache, void* out, const float* slopes,
                                     const int32_t* pos, int B, int Tmax, int H, int dtype, hipStream_t stream) {
  VG_REQUIRE(B > 0 && Tmax > 0 && H > 0, "vg_attn_decode_append: empty problem");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_attn_decode_append: bad dtype %d", dtype);
  dim3 grid(H, B);
  if (dtype == VG_BF16)
    attn_decode_append_kernel<bf16_t><<<grid, dim3(256), 0, stream>>>((const bf16_t*)qkv, (bf16_t*)kcache, (bf16_t*)vcache,
                                                                    (bf16_t*)out, slopes, pos, Tmax, H);
  else
    attn_decode_append_kernel<float><<<grid, dim3(256), 0, stream>>>((const float*)qkv, (float*)kcache, (float*)vcache,
                                                                   (float*)out, slopes, pos, Tmax, H);
  return vg_host::check_launch("vg_attn_decode_append");
}

extern "C" int vg_attn_layer_decode(const float* x, const float* norm_scale, float norm_eps, const void* wqkv,
                                    const float* bqkv, const void* wo, const float* bo, void* kcache, void* vcache,
                                    const float* slopes, const int32_t* pos, float* x1, float* zero_buf, int B, int Tmax,
                                    int H, int dtype, hipStream_t stream) {
  VG_REQUIRE(B > 0 && Tmax > 0 && H > 0 && H * 64 <= 1024 && H % 4 == 0,
             "vg_attn_layer_decode: B=%d Tmax=%d H=%d (model width 64 H: a multiple of 256, at most 1024)", B, Tmax, H);
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_attn_layer_decode: bad dtype %d", dtype);
  VG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)norm_scale % 16) == 0 && ((uintptr_t)wqkv % 16) == 0 &&
                 ((uintptr_t)wo % 16) == 0 && ((uintptr_t)kcache % 16) == 0 && ((uintptr_t)vcache % 16) == 0,
             "vg_attn_layer_decode: operands must be 16-byte aligned");
  dim3 grid(H, B);
  if (dtype == VG_BF16)
    attn_layer_decode_kernel<bf16_t, 16><<<grid, dim3(1024), 0, stream>>>(x, norm_scale, norm_eps, (const bf16_t*)wqkv, bqkv,
                                                                          (const bf16_t*)wo, bo, (bf16_t*)kcache,
                                                                          (bf16_t*)vcache, slopes, pos, Tmax, H, x1, zero_buf);
  else
    attn_layer_decode_kernel<float, 16><<<grid, dim3(1024), 0, stream>>>(x, norm_scale, norm_eps, (const float*)wqkv, bqkv,
                                                                         (const float*)wo, bo, (float*)kcache, (float*)vcache,
                                                                         slopes, pos, Tmax, H, x1, zero_buf);
  return vg_host::check_launch("vg_attn_layer_decode");
}

extern "C" int vg_touch(const void* ptr, int64_t bytes, int blocks, hipStream_t stream) {
  VG_REQUIRE(ptr != nullptr && bytes >= 16 && blocks >= 1 && ((uintptr_t)ptr % 16) == 0, "vg_touch: bytes=%lld blocks=%d", (long long)bytes, blocks);
  touch_kernel<<<dim3(blocks), dim3(256), 0, stream>>>((const f32x4*)ptr, bytes / 16, nullptr);
  return vg_host::check_launch("vg_touch");
}

extern "C" int vg_advance(int32_t* pos, int n, int by, hipStream_t stream) {
  VG_REQUIRE(n > 0, "vg_advance: empty");
  advance_kernel<<<dim3((n + 63) / 64), dim3(64), 0, stream>>>(pos, n, by);
  return vg_host::check_launch("vg_advance");
}

extern "C" int vg_embed_fuse(const float* frame, int ldf, const float* emb, int vocab, int E, const float* wf,
                             const float* bf, int latent, void* out, int B, int dtype, hipStream_t stream) {
  VG_REQUIRE(B > 0 && E > 0 && vocab > 0 && latent >= 0, "vg_embed_fuse: empty problem");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_embed_fuse: bad dtype %d", dtype);
  if (dtype == VG_BF16)
    embed_fuse_kernel<bf16_t><<<dim3(B), dim3(64), 0, stream>>>(frame, ldf, emb, vocab, E, wf, bf, latent, (bf16_t*)out);
  else
    embed_fuse_kernel<float><<<dim3(B), dim3(64), 0, stream>>>(frame, ldf, emb, vocab, E, wf, bf, latent, (float*)out);
  return vg_host::check_launch("vg_embed_fuse");
}

extern "C" int vg_sample_token(const float* logits, int V, float temperature, const float* uniform, float* frame,
                               int ldf, int32_t* pos, int B, hipStream_t stream) {
  VG_REQUIRE(B > 0 && V > 0 && temperature > 0.f, "vg_sample_token: B=%d V=%d temperature=%g", B, V, temperature);
  sample_token_kernel<<<dim3(B), dim3(64), 0, stream>>>(logits, V, 1.0f / temperature, uniform, frame, ldf, pos);
  return vg_host::check_launch("vg_sample_token");
}

extern "C" int vg_decode_noise(uint64_t seed, const int32_t* pos, const int32_t* epoch, float* normal, int n_normal,
                               float* uniform, int B, hipStream_t stream) {
  VG_REQUIRE(B > 0 && n_normal >= 0 && pos != nullptr, "vg_decode_noise: B=%d n_normal=%d", B, n_normal);
  decode_noise_kernel<<<dim3((B + 63) / 64), dim3(64), 0, stream>>>(seed, pos, epoch, normal, n_normal, uniform, B);
  return vg_host::check_launch("vg_decode_noise");
}

#ifdef VG_LAB_DSTAMP
extern "C" int vg_lab_set_dstamp(long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dstamp), &buf, sizeof(buf)); }
#endif
