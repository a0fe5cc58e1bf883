// Kernels of the autoregressive decode step: one new frame per sequence.
//
//  * vg_gemm_rows: y[M<=16][N] = epi(x[M][K] W[N][K]^T) -- the Linear layers of LVTR.step
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
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

using namespace vg;

namespace {

constexpr int RC = 8;           // output columns per block
constexpr int RCHUNK = 512;     // k range of one wave (64 lanes x 8 elements: 1 KiB of a bf16 weight row per load)
constexpr int RMAXW = 8;        // waves per block = min(8, ceil(K / 512)); longer K: a wave walks several ranges
constexpr int RMAXM = 16;

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

// Block = 8 output columns x all of K; wave w owns k in [512 w, 512 w + 512), lane l the 8 elements at
// 512 w + 8 l: every load is a fully coalesced 16-byte-per-lane row segment, the 8 weight rows of the
// block are requested back to back (8 KiB in flight per wave) and the x rows are read once per lane.
// Each lane then holds MM x 8 partial dot products; a recursive-halving exchange (xor 32, 16, .. 1; 63
// shuffles for 64 values instead of 64 full reductions) leaves lane l with the wave-wide sum of value
// index l (= row l / 8, column l % 8); the waves' sums meet in LDS and wave 0 applies the epilogue.
template <typename T, int MM>
__global__ __launch_bounds__(RMAXW * 64) void gemm_rows_kernel(const T* __restrict__ x, long ldx,
                                                               const T* __restrict__ w, long ldw,
                                                               const float* __restrict__ bias,
                                                               const T* __restrict__ residual, long ldr,
                                                               void* __restrict__ y, long ldy, int M, int N, int K,
                                                               int act, int out_f32,
                                                               const float* __restrict__ norm_scale, float norm_eps) {
  constexpr int V = MM * RC;            // values per lane before the exchange
  constexpr int PER = V / 64;           // values per lane after it (1 for MM = 8, 2 for MM = 16)
  __shared__ float red[RMAXW][V];
  __shared__ float ssq[RMAXW][MM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int n0 = blockIdx.x * RC;
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
      for (int m = 0; m < MM; ++m) Ld8<T>::get(x + (long)min(m, M - 1) * ldx + k0, xv[m]);
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
          Ld8<T>::get(x + (long)m * ldx + k0, xv);
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
        if (residual) r += to_f32<T>(residual[(long)m * ldr + n]);
        if (out_f32) reinterpret_cast<float*>(y)[(long)m * ldy + n] = r;
        else reinterpret_cast<T*>(y)[(long)m * ldy + n] = from_f32<T>(r);
      }
    }
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

__global__ void advance_kernel(int* __restrict__ pos, int n, int by) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[i] += by;
}

template <typename T>
int launch_rows(const void* x, long ldx, const void* w, long ldw, const float* bias, const void* res, long ldr, void* y,
                long ldy, int M, int N, int K, int act, int out_f32, const float* norm_scale, float norm_eps,
                hipStream_t stream) {
  int nwaves = (K + RCHUNK - 1) / RCHUNK;
  if (nwaves > RMAXW) nwaves = RMAXW;
  dim3 grid((N + RC - 1) / RC), block(nwaves * 64);
  if (M <= 8)
    gemm_rows_kernel<T, 8><<<grid, block, 0, stream>>>((const T*)x, ldx, (const T*)w, ldw, bias, (const T*)res, ldr, y,
                                                      ldy, M, N, K, act, out_f32, norm_scale, norm_eps);
  else
    gemm_rows_kernel<T, 16><<<grid, block, 0, stream>>>((const T*)x, ldx, (const T*)w, ldw, bias, (const T*)res, ldr, y,
                                                       ldy, M, N, K, act, out_f32, norm_scale, norm_eps);
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
    return launch_rows<bf16_t>(x, ldx, w, ldw, bias, residual, ldr, y, ldy, M, N, K, act, out_f32, norm_scale, norm_eps,
                               stream);
  return launch_rows<float>(x, ldx, w, ldw, bias, residual, ldr, y, ldy, M, N, K, act, out_f32, norm_scale, norm_eps,
                            stream);
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
