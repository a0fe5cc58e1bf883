// Channels-last kernels for the bottleneck conv blocks of the posterior encoder and
// the diffusion UNet (reference modules/conv/layers.py:70-135,231-295 on (B, C, T);
// here on [M = B*T, C] rows so the 1x1 convolutions are plain GEMMs on the MFMA
// path and everything else is one HBM-bound row kernel):
//
//   dwnorm_fwd : v = depthwise_conv_k(x)[t] + conv_bias + time_emb[b]        (k taps along t,
//                y = gamma * (v - mean_c v) * rsqrt(var_c,unbiased v + eps) + beta   causal or look-ahead)
//   dwnorm_bwd_norm : du = d(loss)/dv   (+ per-block partial sums for gamma/beta grads)
//   dwnorm_bwd_conv : dx[t] = sum_k w[k] du[t - off_k]  (+ optional residual-gradient add),
//                     per-block partial sums of the tap gradients
// One wave64 per frame, 16-byte loads, fp32 statistics; channels per lane = C / 64.
// The per-frame norm is the reference's "InstanceNorm" (modules/norm.py:43-47:
// statistics over the channel axis, unbiased variance, fp32).
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

using namespace vg;

namespace {

constexpr int MAXV = 2;       // 16-byte vectors per lane (C <= 1024 bf16 / 512 f32)
constexpr int MAXTAPS = 8;

template <typename T> struct V8;
template <> struct V8<float> {
  static constexpr int N = 4;
  static VG_DEVICE void load(const float* p, float (&o)[8]) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
  static VG_DEVICE void store(float* p, const float (&o)[8]) {
    f32x4 v = {o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(p) = v;
  }
};
template <> struct V8<bf16_t> {
  static constexpr int N = 8;
  static VG_DEVICE void load(const bf16_t* p, float (&o)[8]) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  static VG_DEVICE void store(bf16_t* p, const float (&o)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)o[i];
    *reinterpret_cast<bf16x8*>(p) = v;
  }
};

struct DwArgs {
  int M, C, Tn, taps, shift;   // output frame t reads input frames t + k - shift, k = 0..taps-1
  float eps;
};

// v = conv(x)[row] + cbias + temb[b]   for this lane's channels (taps == 0: v = x[row])
template <typename T>
VG_DEVICE void conv_row(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ cbias,
                        const float* __restrict__ temb, const DwArgs& a, int row, int lane, float (&v)[MAXV][8]) {
  constexpr int N = V8<T>::N;
  const int nvec = a.C / N;
  const int b = row / a.Tn, t = row - b * a.Tn;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c >= nvec) continue;
    if (a.taps == 0) {
      V8<T>::load(x + (long)row * a.C + c * N, v[i]);
      continue;
    }
#pragma unroll
    for (int e = 0; e < N; ++e) v[i][e] = cbias ? cbias[c * N + e] : 0.f;
    if (temb) {
#pragma unroll
      for (int e = 0; e < N; ++e) v[i][e] += temb[(long)b * a.C + c * N + e];
    }
    for (int k = 0; k < a.taps; ++k) {
      const int ts = t + k - a.shift;
      if (ts < 0 || ts >= a.Tn) continue;
      float xv[8];
      V8<T>::load(x + ((long)b * a.Tn + ts) * a.C + c * N, xv);
#pragma unroll
      for (int e = 0; e < N; ++e) v[i][e] = fmaf(w[(c * N + e) * a.taps + k], xv[e], v[i][e]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwnorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ cbias,
                                                         const float* __restrict__ temb,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         DwArgs a) {
  constexpr int N = V8<T>::N;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const int nvec = a.C / N;
  float v[MAXV][8];
  conv_row<T>(x, w, cbias, temb, a, row, lane, v);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + 64 * i < nvec) {
#pragma unroll
      for (int e = 0; e < N; ++e) s += v[i][e];
    }
  const float mean = wave_sum(s) / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + 64 * i < nvec) {
#pragma unroll
      for (int e = 0; e < N; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)(a.C - 1) + a.eps);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c >= nvec) continue;
    float o[8];
#pragma unroll
    for (int e = 0; e < N; ++e) o[e] = fmaf(gamma[c * N + e], (v[i][e] - mean) * rstd, beta[c * N + e]);
    V8<T>::store(y + (long)row * a.C + c * N, o);
  }
}

// du = r * (g - mean(g)) - r^3 / (C - 1) * d * sum(g * d),  g = dy * gamma, d = v - mean
// part[block][0][c] = sum_rows dy * xhat ; part[block][1][c] = sum_rows dy
template <typename T>
__global__ __launch_bounds__(256) void dwnorm_bwd_norm_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ cbias,
                                                              const float* __restrict__ temb,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ mean_in,
                                                              const float* __restrict__ rstd_in, T* __restrict__ du,
                                                              float* __restrict__ part, DwArgs a) {
  constexpr int N = V8<T>::N;
  __shared__ float red[4][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = a.C / N;
  float sg[MAXV][8], sb[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) sg[i][e] = sb[i][e] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < a.M; row += gridDim.x * 4) {
    float v[MAXV][8], g[MAXV][8];
    conv_row<T>(x, w, cbias, temb, a, row, lane, v);
    const float mean = mean_in[row], r = rstd_in[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c >= nvec) continue;
      float dyv[8];
      V8<T>::load(dy + (long)row * a.C + c * N, dyv);
#pragma unroll
      for (int e = 0; e < N; ++e) {
        const float d = v[i][e] - mean;
        v[i][e] = d;
        g[i][e] = dyv[e] * gamma[c * N + e];
        s1 += g[i][e];
        s2 += g[i][e] * d;
        sg[i][e] += dyv[e] * d * r;
        sb[i][e] += dyv[e];
      }
    }
    s1 = wave_sum(s1) / (float)a.C;
    s2 = wave_sum(s2) * r * r * r / (float)(a.C - 1);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c >= nvec) continue;
      float o[8];
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = r * (g[i][e] - s1) - s2 * v[i][e];
      V8<T>::store(du + (long)row * a.C + c * N, o);
    }
  }
  float* scratch = &red[0][0];
#pragma unroll
  for (int which = 0; which < 2; ++which)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < N; ++e) scratch[(wave * 64 + lane) * N + e] = which ? sb[i][e] : sg[i][e];
      __syncthreads();
      const int c = lane + 64 * i;
      if (wave == 0 && c < nvec) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          float t = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) t += scratch[(ww * 64 + lane) * N + e];
          part[((long)blockIdx.x * 2 + which) * a.C + c * N + e] = t;
        }
      }
    }
}

// dx[t] = dx_add[t] + sum_k w[c][k] * du[t - (k - shift)] ; wpart[block][c][k] = sum_rows du[t] * x[t + k - shift]
template <typename T>
__global__ __launch_bounds__(256) void dwnorm_bwd_conv_kernel(const T* __restrict__ du, const T* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              const T* __restrict__ dx_add, T* __restrict__ dx,
                                                              float* __restrict__ wpart, DwArgs a) {
  constexpr int N = V8<T>::N;
  __shared__ float red[4][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = a.C / N;
  float gw[MAXV][MAXTAPS][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int k = 0; k < MAXTAPS; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) gw[i][k][e] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < a.M; row += gridDim.x * 4) {
    const int b = row / a.Tn, t = row - b * a.Tn;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c >= nvec) continue;
      float o[8], duv[8];
      if (dx_add) V8<T>::load(dx_add + (long)row * a.C + c * N, o);
      else {
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = 0.f;
      }
      V8<T>::load(du + (long)row * a.C + c * N, duv);
#pragma unroll
      for (int k = 0; k < MAXTAPS; ++k) {
        if (k >= a.taps) break;
        const int td = t - (k - a.shift);     // output frame whose tap k read input frame t
        if (td >= 0 && td < a.Tn) {
          float dv[8];
          V8<T>::load(du + ((long)b * a.Tn + td) * a.C + c * N, dv);
#pragma unroll
          for (int e = 0; e < N; ++e) o[e] = fmaf(w[(c * N + e) * a.taps + k], dv[e], o[e]);
        }
        const int ts = t + k - a.shift;       // input frame tap k of output frame t reads
        if (ts >= 0 && ts < a.Tn) {
          float xv[8];
          V8<T>::load(x + ((long)b * a.Tn + ts) * a.C + c * N, xv);
#pragma unroll
          for (int e = 0; e < N; ++e) gw[i][k][e] = fmaf(duv[e], xv[e], gw[i][k][e]);
        }
      }
      V8<T>::store(dx + (long)row * a.C + c * N, o);
    }
  }
  float* scratch = &red[0][0];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int k = 0; k < MAXTAPS; ++k) {
      if (k >= a.taps) break;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < N; ++e) scratch[(wave * 64 + lane) * N + e] = gw[i][k][e];
      __syncthreads();
      const int c = lane + 64 * i;
      if (wave == 0 && c < nvec) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          float t = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) t += scratch[(ww * 64 + lane) * N + e];
          wpart[(long)blockIdx.x * a.C * a.taps + (c * N + e) * a.taps + k] = t;
        }
      }
    }
}

int check_shape(const char* who, int M, int C, int T, int taps, int dtype) {
  const int n = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "%s: bad dtype %d", who, dtype);
  VG_REQUIRE(M > 0 && T > 0 && M % T == 0, "%s: M=%d must be a multiple of T=%d", who, M, T);
  VG_REQUIRE(C % n == 0 && C / n <= 64 * MAXV && C > 1, "%s: C=%d unsupported", who, C);
  VG_REQUIRE(taps >= 0 && taps <= MAXTAPS, "%s: taps=%d unsupported", who, taps);
  return 0;
}

}  // namespace

extern "C" int vg_dwnorm_blocks(int M) {
  const int b = (M + 3) / 4;
  return b < 256 ? b : 256;
}

extern "C" int vg_dwnorm_fwd(const void* x, const float* w, const float* cbias, const float* temb,
                             const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int C,
                             int T, int taps, int shift, float eps, int dtype, hipStream_t stream) {
  if (int e = check_shape("vg_dwnorm_fwd", M, C, T, taps, dtype)) return e;
  DwArgs a{M, C, T, taps, shift, eps};
  if (dtype == VG_BF16)
    dwnorm_fwd_kernel<bf16_t><<<dim3((M + 3) / 4), dim3(256), 0, stream>>>(
        (const bf16_t*)x, w, cbias, temb, gamma, beta, (bf16_t*)y, mean, rstd, a);
  else
    dwnorm_fwd_kernel<float><<<dim3((M + 3) / 4), dim3(256), 0, stream>>>((const float*)x, w, cbias, temb, gamma,
                                                                         beta, (float*)y, mean, rstd, a);
  return vg_host::check_launch("vg_dwnorm_fwd");
}

extern "C" int vg_dwnorm_bwd(const void* dy, const void* x, const float* w, const float* cbias, const float* temb,
                             const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du,
                             void* dx, float* norm_part, float* w_part, int M, int C, int T, int taps, int shift,
                             int dtype, hipStream_t stream) {
  if (int e = check_shape("vg_dwnorm_bwd", M, C, T, taps, dtype)) return e;
  DwArgs a{M, C, T, taps, shift, 0.f};
  const int nb = vg_dwnorm_blocks(M);
  if (dtype == VG_BF16) {
    dwnorm_bwd_norm_kernel<bf16_t><<<dim3(nb), dim3(256), 0, stream>>>(
        (const bf16_t*)dy, (const bf16_t*)x, w, cbias, temb, gamma, mean, rstd, (bf16_t*)du, norm_part, a);
    if (taps > 0)
      dwnorm_bwd_conv_kernel<bf16_t><<<dim3(nb), dim3(256), 0, stream>>>(
          (const bf16_t*)du, (const bf16_t*)x, w, (const bf16_t*)dx_add, (bf16_t*)dx, w_part, a);
  } else {
    dwnorm_bwd_norm_kernel<float><<<dim3(nb), dim3(256), 0, stream>>>(
        (const float*)dy, (const float*)x, w, cbias, temb, gamma, mean, rstd, (float*)du, norm_part, a);
    if (taps > 0)
      dwnorm_bwd_conv_kernel<float><<<dim3(nb), dim3(256), 0, stream>>>(
          (const float*)du, (const float*)x, w, (const float*)dx_add, (float*)dx, w_part, a);
  }
  return vg_host::check_launch("vg_dwnorm_bwd");
}
