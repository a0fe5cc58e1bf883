// Conditional affine-coupling flow on the 4-d latent as ONE row kernel per direction
// (reference modules/flow/layers.py:15-98 LinearCoupling, :199-245 the stack; vae-gslm.yaml:
// 4 layers, hidden 64, LayerNorm, FiLM conditioning, erf-GELU, scale_range).
//
// The stack is row-local: every frame pushes its 4 numbers through L tiny MLPs (2 -> 64 -> 4).
// As stock tensor ops that is ~25 launches per layer forward and ~50 backward on [M, <=64] fp32
// tensors, all launch-latency bound.  Here one wave owns one frame and lane j owns hidden unit j:
// Linear(2->64) is two FMAs per lane, LayerNorm and Linear(64->4) are wave reductions, the FiLM
// scale/shift row ([M, L*128] fp32, produced by the MFMA GEMM) is read coalesced.  fp32 throughout
// (the reference keeps the flow in fp32 as well), exact erf GELU.
//
// Layer l (flip = true for every layer, layers.py:233):
//   keep = x[2:4], move = x[0:2]
//   a = W1 keep + b1 ; n = LN(a) * g + be ; f = fw * n + fb ; h = gelu(f) ; o = W2 h + b2
//   s = sigmoid(o[2:4]) * (hi - lo) + lo          (log-scale = log s, layers.py:52-55)
//   x' = (keep, o[0:2] + move * s) ;  logdet += sum(log s)   (masked frames contribute 0)
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

using namespace vg;

namespace {

constexpr int FH = 64;          // hidden units = lanes
constexpr int FMAXL = 8;        // kernels are instantiated for up to 4 and up to 8 layers (register budget)
constexpr int FPL = 580;        // packed parameters per layer: W1[64][2] b1[64] g[64] be[64] W2[4][64] b2[4]
constexpr int O_W1 = 0, O_B1 = 128, O_G = 192, O_BE = 256, O_W2 = 320, O_B2 = 576;

struct LaneParams {
  float w1a, w1b, b1, g, be, w2[4];
};

VG_DEVICE LaneParams load_lane_params(const float* __restrict__ P, int lane) {
  LaneParams q;
  q.w1a = P[O_W1 + 2 * lane];
  q.w1b = P[O_W1 + 2 * lane + 1];
  q.b1 = P[O_B1 + lane];
  q.g = P[O_G + lane];
  q.be = P[O_BE + lane];
#pragma unroll
  for (int o = 0; o < 4; ++o) q.w2[o] = P[O_W2 + o * FH + lane];
  return q;
}

// everything of one layer's forward that the backward needs again
struct LayerFwd {
  float xhat, rstd, n, f, h, o[4], s[2], sg[2];
};

VG_DEVICE LayerFwd layer_forward(const LaneParams& q, const float* __restrict__ b2, float keep0, float keep1,
                                 float fw, float fb, float eps, float hi, float lo) {
  LayerFwd r;
  const float a = fmaf(q.w1a, keep0, fmaf(q.w1b, keep1, q.b1));
  const float mean = wave_sum(a) * (1.0f / FH);
  const float c = a - mean;
  const float var = wave_sum(c * c) * (1.0f / FH);
  r.rstd = rsqrtf(var + eps);
  r.xhat = c * r.rstd;
  r.n = fmaf(r.xhat, q.g, q.be);
  r.f = fmaf(fw, r.n, fb);
  r.h = gelu_erf(r.f);
#pragma unroll
  for (int o = 0; o < 4; ++o) r.o[o] = wave_sum(q.w2[o] * r.h) + b2[o];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    r.sg[i] = 1.0f / (1.0f + expf(-r.o[2 + i]));
    r.s[i] = fmaf(r.sg[i], hi - lo, lo);
  }
  return r;
}

// ------------------------------------------------------------------ forward
template <int LMAX>
__global__ __launch_bounds__(256) void flow_fwd_kernel(const float* __restrict__ z, const float* __restrict__ wb,
                                                       long ldw, const float* __restrict__ params, int L,
                                                       float* __restrict__ u, float* __restrict__ logdet,
                                                       float* __restrict__ states, int M, float eps, float hi, float lo,
                                                       const int* __restrict__ lengths, int Tn) {
  const int lane = threadIdx.x & 63;
  const int stride = gridDim.x * 4;
  LaneParams q[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l)
    if (l < L) q[l] = load_lane_params(params + l * FPL, lane);
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += stride) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(z + 4L * m);
    float x[4] = {x0[0], x0[1], x0[2], x0[3]};
    float ld = 0.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      if (l < L) {
        if (states && lane < 4) states[((long)m * L + l) * 4 + lane] = x[lane];
        const float fw = wb[(long)m * ldw + l * 2 * FH + lane];
        const float fb = wb[(long)m * ldw + l * 2 * FH + FH + lane];
        const LayerFwd r = layer_forward(q[l], params + l * FPL + O_B2, x[2], x[3], fw, fb, eps, hi, lo);
        const float n0 = fmaf(x[0], r.s[0], r.o[0]), n1 = fmaf(x[1], r.s[1], r.o[1]);
        x[0] = x[2];
        x[1] = x[3];
        x[2] = n0;
        x[3] = n1;
        ld += logf(r.s[0]) + logf(r.s[1]);
      }
    }
    if (lane == 0) {
      *reinterpret_cast<f32x4*>(u + 4L * m) = f32x4{x[0], x[1], x[2], x[3]};
      logdet[m] = row_valid(lengths, Tn, m) ? ld : 0.f;
    }
  }
}

// ------------------------------------------------------------------ reverse (sampling, layers.py:76-81,241-245)
template <int LMAX>
__global__ __launch_bounds__(256) void flow_rev_kernel(const float* __restrict__ u, const float* __restrict__ wb,
                                                       long ldw, const float* __restrict__ params, int L,
                                                       float* __restrict__ z, long ldz, int M, float eps, float hi,
                                                       float lo, const float* __restrict__ mu_ls, long ldm,
                                                       float temperature) {
  const int lane = threadIdx.x & 63;
  const int stride = gridDim.x * 4;
  LaneParams q[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l)
    if (l < L) q[l] = load_lane_params(params + l * FPL, lane);
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += stride) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(u + 4L * m);
    float x[4] = {x0[0], x0[1], x0[2], x0[3]};
    if (mu_ls) {      // u holds unit noise: draw from the prior head first (linear/layers.py:110-128)
#pragma unroll
      for (int d = 0; d < 4; ++d) x[d] = fmaf(x[d] * temperature, expf(mu_ls[m * ldm + 4 + d]), mu_ls[m * ldm + d]);
    }
#pragma unroll
    for (int l = LMAX - 1; l >= 0; --l) {
      if (l < L) {
        // x = (keep, moved): invert moved = shift + orig * s
        const float fw = wb[(long)m * ldw + l * 2 * FH + lane];
        const float fb = wb[(long)m * ldw + l * 2 * FH + FH + lane];
        const LayerFwd r = layer_forward(q[l], params + l * FPL + O_B2, x[0], x[1], fw, fb, eps, hi, lo);
        const float o0 = (x[2] - r.o[0]) / r.s[0], o1 = (x[3] - r.o[1]) / r.s[1];
        x[2] = x[0];
        x[3] = x[1];
        x[0] = o0;
        x[1] = o1;
      }
    }
    if (lane < 4) z[m * ldz + lane] = x[lane];
  }
}

// ------------------------------------------------------------------ backward
// du [M][4], dlogdet [M] (gradient of the per-frame log-det sum; ignored on padded frames);
// outputs dz [M][4], dwb [M][ldw] (FiLM scale/shift gradients, every column of the L*128 block written),
// dparams_partial [gridDim.x][L*FPL] (per-block sums; reduce with vg_colsum).
template <int LMAX>
__global__ __launch_bounds__(256) void flow_bwd_kernel(const float* __restrict__ states, const float* __restrict__ wb,
                                                       long ldw, const float* __restrict__ params, int L,
                                                       const float* __restrict__ du, const float* __restrict__ dlogdet,
                                                       float* __restrict__ dz, float* __restrict__ dwb,
                                                       float* __restrict__ dpart, int M, float eps, float hi, float lo,
                                                       const int* __restrict__ lengths, int Tn) {
  __shared__ float red[4][FH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stride = gridDim.x * 4;
  LaneParams q[LMAX];
  float gw1a[LMAX], gw1b[LMAX], gb1[LMAX], gg[LMAX], gbe[LMAX], gw2[LMAX][4], gb2[LMAX][4];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) {
    if (l < L) q[l] = load_lane_params(params + l * FPL, lane);
    gw1a[l] = gw1b[l] = gb1[l] = gg[l] = gbe[l] = 0.f;
#pragma unroll
    for (int o = 0; o < 4; ++o) gw2[l][o] = gb2[l][o] = 0.f;
  }
  for (int m = blockIdx.x * 4 + wave; m < M; m += stride) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(du + 4L * m);
    float dx[4] = {g0[0], g0[1], g0[2], g0[3]};
    const float dld = row_valid(lengths, Tn, m) ? dlogdet[m] : 0.f;
#pragma unroll
    for (int l = LMAX - 1; l >= 0; --l) {
      if (l < L) {
        const f32x4 xin = *reinterpret_cast<const f32x4*>(states + ((long)m * L + l) * 4);   // (move, keep)
        const float fw = wb[(long)m * ldw + l * 2 * FH + lane];
        const float fb = wb[(long)m * ldw + l * 2 * FH + FH + lane];
        const LayerFwd r = layer_forward(q[l], params + l * FPL + O_B2, xin[2], xin[3], fw, fb, eps, hi, lo);
        // output = (keep0, keep1, o0 + move0 s0, o1 + move1 s1)
        float dout[4];
        float dmove[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float dn = dx[2 + i];
          dout[i] = dn;                                               // d shift
          dmove[i] = dn * r.s[i];
          const float ds = fmaf(dn, xin[i], dld / r.s[i]);            // new = .. + move s ; logdet += log s
          dout[2 + i] = ds * (hi - lo) * r.sg[i] * (1.0f - r.sg[i]);
        }
        float dh = 0.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          dh = fmaf(q[l].w2[o], dout[o], dh);
          gw2[l][o] = fmaf(dout[o], r.h, gw2[l][o]);
          gb2[l][o] += dout[o];
        }
        const float df = dh * gelu_erf_grad(r.f);
        dwb[(long)m * ldw + l * 2 * FH + lane] = df * r.n;
        dwb[(long)m * ldw + l * 2 * FH + FH + lane] = df;
        const float dn_ = df * fw;
        gg[l] = fmaf(dn_, r.xhat, gg[l]);
        gbe[l] += dn_;
        const float dxh = dn_ * q[l].g;
        const float m1 = wave_sum(dxh) * (1.0f / FH);
        const float m2 = wave_sum(dxh * r.xhat) * (1.0f / FH);
        const float da = r.rstd * (dxh - m1 - r.xhat * m2);
        gw1a[l] = fmaf(da, xin[2], gw1a[l]);
        gw1b[l] = fmaf(da, xin[3], gw1b[l]);
        gb1[l] += da;
        const float dk0 = wave_sum(q[l].w1a * da), dk1 = wave_sum(q[l].w1b * da);
        const float k0 = dx[0] + dk0, k1 = dx[1] + dk1;
        dx[0] = dmove[0];
        dx[1] = dmove[1];
        dx[2] = k0;
        dx[3] = k1;
      }
    }
    if (lane == 0) *reinterpret_cast<f32x4*>(dz + 4L * m) = f32x4{dx[0], dx[1], dx[2], dx[3]};
  }
  // block-level sums of the parameter gradients -> dpart[block][l * FPL + ...]
  float* __restrict__ out = dpart + (long)blockIdx.x * L * FPL;
  auto block_sum = [&](float v) {
    __syncthreads();
    red[wave][lane] = v;
    __syncthreads();
    return red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
  };
#pragma unroll
  for (int l = 0; l < LMAX; ++l) {
    if (l < L) {
      float* __restrict__ o = out + l * FPL;
      float t;
      t = block_sum(gw1a[l]); if (wave == 0) o[O_W1 + 2 * lane] = t;
      t = block_sum(gw1b[l]); if (wave == 0) o[O_W1 + 2 * lane + 1] = t;
      t = block_sum(gb1[l]);  if (wave == 0) o[O_B1 + lane] = t;
      t = block_sum(gg[l]);   if (wave == 0) o[O_G + lane] = t;
      t = block_sum(gbe[l]);  if (wave == 0) o[O_BE + lane] = t;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t = block_sum(gw2[l][k]); if (wave == 0) o[O_W2 + k * FH + lane] = t;
        t = block_sum(gb2[l][k]); if (wave == 0 && lane == 0) o[O_B2 + k] = t;     // lane-uniform value
      }
    }
  }
}

int flow_blocks(int M) {
  const int b = (M + 3) / 4;
  return b < 512 ? b : 512;
}

}  // namespace

extern "C" int vg_flow_blocks(int M) { return flow_blocks(M); }

extern "C" int vg_flow_fwd(const float* z, const float* wb, int64_t ldw, const float* params, int L, float* u,
                           float* logdet_sum, float* states, int M, float eps, float hi, float lo,
                           const int32_t* lengths, int T, hipStream_t stream) {
  VG_REQUIRE(M > 0 && L > 0 && L <= FMAXL, "vg_flow_fwd: M=%d L=%d (L <= %d)", M, L, FMAXL);
  VG_REQUIRE(ldw >= 2L * FH * L, "vg_flow_fwd: FiLM row stride %ld < %d", (long)ldw, 2 * FH * L);
  VG_REQUIRE(((uintptr_t)z % 16) == 0 && ((uintptr_t)u % 16) == 0 && (states == nullptr || ((uintptr_t)states % 16) == 0),
             "vg_flow_fwd: unaligned");
  auto k = L <= 4 ? flow_fwd_kernel<4> : flow_fwd_kernel<8>;
  k<<<dim3(flow_blocks(M)), dim3(256), 0, stream>>>(z, wb, (long)ldw, params, L, u, logdet_sum, states, M, eps, hi, lo,
                                                   lengths, T > 0 ? T : 1);
  return vg_host::check_launch("vg_flow_fwd");
}

extern "C" int vg_flow_reverse(const float* u, const float* wb, int64_t ldw, const float* params, int L, float* z,
                               int64_t ldz, int M, float eps, float hi, float lo, const float* mu_ls, int64_t ld_mu_ls,
                               float temperature, hipStream_t stream) {
  VG_REQUIRE(M > 0 && L > 0 && L <= FMAXL, "vg_flow_reverse: M=%d L=%d (L <= %d)", M, L, FMAXL);
  VG_REQUIRE(ldw >= 2L * FH * L, "vg_flow_reverse: FiLM row stride %ld < %d", (long)ldw, 2 * FH * L);
  VG_REQUIRE(((uintptr_t)u % 16) == 0 && ldz >= 4, "vg_flow_reverse: u must be 16-byte aligned, ldz >= 4");
  auto k = L <= 4 ? flow_rev_kernel<4> : flow_rev_kernel<8>;
  k<<<dim3(flow_blocks(M)), dim3(256), 0, stream>>>(u, wb, (long)ldw, params, L, z, (long)ldz, M, eps, hi, lo, mu_ls,
                                                   (long)ld_mu_ls, temperature);
  return vg_host::check_launch("vg_flow_reverse");
}

extern "C" int vg_flow_bwd(const float* states, const float* wb, int64_t ldw, const float* params, int L,
                           const float* du, const float* dlogdet_sum, float* dz, float* dwb, float* dparams_partial,
                           int M, float eps, float hi, float lo, const int32_t* lengths, int T, hipStream_t stream) {
  VG_REQUIRE(M > 0 && L > 0 && L <= FMAXL, "vg_flow_bwd: M=%d L=%d (L <= %d)", M, L, FMAXL);
  VG_REQUIRE(ldw >= 2L * FH * L, "vg_flow_bwd: FiLM row stride %ld < %d", (long)ldw, 2 * FH * L);
  VG_REQUIRE(((uintptr_t)states % 16) == 0 && ((uintptr_t)du % 16) == 0 && ((uintptr_t)dz % 16) == 0,
             "vg_flow_bwd: unaligned");
  auto k = L <= 4 ? flow_bwd_kernel<4> : flow_bwd_kernel<8>;
  k<<<dim3(flow_blocks(M)), dim3(256), 0, stream>>>(states, wb, (long)ldw, params, L, du, dlogdet_sum, dz, dwb,
                                                   dparams_partial, M, eps, hi, lo, lengths, T > 0 ? T : 1);
  return vg_host::check_launch("vg_flow_bwd");
}
