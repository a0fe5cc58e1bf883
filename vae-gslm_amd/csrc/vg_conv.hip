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
// One wave64 per frame, 16-byte loads, fp32 statistics.  Every wave walks a strided set of
// frames with its channels' tap weights, bias and affine parameters held in registers
// (they are frame-invariant), so the inner loop is loads of x / dy rows and FMAs only.
// The per-frame norm is the reference's "InstanceNorm" (modules/norm.py:43-47:
// statistics over the channel axis, unbiased variance, fp32).
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

using namespace vg;

namespace {

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
  // the 16 bytes as loaded (kept packed while several rows are in flight) and their expansion
  static VG_DEVICE uint4 raw(const float* p) { return *reinterpret_cast<const uint4*>(p); }
  static VG_DEVICE void expand(const uint4& r, float (&o)[8]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(&r);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
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
  static VG_DEVICE uint4 raw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static VG_DEVICE void expand(const uint4& r, float (&o)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(&r);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
};

struct DwArgs {
  int M, C, Tn, taps, shift;   // output frame t reads input frames t + k - shift, k = 0..taps-1
  float eps;
  // packed rows (round 5; the run kernels only): sequence s = rows [cu[s], cu[s + 1]) for s < nseq <= 64 -- every
  // sequence with its own zero padding on both sides -- and the time-embedding row min(s, nbatch - 1); nullptr: M / Tn
  // sequences of Tn rows each
  const int* cu = nullptr;
  int nseq = 0, nbatch = 0;
  // round 6 (the run kernels only; vg_dwnorm_fwd_cat / vg_dwnorm_bwd_ld): the forward's output rows / the backward's
  // incoming-gradient rows are `ldy` elements apart (0: C), and the forward appends the block's condition channels to
  // every output row -- y[row][C + j] = cond[row][j] for j < cond_cols, 0 up to C + 64 -- so that the 1x1 convolution
  // over [norm(dwconv(x)) ; cond] (modules/conv/layers.py:112-113) is ONE product over K = C + 64 instead of a K = C
  // product with a pre-activation operand that a K = 32 product wrote first
  long ldy = 0;
  const bf16_t* cond = nullptr;
  long ldcond = 0;
  int cond_cols = 0;
};

// The frame-invariant parameters (taps x C weights, conv bias, norm affine) reach the lanes through LDS: one
// coalesced copy per block instead of ~80 strided scalar loads per lane (which cost more L2 traffic than the
// activations themselves once a launch has a thousand blocks).
constexpr int MAXC = 1024;
struct ParamLds {
  float w[MAXC * MAXTAPS];
  float cb[MAXC], gamma[MAXC], beta[MAXC];
};
VG_DEVICE void stage_params(ParamLds& P, const float* __restrict__ w, const float* __restrict__ cbias,
                            const float* __restrict__ gamma, const float* __restrict__ beta, const DwArgs& a) {
  const int tid = threadIdx.x, nt = blockDim.x;
  // batches of independent loads first, LDS stores afterwards: a load -> store loop would pay one L2 round
  // trip per iteration (14 of them for 512 channels x 7 taps), i.e. ~10 us of set-up per block
  constexpr int PF = 8;
  const int nw = a.C * a.taps;
  for (int base = 0; base < nw; base += nt * PF) {
    float r[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = base + j * nt + tid;
      r[j] = i < nw ? w[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = base + j * nt + tid;
      if (i < nw) P.w[(i % a.taps) * a.C + i / a.taps] = r[j];   // [tap][C]: lanes read 8 channels contiguously
    }
  }
  for (int base = 0; base < a.C; base += nt * 2) {
    float rc[2], rg[2], rb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = base + j * nt + tid;
      const bool in = i < a.C;
      rc[j] = (in && cbias && a.taps > 0) ? cbias[i] : 0.f;
      rg[j] = (in && gamma) ? gamma[i] : 1.f;
      rb[j] = (in && beta) ? beta[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = base + j * nt + tid;
      if (i < a.C) {
        P.cb[i] = rc[j];
        P.gamma[i] = rg[j];
        P.beta[i] = rb[j];
      }
    }
  }
  __syncthreads();
}

// Frames are dealt to waves in runs of RUN consecutive frames (a block's 4 waves cover 4 * RUN consecutive
// frames before jumping ahead): the taps of neighbouring frames then hit the CU's vector L1 instead of
// going to L2 seven times per frame.
constexpr int RUN = 4;
struct RowWalk {
  int chunk, r, cstride, M;
  VG_DEVICE RowWalk(int M_)
      : chunk(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), r(0), cstride(gridDim.x * (blockDim.x >> 6)), M(M_) {}
  VG_DEVICE int row() const { return chunk * RUN + r; }
  VG_DEVICE bool valid() const { return row() < M; }
  VG_DEVICE void next() {
    if (++r == RUN) {
      r = 0;
      chunk += cstride;
    }
  }
};

// frame-invariant per-lane parameters (NV 16-byte channel vectors per lane)
template <typename T, int NV> struct LaneParams {
  float w[NV][MAXTAPS][V8<T>::N];
  float cb[NV][V8<T>::N];
  VG_DEVICE void load(const ParamLds& P, const DwArgs& a, int lane) {
    constexpr int N = V8<T>::N;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
#pragma unroll
      for (int e = 0; e < N; ++e) cb[i][e] = P.cb[c * N + e];
#pragma unroll
      for (int k = 0; k < MAXTAPS; ++k)
#pragma unroll
        for (int e = 0; e < N; ++e) w[i][k][e] = (k < a.taps) ? P.w[k * a.C + c * N + e] : 0.f;
    }
  }
};

// v = conv(x)[row] + cbias + temb[b]   for this lane's channels (taps == 0: v = x[row]), in two phases so that
// a kernel can request the rows of its NEXT frame before it reduces the current one:
//   conv_row_issue : all tap rows as packed 16-byte registers (clamped addresses: one memory latency per
//                    frame, not one per tap) + the time-embedding vector
//   conv_row_finish: expansion, 0/1 masks for taps outside the sequence, FMAs
template <typename T, int NV> struct RowRaw {
  uint4 x[NV][MAXTAPS];
  float te[NV][8];
};

template <typename T, int NV>
VG_DEVICE void conv_row_issue(const T* __restrict__ x, const float* __restrict__ temb, const DwArgs& a, int row,
                              int lane, RowRaw<T, NV>& r) {
  constexpr int N = V8<T>::N;
  const int b = row / a.Tn, t = row - b * a.Tn;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (a.taps == 0) {
      r.x[i][0] = V8<T>::raw(x + (long)row * a.C + c * N);
      continue;
    }
#pragma unroll
    for (int k = 0; k < MAXTAPS; ++k) {
      if (k < a.taps) {
        const int ts = min(max(t + k - a.shift, 0), a.Tn - 1);
        r.x[i][k] = V8<T>::raw(x + ((long)b * a.Tn + ts) * a.C + c * N);
      }
    }
    if (temb) {
#pragma unroll
      for (int e = 0; e < N; ++e) r.te[i][e] = temb[(long)b * a.C + c * N + e];
    }
  }
}

template <typename T, int NV>
VG_DEVICE void conv_row_finish(const RowRaw<T, NV>& r, const LaneParams<T, NV>& lp, bool has_temb, const DwArgs& a,
                               int row, float (&v)[NV][8]) {
  constexpr int N = V8<T>::N;
  const int b = row / a.Tn, t = row - b * a.Tn;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (a.taps == 0) {
      V8<T>::expand(r.x[i][0], v[i]);
      continue;
    }
#pragma unroll
    for (int e = 0; e < N; ++e) v[i][e] = lp.cb[i][e] + (has_temb ? r.te[i][e] : 0.f);
#pragma unroll
    for (int k = 0; k < MAXTAPS; ++k) {
      if (k < a.taps) {
        const int ts = t + k - a.shift;
        const float on = (ts >= 0 && ts < a.Tn) ? 1.f : 0.f;
        float xv[8];
        V8<T>::expand(r.x[i][k], xv);
#pragma unroll
        for (int e = 0; e < N; ++e) v[i][e] = fmaf(lp.w[i][k][e] * on, xv[e], v[i][e]);
      }
    }
  }
}

template <typename T, int NV>
VG_DEVICE void conv_row(const T* __restrict__ x, const LaneParams<T, NV>& lp, const float* __restrict__ temb,
                        const DwArgs& a, int row, int lane, float (&v)[NV][8]) {
  RowRaw<T, NV> r;
  conv_row_issue<T, NV>(x, temb, a, row, lane, r);
  conv_row_finish<T, NV>(r, lp, temb != nullptr, a, row, v);
}

template <typename T, int NV>
__global__ __launch_bounds__(512) void dwnorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ cbias,
                                                         const float* __restrict__ temb,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         DwArgs a) {
  constexpr int N = V8<T>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ ParamLds P;
  stage_params(P, w, cbias, gamma, beta, a);
  LaneParams<T, NV> lp;
  lp.load(P, a, lane);
  float gm[NV][N], bt[NV][N];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < N; ++e) {
      gm[i][e] = P.gamma[(lane + 64 * i) * N + e];
      bt[i][e] = P.beta[(lane + 64 * i) * N + e];
    }
  RowWalk walk(a.M);
  RowRaw<T, NV> cur, nxt;
  if (walk.valid()) conv_row_issue<T, NV>(x, temb, a, walk.row(), lane, cur);
  for (; walk.valid(); walk.next()) {
    const int row = walk.row();
    RowWalk ahead = walk;
    ahead.next();
    if (ahead.valid()) conv_row_issue<T, NV>(x, temb, a, ahead.row(), lane, nxt);   // next frame's rows in flight
    float v[NV][8];
    conv_row_finish<T, NV>(cur, lp, temb != nullptr, a, row, v);
    cur = nxt;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < N; ++e) s += v[i][e];
    const float mean = wave_sum(s) / (float)a.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < N; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)(a.C - 1) + a.eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float o[8];
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = fmaf(gm[i][e], (v[i][e] - mean) * rstd, bt[i][e]);
      V8<T>::store(y + (long)row * a.C + (lane + 64 * i) * N, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Round 3: run-based forward for the shape the model runs (bf16, one 16-byte channel vector per lane, 7 taps).
// A wave owns a RUN of 8 consecutive frames of one sequence instead of one frame at a time:
//   * the 8 + 6 input rows of the run are requested together (one memory round trip per run; the frame-at-a-time
//     kernel fetched 7 rows per frame, 6 of them again through L1 / L2, behind one dependent round trip per frame);
//   * every input row is expanded to fp32 once and added into the up to 7 frames it feeds (64 accumulators);
//   * the 2 x 8 wave reductions of a run are independent DPP chains that interleave, instead of two exposed
//     reductions per frame;
//   * the lane's frame-invariant parameters (56 tap weights, bias, affine) come straight from global memory with
//     16-byte loads: no LDS staging, no block barrier.
// Same arithmetic as dwnorm_fwd_kernel (fp32 statistics, two-pass unbiased variance).
// ---------------------------------------------------------------------------------------------------------
constexpr int RUNF = 8;

// Runs of RF consecutive frames of ONE sequence, numbered through the launch.  Uniform sequences: run r belongs to
// sequence r / ceil(Tn / RF).  Packed rows: lane s of every wave holds sequence s (its first row, its length and the
// number of runs before it, from one 64-wide scan at kernel entry); a run finds its sequence with one ballot over
// "all my runs come before r" and three v_readlane.
template <int RF> struct SegRuns {
  int len, base, excl, incl, rps;
  int nruns;
  VG_DEVICE void init(const DwArgs& a, int lane) {
    if (a.cu) {
      const bool in = lane < a.nseq;
      base = in ? a.cu[lane] : 0;
      len = in ? a.cu[lane + 1] - base : 0;
      const int n = (len + RF - 1) / RF;
      int sum = n;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(sum, d);
        if (lane >= d) sum += o;
      }
      incl = sum;
      excl = sum - n;
      nruns = __builtin_amdgcn_readlane(sum, 63);
      rps = 1;
    } else {
      rps = (a.Tn + RF - 1) / RF;
      nruns = (a.M / a.Tn) * rps;
      len = base = excl = incl = 0;
    }
  }
  // sequence b (time-embedding row bt), first frame t0 of the run, the sequence's length Tn and first row
  VG_DEVICE void locate(const DwArgs& a, int run, int& b, int& bt, int& t0, int& Tn, long& row_base) const {
    if (a.cu) {
      const int s = __builtin_popcountll(__ballot(incl <= run));
      b = s;
      bt = min(s, a.nbatch - 1);
      t0 = (run - __builtin_amdgcn_readlane(excl, s)) * RF;
      Tn = __builtin_amdgcn_readlane(len, s);
      row_base = __builtin_amdgcn_readlane(base, s);
    } else {
      b = run / rps;
      bt = b;
      t0 = (run - b * rps) * RF;
      Tn = a.Tn;
      row_base = (long)b * a.Tn;
    }
  }
};

VG_DEVICE void load8(const float* p, float (&o)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
  o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}

template <int TAPS>
struct RunParams {
  float w[8][TAPS];            // [channel of the lane][tap]
  VG_DEVICE void load(const float* __restrict__ wg, int lane) {
    // weights are [C][taps]: the lane's 8 channels x TAPS taps are 8 * TAPS consecutive floats
    float flat[8 * TAPS];
    const float* p = wg + (long)lane * 8 * TAPS;
#pragma unroll
    for (int i = 0; i < 2 * TAPS; ++i) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(p + 4 * i);
      flat[4 * i] = q[0]; flat[4 * i + 1] = q[1]; flat[4 * i + 2] = q[2]; flat[4 * i + 3] = q[3];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int k = 0; k < TAPS; ++k) w[e][k] = flat[e * TAPS + k];
  }
};

template <int TAPS>
__global__ __launch_bounds__(256) void dwnorm_fwd_run_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ cbias,
                                                             const float* __restrict__ temb,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             DwArgs a) {
  constexpr int NR = RUNF + TAPS - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SegRuns<RUNF> sr;
  sr.init(a, lane);
  RunParams<TAPS> P;
  P.load(w, lane);
  float cb[8], gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cb[e] = 0.f; gm[e] = 1.f; bt[e] = 0.f; }
  if (cbias) load8(cbias + lane * 8, cb);
  if (gamma) load8(gamma + lane * 8, gm);
  if (beta) load8(beta + lane * 8, bt);
  const float inv_c = 1.0f / (float)a.C, inv_c1 = 1.0f / (float)(a.C - 1);
  const long ldy = a.ldy > 0 ? a.ldy : a.C;
  for (int run = blockIdx.x * (blockDim.x >> 6) + wave; run < sr.nruns; run += gridDim.x * (blockDim.x >> 6)) {
    int b, bte, t0, Tn;
    long rbase;
    sr.locate(a, run, b, bte, t0, Tn, rbase);
    const bf16_t* xb = x + rbase * a.C + lane * 8;
    uint4 raw[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int tc = min(max(t0 - a.shift + j, 0), Tn - 1);
      raw[j] = *reinterpret_cast<const uint4*>(xb + (long)tc * a.C);
    }
    float v[RUNF][8];
    {
      float te[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (temb) load8(temb + (long)bte * a.C + lane * 8, te);
#pragma unroll
      for (int f = 0; f < RUNF; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[f][e] = cb[e] + te[e];
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int ti = t0 - a.shift + j;
      const float on = (ti >= 0 && ti < Tn) ? 1.f : 0.f;         // per-sequence zero padding
      float xv[8];
      V8<bf16_t>::expand(raw[j], xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] *= on;
#pragma unroll
      for (int f = 0; f < RUNF; ++f) {
        const int k = j - f;                                        // tap of frame f that reads window row j
        if (k >= 0 && k < TAPS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[f][e] = fmaf(P.w[e][k], xv[e], v[f][e]);
        }
      }
    }
    float mean[RUNF], rstd[RUNF];
#pragma unroll
    for (int f = 0; f < RUNF; ++f) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[f][e];
      mean[f] = wave_sum(s) * inv_c;
    }
#pragma unroll
    for (int f = 0; f < RUNF; ++f) {
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[f][e] - mean[f]; q = fmaf(d, d, q); }
      rstd[f] = rsqrtf(wave_sum(q) * inv_c1 + a.eps);
    }
    const long row0 = rbase + t0;
#pragma unroll
    for (int f = 0; f < RUNF; ++f) {
      if (t0 + f < Tn) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(gm[e], (v[f][e] - mean[f]) * rstd[f], bt[e]);
        V8<bf16_t>::store(y + (row0 + f) * ldy + lane * 8, o);
        if (a.cond != nullptr && lane < 8) {        // the row's 64-column tail: condition channels, then zeros
          uint4 c = make_uint4(0u, 0u, 0u, 0u);
          if (lane * 8 < a.cond_cols) c = *reinterpret_cast<const uint4*>(a.cond + (row0 + f) * a.ldcond + lane * 8);
          *reinterpret_cast<uint4*>(y + (row0 + f) * ldy + a.C + lane * 8) = c;
        }
        if (lane == 0) { mean_out[row0 + f] = mean[f]; rstd_out[row0 + f] = rstd[f]; }
      }
    }
  }
}

// The arithmetic of one frame's du = d loss / d v, shared by the run kernel and the one-launch kernel below so that both
// round identically: floating-point contraction is pinned off here (only the written fmaf is fused) -- left to the
// compiler, `a1 += dy * gamma` and `r * (dy * gamma - s1) - s2 * d` contract differently in the two kernels and one
// element in ~10^5 lands on the other side of a bf16 rounding boundary.
//   du_sums : d = v - mean (returned in v), a1 = sum g, a2 = sum g d with g = dy gamma   (this lane's 8 channels)
//   du_value: du = r (g - s1) - s2 d with s1 = mean_c g, s2 = r^3 / (C - 1) sum_c g d
VG_DEVICE void du_sums(const float (&dyv)[8], const float (&gm)[8], float mean, float (&v)[8], float& a1, float& a2) {
#pragma clang fp contract(off)
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float d = v[e] - mean;
    const float g = dyv[e] * gm[e];
    v[e] = d;
    s = s + g;
    q = fmaf(g, d, q);
  }
  a1 = s;
  a2 = q;
}
VG_DEVICE void du_value(const float (&dyv)[8], const float (&gm)[8], const float (&d)[8], float r, float s1, float s2, float (&o)[8]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float g = dyv[e] * gm[e];
    o[e] = r * (g - s1) - s2 * d[e];
  }
}

// run-based backward through the norm (same shapes as dwnorm_fwd_run_kernel; RF frames per run -- 4: the extra dy rows and
// partial sums leave no registers for 8 at two waves per SIMD): recomputes v for the frames of a run from one
// (RF + 6)-row window, the two reductions of the frames interleave; the per-lane gamma / beta partial sums of
// the block's four waves meet in LDS once.  part[block][0][c] = sum_rows dy * xhat ; part[block][1][c] = sum_rows dy
template <int TAPS, int RF>
__global__ __launch_bounds__(256) void dwnorm_bwd_norm_run_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ cbias,
                                                                  const float* __restrict__ temb,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean_in,
                                                                  const float* __restrict__ rstd_in, bf16_t* __restrict__ du,
                                                                  float* __restrict__ part, DwArgs a) {
  constexpr int NR = RF + TAPS - 1;
  __shared__ float red[4][2][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SegRuns<RF> sr;
  sr.init(a, lane);
  RunParams<TAPS> P;
  P.load(w, lane);
  float cb[8], gm[8], sg[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cb[e] = 0.f; gm[e] = 1.f; sg[e] = 0.f; sb[e] = 0.f; }
  if (cbias) load8(cbias + lane * 8, cb);
  if (gamma) load8(gamma + lane * 8, gm);
  const float inv_c = 1.0f / (float)a.C, inv_c1 = 1.0f / (float)(a.C - 1);
  const long ldy = a.ldy > 0 ? a.ldy : a.C;
  for (int run = blockIdx.x * 4 + wave; run < sr.nruns; run += gridDim.x * 4) {
    int b, bte, t0, Tn;
    long rbase;
    sr.locate(a, run, b, bte, t0, Tn, rbase);
    const long row0 = rbase + t0;
    const bf16_t* xb = x + rbase * a.C + lane * 8;
    uint4 raw[NR], rdy[RF];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int tc = min(max(t0 - a.shift + j, 0), Tn - 1);
      raw[j] = *reinterpret_cast<const uint4*>(xb + (long)tc * a.C);
    }
#pragma unroll
    for (int f = 0; f < RF; ++f)
      rdy[f] = *reinterpret_cast<const uint4*>(dy + (rbase + min(t0 + f, Tn - 1)) * ldy + lane * 8);
    float v[RF][8];
    {
      float te[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (temb) load8(temb + (long)bte * a.C + lane * 8, te);
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[f][e] = cb[e] + te[e];
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int ti = t0 - a.shift + j;
      const float on = (ti >= 0 && ti < Tn) ? 1.f : 0.f;
      float xv[8];
      V8<bf16_t>::expand(raw[j], xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] *= on;
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        const int k = j - f;
        if (k >= 0 && k < TAPS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[f][e] = fmaf(P.w[e][k], xv[e], v[f][e]);
        }
      }
    }
    float s1[RF], s2[RF], rr[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const int tf = min(t0 + f, Tn - 1);
      const float mean = mean_in[rbase + tf], r = rstd_in[rbase + tf];
      const float on = t0 + f < Tn ? 1.f : 0.f;                 // frames past the end of the sequence contribute nothing
      rr[f] = r;
      float dyv[8];
      V8<bf16_t>::expand(rdy[f], dyv);
      du_sums(dyv, gm, mean, v[f], s1[f], s2[f]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dd = dyv[e] * on;
        sg[e] = fmaf(dd * v[f][e], r, sg[e]);
        sb[e] += dd;
      }
    }
#pragma unroll
    for (int f = 0; f < RF; ++f) s1[f] = wave_sum(s1[f]) * inv_c;
#pragma unroll
    for (int f = 0; f < RF; ++f) s2[f] = wave_sum(s2[f]) * rr[f] * rr[f] * rr[f] * inv_c1;
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      if (t0 + f < Tn) {
        float dyv[8], o[8];
        V8<bf16_t>::expand(rdy[f], dyv);
        du_value(dyv, gm, v[f], rr[f], s1[f], s2[f], o);
        V8<bf16_t>::store(du + (row0 + f) * a.C + lane * 8, o);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[wave][0][lane * 8 + e] = sg[e];
    red[wave][1][lane * 8 + e] = sb[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 512; i += 256) {
    const int which = i >> 9, c = i & 511;
    part[((long)blockIdx.x * 2 + which) * a.C + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// run-based backward through the depthwise convolution: dx[t] = dx_add[t] + sum_k w[c][k] du[t - (k - shift)] from one
// 14-row window of du, then the tap gradients wpart[block][c][k] = sum_rows du[t] x[t + k - shift] from one 14-row
// window of x and the run's own 8 du rows.
// (one wave per SIMD: 256 VGPRs + 22 AGPRs.  Forced to two -- 31 registers spilled -- the kernel is slower, 24.4 -> 29.3 us at
// 16000 x 512, tools/lab/dw_bwd_probe.py, round 5.)
#ifndef VG_DW_BWDCONV_OCC
#define VG_DW_BWDCONV_OCC 1
#endif
template <int TAPS, int RF>
__global__ __launch_bounds__(256, VG_DW_BWDCONV_OCC) void dwnorm_bwd_conv_run_kernel(const bf16_t* __restrict__ du, const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  const bf16_t* __restrict__ dx_add, bf16_t* __restrict__ dx,
                                                                  float* __restrict__ wpart, DwArgs a) {
  constexpr int NR = RF + TAPS - 1;
  __shared__ float red[4][8 * TAPS * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SegRuns<RF> sr;
  sr.init(a, lane);
  float gw[8][TAPS];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int k = 0; k < TAPS; ++k) gw[e][k] = 0.f;
  for (int run = blockIdx.x * 4 + wave; run < sr.nruns; run += gridDim.x * 4) {
    int b, bte, t0, Tn;
    long rbase;
    sr.locate(a, run, b, bte, t0, Tn, rbase);
    const long row0 = rbase + t0;
    const bf16_t* dub = du + rbase * a.C + lane * 8;
    const bf16_t* xb = x + rbase * a.C + lane * 8;
    {   // ---- dx: window row j holds du frame t0 + shift - (TAPS - 1) + j; frame f, tap k reads row f + TAPS - 1 - k
      RunParams<TAPS> P;
      P.load(w, lane);
      uint4 raw[NR], radd[RF];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int tc = min(max(t0 + a.shift - (TAPS - 1) + j, 0), Tn - 1);
        raw[j] = *reinterpret_cast<const uint4*>(dub + (long)tc * a.C);
      }
      if (dx_add) {
#pragma unroll
        for (int f = 0; f < RF; ++f)
          radd[f] = *reinterpret_cast<const uint4*>(dx_add + (rbase + min(t0 + f, Tn - 1)) * a.C + lane * 8);
      }
      float o[RF][8];
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        if (dx_add) V8<bf16_t>::expand(radd[f], o[f]);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[f][e] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int td = t0 + a.shift - (TAPS - 1) + j;
        const float on = (td >= 0 && td < Tn) ? 1.f : 0.f;
        float dv[8];
        V8<bf16_t>::expand(raw[j], dv);
#pragma unroll
        for (int e = 0; e < 8; ++e) dv[e] *= on;
#pragma unroll
        for (int f = 0; f < RF; ++f) {
          const int k = f + TAPS - 1 - j;
          if (k >= 0 && k < TAPS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[f][e] = fmaf(P.w[e][k], dv[e], o[f][e]);
          }
        }
      }
#pragma unroll
      for (int f = 0; f < RF; ++f)
        if (t0 + f < Tn) V8<bf16_t>::store(dx + (row0 + f) * a.C + lane * 8, o[f]);
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the second phase's loads behind the first phase (registers: two waves per SIMD)
    {   // ---- tap gradients: window row j holds x frame t0 - shift + j; frame f, tap k reads row f + k
      uint4 raw[NR], rdu[RF];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int tc = min(max(t0 - a.shift + j, 0), Tn - 1);
        raw[j] = *reinterpret_cast<const uint4*>(xb + (long)tc * a.C);
      }
#pragma unroll
      for (int f = 0; f < RF; ++f) rdu[f] = *reinterpret_cast<const uint4*>(dub + (long)min(t0 + f, Tn - 1) * a.C);
      float duv[RF][8];
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        V8<bf16_t>::expand(rdu[f], duv[f]);
        const float on = t0 + f < Tn ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) duv[f][e] *= on;
      }
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int ts = t0 - a.shift + j;
        const float on = (ts >= 0 && ts < Tn) ? 1.f : 0.f;
        float xv[8];
        V8<bf16_t>::expand(raw[j], xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] *= on;
#pragma unroll
        for (int f = 0; f < RF; ++f) {
          const int k = j - f;
          if (k >= 0 && k < TAPS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gw[e][k] = fmaf(duv[f][e], xv[e], gw[e][k]);
          }
        }
      }
    }
  }
  // wpart[block][c][k]: the lane's 8 channels x TAPS taps are consecutive floats
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int k = 0; k < TAPS; ++k) red[wave][(lane * 8 + e) * TAPS + k] = gw[e][k];
  __syncthreads();
  for (int i = threadIdx.x; i < 512 * TAPS; i += 256)
    wpart[(long)blockIdx.x * a.C * a.taps + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// ---------------------------------------------------------------------------------------------------------
// Round 6: the whole backward of (depthwise conv -> channel norm) in ONE launch (the tile DESIGN has
// named since round 3).  The two run kernels above meet through HBM: the first writes du = d loss / d v for every frame,
// the second reads it back three times (its 10-row window, the run's own rows, and -- in the caller -- the per-sequence
// column sums).  Here a block owns FT = 26 consecutive frames of ONE sequence:
//   phase 0  the FT + 12 rows of x the tile touches go to LDS once (zero rows outside the sequence: no masks later);
//   phase 1  every wave recomputes v for 8 of the FT + 6 frames whose du the tile's dx needs (two runs of four, windows
//            out of LDS), turns them into du with the saved statistics and parks du (bf16, the rounding the two-launch
//            form stores) in LDS; the gamma / beta sums count the tile's own frames only;
//   phase 2  every wave takes 7 of the tile's frames: dx from a 13-row du window, tap gradients from a 13-row x window,
//            both out of LDS;
//   the four waves' partial sums meet in LDS (the x / du images are dead by then).
// Arithmetic and its order per frame are those of dwnorm_bwd_norm_run_kernel / dwnorm_bwd_conv_run_kernel: du and dx come
// out bitwise equal; the partial sums group other frames per block (fp32, reduced by the same vg_colsum_multi).
// HBM: dy, x, dx_add read once (+ 6 / 12 halo rows per 26, mostly L2 hits of the neighbouring block), dx (and du, when the
// caller wants it) written once.  70 KB of LDS, two blocks per CU.
// ---------------------------------------------------------------------------------------------------------
constexpr int FT = 26;              // frames a block owns
constexpr int FD = FT + 6;          // du frames in LDS
constexpr int FX = FT + 12;         // x rows in LDS
constexpr int FUSED_LDS = (FX + FD) * 1024;
constexpr int FRUN = 7;             // frames per wave in phase 2 (4 x 7 >= FT)
#ifndef VG_DW_ONE_RUN
#define VG_DW_ONE_RUN 1             // phase 2 as one run of 7 frames (13-row windows read once); 0: sub-runs of 4 + 3 (lab)
#endif

template <int TAPS>
__global__ __launch_bounds__(256, 2) void dwnorm_bwd_fused_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ w, const float* __restrict__ cbias,
                                                                  const float* __restrict__ temb, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                  const bf16_t* __restrict__ dx_add, bf16_t* __restrict__ du_out,
                                                                  bf16_t* __restrict__ dx, float* __restrict__ part,
                                                                  float* __restrict__ wpart, float* __restrict__ dupart, DwArgs a,
                                                                  int bps) {
  static_assert(TAPS == 7, "FD / FX are laid out for 7 taps");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;                     // x rows T0 - 6 .. T0 + FT + 5
  char* dsm = smem + FX * 1024;        // du frames A .. A + FD - 1
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.x / bps, tile = blockIdx.x - s * bps;
  int Tn, bte;
  long rbase;
  if (a.cu) {
    rbase = a.cu[s];
    Tn = a.cu[s + 1] - (int)rbase;
    bte = min(s, a.nbatch - 1);
  } else {
    Tn = a.Tn;
    rbase = (long)s * a.Tn;
    bte = s;
  }
  const int T0 = tile * FT;
  float* part_b = part + (long)blockIdx.x * 2 * a.C;
  float* wpart_b = wpart + (long)blockIdx.x * a.C * TAPS;
  if (T0 >= Tn) {                      // no frame of this sequence here: the reducers still read the block's rows
    for (int i = threadIdx.x; i < 2 * 512; i += 256) part_b[i] = 0.f;
    for (int i = threadIdx.x; i < 512 * TAPS; i += 256) wpart_b[i] = 0.f;
    if (dupart) for (int i = threadIdx.x; i < 512; i += 256) dupart[(long)blockIdx.x * a.C + i] = 0.f;
    return;
  }
  const int Tend = min(T0 + FT, Tn);   // own frames: [T0, Tend)
  const int A = T0 + a.shift - (TAPS - 1);
  const long ldy = a.ldy > 0 ? a.ldy : a.C;
  const bf16_t* xb = x + rbase * a.C + lane * 8;
  // ---- phase 0: every global read of the block is requested here
  uint4 xr[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) {
    const int r = wave + 4 * q;
    if (r < FX) xr[q] = *reinterpret_cast<const uint4*>(xb + (long)min(max(T0 - 6 + r, 0), Tn - 1) * a.C);
  }
  uint4 rdy[4];                        // the dy rows of the first run of four; the second run's are requested when these are done
#pragma unroll
  for (int i = 0; i < 4; ++i)
    rdy[i] = *reinterpret_cast<const uint4*>(dy + (rbase + min(max(A + 8 * wave + i, 0), Tn - 1)) * ldy + lane * 8);
  const long srow = rbase + min(max(A + 8 * wave + (lane & 7), 0), Tn - 1);
  const float st_mean = mean_in[srow], st_rstd = rstd_in[srow];       // lane i < 8: statistics of the wave's du frame i
  RunParams<TAPS> P;
  P.load(w, lane);
  float cbt[8], gm[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cbt[e] = 0.f; gm[e] = 1.f; }
  if (cbias) load8(cbias + lane * 8, cbt);
  if (gamma) load8(gamma + lane * 8, gm);
  if (temb) {
    float te[8];
    load8(temb + (long)bte * a.C + lane * 8, te);
#pragma unroll
    for (int e = 0; e < 8; ++e) cbt[e] += te[e];
  }
#pragma unroll
  for (int q = 0; q < 10; ++q) {
    const int r = wave + 4 * q;
    if (r < FX) {
      const int tx = T0 - 6 + r;
      const bool in = tx >= 0 && tx < Tn;
      *reinterpret_cast<uint4*>(xs + r * 1024 + lane * 16) = in ? xr[q] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  __syncthreads();
  // ---- phase 1: du of frames A + 8 wave .. + 7
  const float inv_c = 1.0f / (float)a.C, inv_c1 = 1.0f / (float)(a.C - 1);
  float sg[8], sb[8], sdu[8];          // sdu: column sums of the tile's own du rows (as stored: bf16) -- the caller's per-sequence
#pragma unroll                         // sums (time-embedding / conv-bias gradient) without a pass over du
  for (int e = 0; e < 8; ++e) { sg[e] = 0.f; sb[e] = 0.f; sdu[e] = 0.f; }
#pragma unroll
  for (int rn = 0; rn < 2; ++rn) {
    const int i0 = 8 * wave + 4 * rn;
    float v[4][8];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[f][e] = cbt[e];
#pragma unroll
    for (int j = 0; j < 4 + TAPS - 1; ++j) {
      float xv[8];
      V8<bf16_t>::expand(*reinterpret_cast<const uint4*>(xs + (i0 + j) * 1024 + lane * 16), xv);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int k = j - f;
        if (k >= 0 && k < TAPS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[f][e] = fmaf(P.w[e][k], xv[e], v[f][e]);
        }
      }
    }
    float s1[4], s2[4], rr[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int td = A + i0 + f;
      // (the builtin moves 32-bit integers: a float argument would be converted, not copied)
      const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st_mean), 4 * rn + f));
      const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st_rstd), 4 * rn + f));
      const float own = (td >= T0 && td < Tend) ? 1.f : 0.f;       // the partial sums count a frame once: in its own tile
      rr[f] = r;
      float dyv[8];
      V8<bf16_t>::expand(rdy[f], dyv);
      du_sums(dyv, gm, mean, v[f], s1[f], s2[f]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dd = dyv[e] * own;
        sg[e] = fmaf(dd * v[f][e], r, sg[e]);
        sb[e] += dd;
      }
      // pin the sums here: left free, the compiler sinks all eight frames' accumulation below the last barrier and keeps
      // every frame's dy and d alive until then (96 spilled registers)
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(sg[e]), "+v"(sb[e]));
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) s1[f] = wave_sum(s1[f]) * inv_c;
#pragma unroll
    for (int f = 0; f < 4; ++f) s2[f] = wave_sum(s2[f]) * rr[f] * rr[f] * rr[f] * inv_c1;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int td = A + i0 + f;
      const bool in = td >= 0 && td < Tn;                           // du outside the sequence: the convolution's zero padding
      float dyv[8], o[8];
      V8<bf16_t>::expand(rdy[f], dyv);
      du_value(dyv, gm, v[f], rr[f], s1[f], s2[f], o);
      bf16x8 ob;
#pragma unroll
      for (int e = 0; e < 8; ++e) ob[e] = in ? (bf16_t)o[e] : (bf16_t)0.0f;
      *reinterpret_cast<bf16x8*>(dsm + (i0 + f) * 1024 + lane * 16) = ob;
      if (td >= T0 && td < Tend) {
        if (du_out != nullptr) *reinterpret_cast<bf16x8*>(du_out + (rbase + td) * a.C + lane * 8) = ob;
        if (dupart != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sdu[e] += (float)ob[e];
            asm volatile("" : "+v"(sdu[e]));       // (pinned like sg / sb above)
          }
        }
      }
    }
    if (rn == 0) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        rdy[i] = *reinterpret_cast<const uint4*>(dy + (rbase + min(max(A + 8 * wave + 4 + i, 0), Tn - 1)) * ldy + lane * 8);
    }
  }
  // (requested here, not in phase 0: 28 more live registers through phase 1 spill; the other block of the CU covers the wait)
  uint4 radd[FRUN];
  if (dx_add) {
#pragma unroll
    for (int f = 0; f < FRUN; ++f)
      radd[f] = *reinterpret_cast<const uint4*>(dx_add + (rbase + min(T0 + FRUN * wave + f, Tn - 1)) * a.C + lane * 8);
  }
  __syncthreads();
  // ---- phase 2: own frames T0 + 7 wave .. + 6 (the sub-run form dates from the hunt for the spills that turned out to be the
  // compiler sinking the gamma / beta sums: with those pinned, seven frames at once fit)
  const int f0 = FRUN * wave;
  float gw[8][TAPS];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int k = 0; k < TAPS; ++k) gw[e][k] = 0.f;
  auto sub_run = [&](auto fb_c, auto nf_c) __attribute__((always_inline)) {
    constexpr int FB = decltype(fb_c)::value, NF = decltype(nf_c)::value;      // frames f0 + FB .. + NF - 1
    {
      float o[NF][8];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (dx_add) V8<bf16_t>::expand(radd[FB + f], o[f]);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[f][e] = 0.f;
        }
      }
      // du frame t + shift - k sits in LDS row (t - T0) + 6 - k: window row j = f + 6 - k of the sub-run's NF + 6
#pragma unroll
      for (int j = 0; j < NF + TAPS - 1; ++j) {
        float dv[8];
        V8<bf16_t>::expand(*reinterpret_cast<const uint4*>(dsm + min(f0 + FB + j, FD - 1) * 1024 + lane * 16), dv);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int k = f + TAPS - 1 - j;
          if (k >= 0 && k < TAPS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[f][e] = fmaf(P.w[e][k], dv[e], o[f][e]);
          }
        }
      }
#pragma unroll
      for (int f = 0; f < NF; ++f)
        if (T0 + f0 + FB + f < Tend) V8<bf16_t>::store(dx + (rbase + T0 + f0 + FB + f) * a.C + lane * 8, o[f]);
    }
    {
      // own du frame t: LDS row (t - T0) + 6 - shift; x frame t + k - shift: LDS row (t - T0) + 6 - shift + k
      const int base = f0 + FB + (TAPS - 1) - a.shift;
      float duv[NF][8];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        V8<bf16_t>::expand(*reinterpret_cast<const uint4*>(dsm + min(base + f, FD - 1) * 1024 + lane * 16), duv[f]);
        const float on = T0 + f0 + FB + f < Tend ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) duv[f][e] *= on;
      }
#pragma unroll
      for (int j = 0; j < NF + TAPS - 1; ++j) {
        float xv[8];
        V8<bf16_t>::expand(*reinterpret_cast<const uint4*>(xs + min(base + j, FX - 1) * 1024 + lane * 16), xv);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int k = j - f;
          if (k >= 0 && k < TAPS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gw[e][k] = fmaf(duv[f][e], xv[e], gw[e][k]);
          }
        }
      }
    }
  };
#if VG_DW_ONE_RUN
  sub_run(std::integral_constant<int, 0>{}, std::integral_constant<int, FRUN>{});
#else       // (lab: sub-runs of 4 + 3 re-read six window rows of each image: 31.8 against 31.35 us per launch; same bits)
  sub_run(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
  __builtin_amdgcn_sched_barrier(0);
  sub_run(std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{});
#endif
  // ---- the four waves' partial sums
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int k = 0; k < TAPS; ++k) red[wave * 512 * TAPS + (lane * 8 + e) * TAPS + k] = gw[e][k];
  __syncthreads();
  for (int i = threadIdx.x; i < 512 * TAPS; i += 256)
    wpart_b[i] = red[i] + red[512 * TAPS + i] + red[2 * 512 * TAPS + i] + red[3 * 512 * TAPS + i];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[wave * 1024 + lane * 8 + e] = sg[e];
    red[wave * 1024 + 512 + lane * 8 + e] = sb[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 512; i += 256) part_b[i] = red[i] + red[1024 + i] + red[2048 + i] + red[3072 + i];
  if (dupart != nullptr) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[4096 + wave * 512 + lane * 8 + e] = sdu[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256)
      dupart[(long)blockIdx.x * a.C + i] = red[4096 + i] + red[4096 + 512 + i] + red[4096 + 1024 + i] + red[4096 + 1536 + i];
  }
}

// du = r * (g - mean(g)) - r^3 / (C - 1) * d * sum(g * d),  g = dy * gamma, d = v - mean
// part[block][0][c] = sum_rows dy * xhat ; part[block][1][c] = sum_rows dy
template <typename T, int NV>
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
  __shared__ ParamLds P;
  stage_params(P, w, cbias, gamma, nullptr, a);
  LaneParams<T, NV> lp;
  lp.load(P, a, lane);
  float gm[NV][N], sg[NV][N], sb[NV][N];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < N; ++e) {
      gm[i][e] = P.gamma[(lane + 64 * i) * N + e];
      sg[i][e] = sb[i][e] = 0.f;
    }
  for (RowWalk walk(a.M); walk.valid(); walk.next()) {
    const int row = walk.row();
    float v[NV][8], g[NV][8];
    conv_row<T, NV>(x, lp, temb, a, row, lane, v);
    const float mean = mean_in[row], r = rstd_in[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float dyv[8];
      V8<T>::load(dy + (long)row * a.C + (lane + 64 * i) * N, dyv);
#pragma unroll
      for (int e = 0; e < N; ++e) {
        const float d = v[i][e] - mean;
        v[i][e] = d;
        g[i][e] = dyv[e] * gm[i][e];
        s1 += g[i][e];
        s2 += g[i][e] * d;
        sg[i][e] += dyv[e] * d * r;
        sb[i][e] += dyv[e];
      }
    }
    s1 = wave_sum(s1) / (float)a.C;
    s2 = wave_sum(s2) * r * r * r / (float)(a.C - 1);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float o[8];
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = r * (g[i][e] - s1) - s2 * v[i][e];
      V8<T>::store(du + (long)row * a.C + (lane + 64 * i) * N, o);
    }
  }
  float* scratch = &red[0][0];
#pragma unroll
  for (int which = 0; which < 2; ++which)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < N; ++e) scratch[(wave * 64 + lane) * N + e] = which ? sb[i][e] : sg[i][e];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          float t = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) t += scratch[(ww * 64 + lane) * N + e];
          part[((long)blockIdx.x * 2 + which) * a.C + (lane + 64 * i) * N + e] = t;
        }
      }
    }
}

// dx[t] = dx_add[t] + sum_k w[c][k] * du[t - (k - shift)] ; wpart[block][c][k] = sum_rows du[t] * x[t + k - shift]
template <typename T, int NV>
__global__ __launch_bounds__(256) void dwnorm_bwd_conv_kernel(const T* __restrict__ du, const T* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              const T* __restrict__ dx_add, T* __restrict__ dx,
                                                              float* __restrict__ wpart, DwArgs a) {
  constexpr int N = V8<T>::N;
  __shared__ float red[4][64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ ParamLds P;
  stage_params(P, w, nullptr, nullptr, nullptr, a);
  LaneParams<T, NV> lp;
  lp.load(P, a, lane);
  float gw[NV][MAXTAPS][N];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int k = 0; k < MAXTAPS; ++k)
#pragma unroll
      for (int e = 0; e < N; ++e) gw[i][k][e] = 0.f;
  for (RowWalk walk(a.M); walk.valid(); walk.next()) {
    const int row = walk.row();
    const int b = row / a.Tn, t = row - b * a.Tn;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      float o[8], duv[8];
      if (dx_add) V8<T>::load(dx_add + (long)row * a.C + c * N, o);
      else {
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = 0.f;
      }
      V8<T>::load(du + (long)row * a.C + c * N, duv);
      // request every neighbour row first (clamped addresses), then accumulate with 0/1 masks
      uint4 dr[MAXTAPS], xr[MAXTAPS];
#pragma unroll
      for (int k = 0; k < MAXTAPS; ++k) {
        if (k < a.taps) {
          const int td = min(max(t - (k - a.shift), 0), a.Tn - 1);   // output frame whose tap k read input frame t
          const int ts = min(max(t + k - a.shift, 0), a.Tn - 1);     // input frame tap k of output frame t reads
          dr[k] = V8<T>::raw(du + ((long)b * a.Tn + td) * a.C + c * N);
          xr[k] = V8<T>::raw(x + ((long)b * a.Tn + ts) * a.C + c * N);
        }
      }
#pragma unroll
      for (int k = 0; k < MAXTAPS; ++k) {
        if (k < a.taps) {
          const int td = t - (k - a.shift), ts = t + k - a.shift;
          const float on_d = (td >= 0 && td < a.Tn) ? 1.f : 0.f, on_s = (ts >= 0 && ts < a.Tn) ? 1.f : 0.f;
          float dv[8], xv[8];
          V8<T>::expand(dr[k], dv);
          V8<T>::expand(xr[k], xv);
#pragma unroll
          for (int e = 0; e < N; ++e) {
            o[e] = fmaf(lp.w[i][k][e] * on_d, dv[e], o[e]);
            gw[i][k][e] = fmaf(duv[e] * on_s, xv[e], gw[i][k][e]);
          }
        }
      }
      V8<T>::store(dx + (long)row * a.C + c * N, o);
    }
  }
  float* scratch = &red[0][0];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int k = 0; k < MAXTAPS; ++k) {
      if (k >= a.taps) break;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < N; ++e) scratch[(wave * 64 + lane) * N + e] = gw[i][k][e];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          float t = 0.f;
#pragma unroll
          for (int ww = 0; ww < 4; ++ww) t += scratch[(ww * 64 + lane) * N + e];
          wpart[(long)blockIdx.x * a.C * a.taps + ((lane + 64 * i) * N + e) * a.taps + k] = t;
        }
      }
    }
}

int check_shape(const char* who, int M, int C, int T, int taps, int dtype) {
  const int n = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "%s: bad dtype %d", who, dtype);
  VG_REQUIRE(M > 0 && T > 0 && M % T == 0, "%s: M=%d must be a multiple of T=%d", who, M, T);
  VG_REQUIRE(C <= MAXC && C % (64 * n) == 0 && C / (64 * n) <= 2, "%s: C=%d unsupported (multiple of %d, at most %d)", who, C,
             64 * n, 128 * n);
  VG_REQUIRE(taps >= 0 && taps <= MAXTAPS, "%s: taps=%d unsupported", who, taps);
  return 0;
}

template <typename T, int NV>
void launch_fwd(const void* x, const float* w, const float* cbias, const float* temb, const float* gamma,
                const float* beta, void* y, float* mean, float* rstd, const DwArgs& a, int nb, hipStream_t stream) {
  static const int wpb = [] { const char* e = getenv("VG_DW_WAVES"); return e ? atoi(e) : 4; }();
  dwnorm_fwd_kernel<T, NV><<<dim3(nb), dim3(64 * wpb), 0, stream>>>((const T*)x, w, cbias, temb, gamma, beta, (T*)y, mean,
                                                              rstd, a);
}
template <typename T, int NV>
void launch_bwd(const void* dy, const void* x, const float* w, const float* cbias, const float* temb,
                const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du, void* dx,
                float* norm_part, float* w_part, const DwArgs& a, int nb, hipStream_t stream) {
  dwnorm_bwd_norm_kernel<T, NV><<<dim3(nb), dim3(256), 0, stream>>>((const T*)dy, (const T*)x, w, cbias, temb, gamma,
                                                                   mean, rstd, (T*)du, norm_part, a);
  if (a.taps > 0)
    dwnorm_bwd_conv_kernel<T, NV><<<dim3(nb), dim3(256), 0, stream>>>((const T*)du, (const T*)x, w,
                                                                     (const T*)dx_add, (T*)dx, w_part, a);
}

}  // namespace

extern "C" int vg_dwnorm_blocks(int M) {
  const int b = (M + 3) / 4;
  return b < 512 ? b : 512;
}

extern "C" int vg_dwnorm_fwd(const void* x, const float* w, const float* cbias, const float* temb,
                             const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int C,
                             int T, int taps, int shift, float eps, int dtype, hipStream_t stream) {
  if (int e = check_shape("vg_dwnorm_fwd", M, C, T, taps, dtype)) return e;
  DwArgs a{M, C, T, taps, shift, eps};
  static const int nb_env = [] { const char* e = getenv("VG_DW_BLOCKS"); return e ? atoi(e) : 0; }();
  const int nb = min((M + 3) / 4, nb_env > 0 ? nb_env : 512);   // measured: 2048 waves balance per-wave set-up against parallelism
  const int nv = C / (64 * (dtype == VG_BF16 ? 8 : 4));
  // algorithmic bytes: every frame read once and written once (the taps re-read neighbours from cache), + statistics
  const int tok = vg_host::prof_begin(VG_PROF_DWNORM_FWD, (double)M * (2.0 * C * (dtype == VG_BF16 ? 2 : 4) + 8.0), stream);
  static const int runs_off = [] { const char* e = getenv("VG_DW_RUNS"); return e && atoi(e) == 0; }();
  if (dtype == VG_BF16 && nv == 1 && taps == 7 && !runs_off) {
    const int nruns = (M / T) * ((T + RUNF - 1) / RUNF);
    dwnorm_fwd_run_kernel<7><<<dim3(min((nruns + 3) / 4, 1024)), dim3(256), 0, stream>>>(
        (const bf16_t*)x, w, cbias, temb, gamma, beta, (bf16_t*)y, mean, rstd, a);
  } else if (dtype == VG_BF16) {
    if (nv == 1) launch_fwd<bf16_t, 1>(x, w, cbias, temb, gamma, beta, y, mean, rstd, a, nb, stream);
    else launch_fwd<bf16_t, 2>(x, w, cbias, temb, gamma, beta, y, mean, rstd, a, nb, stream);
  } else {
    if (nv == 1) launch_fwd<float, 1>(x, w, cbias, temb, gamma, beta, y, mean, rstd, a, nb, stream);
    else launch_fwd<float, 2>(x, w, cbias, temb, gamma, beta, y, mean, rstd, a, nb, stream);
  }
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_dwnorm_fwd");
}

extern "C" int vg_dwnorm_bwd(const void* dy, const void* x, const float* w, const float* cbias, const float* temb,
                             const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du,
                             void* dx, float* norm_part, float* w_part, int M, int C, int T, int taps, int shift,
                             int dtype, hipStream_t stream) {
  if (int e = check_shape("vg_dwnorm_bwd", M, C, T, taps, dtype)) return e;
  DwArgs a{M, C, T, taps, shift, 0.f};
  const int nb = vg_dwnorm_blocks(M);
  const int nv = C / (64 * (dtype == VG_BF16 ? 8 : 4));
  // algorithmic bytes: dy, x read; du, dx written (+ the residual-path gradient when given)
  const int tok = vg_host::prof_begin(VG_PROF_DWNORM_BWD, (double)M * ((dx_add ? 5.0 : 4.0) * C * (dtype == VG_BF16 ? 2 : 4) + 8.0),
                                      stream);
  static const int runs_off = [] { const char* e = getenv("VG_DW_RUNS"); return e && atoi(e) == 0; }();
  if (dtype == VG_BF16 && nv == 1 && taps == 7 && C == 512 && !runs_off) {
    // (the grid stays vg_dwnorm_blocks(M): the caller sized the partial-sum arrays for it)
    dwnorm_bwd_norm_run_kernel<7, 4><<<dim3(nb), dim3(256), 0, stream>>>((const bf16_t*)dy, (const bf16_t*)x, w, cbias, temb, gamma,
                                                                     mean, rstd, (bf16_t*)du, norm_part, a);
    dwnorm_bwd_conv_run_kernel<7, 4><<<dim3(nb), dim3(256), 0, stream>>>((const bf16_t*)du, (const bf16_t*)x, w,
                                                                     (const bf16_t*)dx_add, (bf16_t*)dx, w_part, a);
  } else if (dtype == VG_BF16) {
    if (nv == 1) launch_bwd<bf16_t, 1>(dy, x, w, cbias, temb, gamma, mean, rstd, dx_add, du, dx, norm_part, w_part, a, nb, stream);
    else launch_bwd<bf16_t, 2>(dy, x, w, cbias, temb, gamma, mean, rstd, dx_add, du, dx, norm_part, w_part, a, nb, stream);
  } else {
    if (nv == 1) launch_bwd<float, 1>(dy, x, w, cbias, temb, gamma, mean, rstd, dx_add, du, dx, norm_part, w_part, a, nb, stream);
    else launch_bwd<float, 2>(dy, x, w, cbias, temb, gamma, mean, rstd, dx_add, du, dx, norm_part, w_part, a, nb, stream);
  }
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_dwnorm_bwd");
}

// ---- packed rows (round 5): the same block on ragged sequences laid end to end -- sequence s = rows [cu_rows[s],
// cu_rows[s + 1]), each with its own zero padding; temb row min(s, nbatch - 1).  bf16, 512 channels, 7 taps (the run
// kernels: every conv block of vae-gslm.yaml); anything else is refused and the caller keeps padded rows.
static int check_seg(const char* who, int M, int C, const int* cu_rows, int nseq, int nbatch, int taps, int dtype) {
  VG_REQUIRE(dtype == VG_BF16 && C == 512 && taps == 7, "%s: packed rows need bf16, C = 512, 7 taps (C=%d taps=%d dtype=%d)", who, C,
             taps, dtype);
  VG_REQUIRE(M > 0 && cu_rows != nullptr && nseq >= 1 && nseq <= 64 && nbatch >= 1 && nbatch <= nseq,
             "%s: M=%d nseq=%d (1..64) nbatch=%d", who, M, nseq, nbatch);
  return 0;
}

extern "C" int vg_dwnorm_fwd_seg(const void* x, const float* w, const float* cbias, const float* temb, const float* gamma,
                                 const float* beta, void* y, float* mean, float* rstd, int M, int C, const int* cu_rows,
                                 int nseq, int nbatch, int taps, int shift, float eps, int dtype, hipStream_t stream) {
  if (int e = check_seg("vg_dwnorm_fwd_seg", M, C, cu_rows, nseq, nbatch, taps, dtype)) return e;
  DwArgs a{M, C, M, taps, shift, eps, cu_rows, nseq, nbatch};
  const int tok = vg_host::prof_begin(VG_PROF_DWNORM_FWD, (double)M * (2.0 * C * 2 + 8.0), stream);
  const int nruns = M / RUNF + nseq;          // an upper bound: the kernel counts the real ones from cu_rows
  dwnorm_fwd_run_kernel<7><<<dim3(min((nruns + 3) / 4, 1024)), dim3(256), 0, stream>>>(
      (const bf16_t*)x, w, cbias, temb, gamma, beta, (bf16_t*)y, mean, rstd, a);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_dwnorm_fwd_seg");
}

extern "C" int vg_dwnorm_bwd_seg(const void* dy, const void* x, const float* w, const float* cbias, const float* temb,
                                 const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du,
                                 void* dx, float* norm_part, float* w_part, int M, int C, const int* cu_rows, int nseq,
                                 int nbatch, int taps, int shift, int dtype, hipStream_t stream) {
  if (int e = check_seg("vg_dwnorm_bwd_seg", M, C, cu_rows, nseq, nbatch, taps, dtype)) return e;
  DwArgs a{M, C, M, taps, shift, 0.f, cu_rows, nseq, nbatch};
  const int nb = vg_dwnorm_blocks(M);          // the caller sized the partial-sum arrays for it
  const int tok = vg_host::prof_begin(VG_PROF_DWNORM_BWD, (double)M * ((dx_add ? 5.0 : 4.0) * C * 2 + 8.0), stream);
  dwnorm_bwd_norm_run_kernel<7, 4><<<dim3(nb), dim3(256), 0, stream>>>((const bf16_t*)dy, (const bf16_t*)x, w, cbias, temb, gamma,
                                                                   mean, rstd, (bf16_t*)du, norm_part, a);
  dwnorm_bwd_conv_run_kernel<7, 4><<<dim3(nb), dim3(256), 0, stream>>>((const bf16_t*)du, (const bf16_t*)x, w,
                                                                   (const bf16_t*)dx_add, (bf16_t*)dx, w_part, a);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_dwnorm_bwd_seg");
}


// ---- round 6: the conv block's conditioning merged into its first 1x1 convolution (DwArgs::ldy / cond).  bf16, C = 512,
// 7 taps (the run kernels: every conditional block of vae-gslm.yaml); cu_rows == nullptr: M / T sequences of T rows,
// else packed rows as in vg_dwnorm_fwd_seg.
static int check_cat(const char* who, int M, int C, int T, const int* cu_rows, int nseq, int nbatch, int taps, int dtype, long ldy) {
  VG_REQUIRE(dtype == VG_BF16 && C == 512 && taps == 7, "%s: needs bf16, C = 512, 7 taps (C=%d taps=%d dtype=%d)", who, C, taps, dtype);
  VG_REQUIRE(M > 0 && ldy >= C && ldy % 8 == 0, "%s: M=%d ldy=%ld", who, M, ldy);
  if (cu_rows != nullptr) VG_REQUIRE(nseq >= 1 && nseq <= 64 && nbatch >= 1 && nbatch <= nseq, "%s: nseq=%d (1..64) nbatch=%d", who, nseq, nbatch);
  else VG_REQUIRE(T > 0 && M % T == 0, "%s: M=%d is not a multiple of T=%d", who, M, T);
  return 0;
}

extern "C" int vg_dwnorm_fwd_cat(const void* x, const float* w, const float* cbias, const float* temb, const float* gamma,
                                 const float* beta, void* y, int64_t ldy, const void* cond, int64_t ldcond, int cond_cols,
                                 float* mean, float* rstd, int M, int C, int T, const int* cu_rows, int nseq, int nbatch,
                                 int taps, int shift, float eps, int dtype, hipStream_t stream) {
  if (int e = check_cat("vg_dwnorm_fwd_cat", M, C, T, cu_rows, nseq, nbatch, taps, dtype, (long)ldy)) return e;
  VG_REQUIRE(cond == nullptr || (cond_cols > 0 && cond_cols <= 64 && cond_cols % 8 == 0 && ldcond % 8 == 0 && ldy >= C + 64 &&
                                 ((uintptr_t)cond % 16) == 0),
             "vg_dwnorm_fwd_cat: the condition tail is 64 columns of 16-byte pieces (cond_cols=%d ldcond=%ld ldy=%ld)", cond_cols,
             (long)ldcond, (long)ldy);
  DwArgs a{M, C, cu_rows ? M : T, taps, shift, eps, cu_rows, cu_rows ? nseq : 0, cu_rows ? nbatch : 0};
  a.ldy = (long)ldy;
  a.cond = (const bf16_t*)cond;
  a.ldcond = (long)ldcond;
  a.cond_cols = cond ? cond_cols : 0;
  const int tok = vg_host::prof_begin(VG_PROF_DWNORM_FWD, (double)M * (2.0 * C * 2 + 8.0 + (cond ? 2.0 * cond_cols + 128.0 : 0.0)), stream);
  const int nruns = cu_rows ? M / RUNF + nseq : (M / T) * ((T + RUNF - 1) / RUNF);
  dwnorm_fwd_run_kernel<7><<<dim3(min((nruns + 3) / 4, 1024)), dim3(256), 0, stream>>>(
      (const bf16_t*)x, w, cbias, temb, gamma, beta, (bf16_t*)y, mean, rstd, a);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_dwnorm_fwd_cat");
}

extern "C" int vg_dwnorm_bwd_ld(const void* dy, int64_t ldy, const void* x, const float* w, const float* cbias, const float* temb,
                                const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du, void* dx,
                                float* norm_part, float* w_part, int M, int C, int T, const int* cu_rows, int nseq, int nbatch,
                                int taps, int shift, int dtype, hipStream_t stream) {
  if (int e = check_cat("vg_dwnorm_bwd_ld", M, C, T, cu_rows, nseq, nbatch, taps, dtype, (long)ldy)) return e;
  DwArgs a{M, C, cu_rows ? M : T, taps, shift, 0.f, cu_rows, cu_rows ? nseq : 0, cu_rows ? nbatch : 0};
  a.ldy = (long)ldy;
  const int nb = vg_dwnorm_blocks(M);          // the caller sized the partial-sum arrays for it
  const int tok = vg_host::prof_begin(VG_PROF_DWNORM_BWD, (double)M * ((dx_add ? 5.0 : 4.0) * C * 2 + 8.0), stream);
  dwnorm_bwd_norm_run_kernel<7, 4><<<dim3(nb), dim3(256), 0, stream>>>((const bf16_t*)dy, (const bf16_t*)x, w, cbias, temb, gamma,
                                                                   mean, rstd, (bf16_t*)du, norm_part, a);
  a.ldy = 0;                                   // (the second kernel reads the first one's dense du)
  dwnorm_bwd_conv_run_kernel<7, 4><<<dim3(nb), dim3(256), 0, stream>>>((const bf16_t*)du, (const bf16_t*)x, w,
                                                                   (const bf16_t*)dx_add, (bf16_t*)dx, w_part, a);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_dwnorm_bwd_ld");
}

// ---- round 6: the one-launch backward (dwnorm_bwd_fused_kernel).  bf16, C = 512, 7 taps; cu_rows == nullptr: M / T
// sequences of T rows (max_len = T), else packed rows whose sequences have at most max_len rows.  The partial-sum arrays
// have vg_dwnorm_bwd_fused_blocks(nseq, max_len) rows; du may be nullptr (nobody reads it); du_part (nullable): [blocks][C]
// column sums of the du rows of each block (block b belongs to sequence b / (blocks / nseq)).
extern "C" int vg_dwnorm_bwd_fused_blocks(int nseq, int max_len) { return nseq * ((max_len + FT - 1) / FT); }

extern "C" int vg_dwnorm_bwd_fused(const void* dy, int64_t ldy, const void* x, const float* w, const float* cbias, const float* temb,
                                   const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du, void* dx,
                                   float* norm_part, float* w_part, float* du_part, int M, int C, int T, const int* cu_rows, int nseq,
                                   int nbatch, int max_len, int taps, int shift, int dtype, hipStream_t stream) {
  if (int e = check_cat("vg_dwnorm_bwd_fused", M, C, T, cu_rows, nseq, nbatch, taps, dtype, (long)ldy)) return e;
  VG_REQUIRE(shift >= 0 && shift <= 6, "vg_dwnorm_bwd_fused: shift=%d (0..6)", shift);
  const int ns = cu_rows ? nseq : M / T, ml = cu_rows ? max_len : T;
  VG_REQUIRE(ml > 0 && ns > 0, "vg_dwnorm_bwd_fused: nseq=%d max_len=%d", ns, ml);
  DwArgs a{M, C, cu_rows ? M : T, taps, shift, 0.f, cu_rows, cu_rows ? nseq : 0, cu_rows ? nbatch : 0};
  a.ldy = (long)ldy;
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwnorm_bwd_fused_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              FUSED_LDS);
    return true;
  }();
  (void)once;
  const int bps = (ml + FT - 1) / FT;
  const int tok = vg_host::prof_begin(VG_PROF_DWNORM_BWD, (double)M * ((dx_add ? 4.0 : 3.0) * C * 2 + (du ? 2.0 * C : 0.0) + 8.0), stream);
  dwnorm_bwd_fused_kernel<7><<<dim3(ns * bps), dim3(256), FUSED_LDS, stream>>>(
      (const bf16_t*)dy, (const bf16_t*)x, w, cbias, temb, gamma, mean, rstd, (const bf16_t*)dx_add, (bf16_t*)du, (bf16_t*)dx, norm_part,
      w_part, du_part, a, bps);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_dwnorm_bwd_fused");
}
