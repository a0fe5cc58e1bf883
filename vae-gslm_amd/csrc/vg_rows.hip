// Row-wise (HBM-bound) kernels: RMSNorm, token cross-entropy, the VAE
// reparameterisation / prior log-density / KL terms, and the small
// deterministic reductions that go with them.  One wave64 per frame for the
// wide rows (coalesced 16-byte loads, shuffle reductions, fp32 math); one
// thread per element/row for the 4-wide latent tensors.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

using namespace vg;

// ------------------------------------------------------------------ host error plumbing
namespace vg_host {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}
}  // namespace vg_host

extern "C" int vg_version(void) { return 100; }
extern "C" int vg_last_error(char* buf, int buflen) {
  if (buf == nullptr || buflen <= 0) return 1;
  strncpy(buf, vg_host::g_err, buflen - 1);
  buf[buflen - 1] = 0;
  return 0;
}

namespace {

constexpr int MAXV = 4;   // 16-byte vectors per lane per row (C <= 2048 bf16 / 1024 f32)
constexpr float HALF_LOG_2PI = 0.91893853320467274178f;

template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  typedef f32x4 raw_t;
  static VG_DEVICE raw_t load_raw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static VG_DEVICE void unpack(raw_t v, float (&o)[8]) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
  static VG_DEVICE void load(const float* p, float (&o)[8]) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  }
  static VG_DEVICE void store(float* p, const float (&o)[8]) {
    f32x4 v = {o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(p) = v;
  }
};
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  typedef bf16x8 raw_t;
  static VG_DEVICE raw_t load_raw(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static VG_DEVICE void unpack(raw_t v, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
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

// ------------------------------------------------------------------ RMSNorm forward
// one wave per frame, grid-stride over frames; the scale vector is read once per wave and the
// next frame's loads are issued before the current frame's reduction (two frames in flight)
template <typename T, int NV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                          T* __restrict__ y, float* __restrict__ rstd, int M, int C,
                                                          float eps, const int* __restrict__ lengths, int Tlen) {
  constexpr int N = Vec<T>::N;
  const int lane = threadIdx.x & 63;
  const int nvec = C / N;
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float sc[NV][8], v[NV][8], nx[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) {
      Vec<T>::load(x + (long)row * C + c * N, v[i]);
#pragma unroll
      for (int e = 0; e < N; ++e) sc[i][e] = scale[c * N + e];
    }
  }
  const float inv_c = 1.0f / (float)C;
  for (; row < M; row += stride) {
    const int nrow = row + stride;
    if (nrow < M) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) Vec<T>::load(x + (long)nrow * C + c * N, nx[i]);
      }
    }
    const bool valid = row_valid(lengths, Tlen, row);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
#pragma unroll
        for (int e = 0; e < N; ++e) ss += v[i][e] * v[i][e];
      }
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss * inv_c + eps);
    if (lane == 0) rstd[row] = r;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        float o[8];
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = valid ? sc[i][e] * (v[i][e] * r) : 0.f;
        Vec<T>::store(y + (long)row * C + c * N, o);
#pragma unroll
        for (int e = 0; e < N; ++e) v[i][e] = nx[i][e];
      }
    }
  }
}

// ------------------------------------------------------------------ RMSNorm backward
// dx = dx_add + mask * rstd * (g - xhat * mean(g * xhat)),  g = dy * scale, xhat = x * rstd
// dscale_partial[block][c] = sum over this block's valid rows of dy * xhat
// COLS (round 4): dx_colsum_partial[block][c] = sum over this block's rows of the STORED dx (rounded to T first): the
// layer below needs the column sums of exactly this tensor for its FFN-out bias gradient -- a 33 MB read of its own
// otherwise (colsum_partials, ~10 us per layer); here it rides on a kernel that is bound by its four row streams.
template <typename T, int NV, bool COLS>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ rstd, const T* __restrict__ dx_add,
                                                          T* __restrict__ dx, float* __restrict__ dscale_partial,
                                                          float* __restrict__ dx_colsum_partial,
                                                          int M, int C, const int* __restrict__ lengths, int Tlen) {
  constexpr int N = Vec<T>::N;
  __shared__ float red[4][64 * 8];             // [wave][lane * N + e], N <= 8
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C / N;
  float ds[NV][8], cs[COLS ? NV : 1][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ds[i][e] = 0.f;
      if constexpr (COLS) cs[i][e] = 0.f;
    }

  // two frames in flight per wave: the next frame's three row segments are requested (and kept packed) before
  // this frame's reduction
  typedef typename Vec<T>::raw_t raw_t;
  const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  raw_t ra[NV], rb[NV], ro[NV], na[NV], nb[NV], no[NV];
  auto fetch = [&](int rw, raw_t (&fa)[NV], raw_t (&fb)[NV], raw_t (&fo)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        fa[i] = Vec<T>::load_raw(dy + (long)rw * C + c * N);
        fb[i] = Vec<T>::load_raw(x + (long)rw * C + c * N);
        if (dx_add) fo[i] = Vec<T>::load_raw(dx_add + (long)rw * C + c * N);
      }
    }
  };
  if (row < M) fetch(row, ra, rb, ro);
  for (; row < M; row += stride) {
    const int nrow = row + stride;
    if (nrow < M) fetch(nrow, na, nb, no);
    const bool valid = row_valid(lengths, Tlen, row);
    const float r = rstd[row];
    float g[NV][8], xh[NV][8];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        float a[8], b[8];
        Vec<T>::unpack(ra[i], a);
        Vec<T>::unpack(rb[i], b);
#pragma unroll
        for (int e = 0; e < N; ++e) {
          xh[i][e] = b[e] * r;
          g[i][e] = a[e] * scale[c * N + e];
          dot += g[i][e] * xh[i][e];
          if (valid) ds[i][e] += a[e] * xh[i][e];
        }
      }
    }
    dot = wave_sum(dot) / (float)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        float out[8];
        if (dx_add) Vec<T>::unpack(ro[i], out);
        else {
#pragma unroll
          for (int e = 0; e < N; ++e) out[e] = 0.f;
        }
        if (valid) {
#pragma unroll
          for (int e = 0; e < N; ++e) out[e] += r * (g[i][e] - xh[i][e] * dot);
        }
        Vec<T>::store(dx + (long)row * C + c * N, out);
        if constexpr (COLS) {
#pragma unroll
          for (int e = 0; e < N; ++e) cs[i][e] += to_f32<T>(from_f32<T>(out[e]));
        }
        ra[i] = na[i];
        rb[i] = nb[i];
        ro[i] = no[i];
      }
    }
  }
  // cross-wave reduction of the scale-gradient partials, NV passes of [4][64*N]
  float* scratch = &red[0][0];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < N; ++e) scratch[(wave * 64 + lane) * N + e] = ds[i][e];
    __syncthreads();
    const int c = lane + 64 * i;
    if (wave == 0 && c < nvec) {
#pragma unroll
      for (int e = 0; e < N; ++e) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += scratch[(w * 64 + lane) * N + e];
        dscale_partial[(long)blockIdx.x * C + c * N + e] = s;
      }
    }
  }
  if constexpr (COLS) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < N; ++e) scratch[(wave * 64 + lane) * N + e] = cs[i][e];
      __syncthreads();
      const int c = lane + 64 * i;
      if (wave == 0 && c < nvec) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) t += scratch[(w * 64 + lane) * N + e];
          dx_colsum_partial[(long)blockIdx.x * C + c * N + e] = t;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ column sums
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, int M, int N, long ld, float* __restrict__ out) {
  __shared__ float red[4][64][4];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c0 = (blockIdx.x * 64 + tx) * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    for (int m = blockIdx.y * 4 + ty; m < M; m += gridDim.y * 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += to_f32<T>(x[(long)m * ld + c0 + e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[ty][tx][e] = s[e];
  __syncthreads();
  if (ty == 0 && c0 < N) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      out[(long)blockIdx.y * N + c0 + e] = red[0][tx][e] + red[1][tx][e] + red[2][tx][e] + red[3][tx][e];
  }
}

// first stage with 16-byte loads (8 bf16 / 4 fp32 columns per thread, two rows in flight per thread)
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ x, int M, int N, long ld,
                                                         float* __restrict__ out) {
  constexpr int V = Vec<T>::N;
  __shared__ float red[4][64][V];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c0 = (blockIdx.x * 64 + tx) * V;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    const int step = gridDim.y * 4;
    int m = blockIdx.y * 4 + ty;
    for (; m + step < M; m += 2 * step) {
      float a[8], b[8];
      Vec<T>::load(x + (long)m * ld + c0, a);
      Vec<T>::load(x + (long)(m + step) * ld + c0, b);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += a[e] + b[e];
    }
    if (m < M) {
      float a[8];
      Vec<T>::load(x + (long)m * ld + c0, a);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += a[e];
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[ty][tx][e] = s[e];
  __syncthreads();
  if (ty == 0 && c0 < N) {
#pragma unroll
    for (int e = 0; e < V; ++e)
      out[(long)blockIdx.y * N + c0 + e] = red[0][tx][e] + red[1][tx][e] + red[2][tx][e] + red[3][tx][e];
  }
}

// final / small-input stage: one column per thread, 16 row lanes per block, optional accumulate
template <typename T>
__global__ __launch_bounds__(1024) void colsum_final_kernel(const T* __restrict__ x, int M, int N, long ld,
                                                            float* __restrict__ out, int accumulate) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (c < N)
    for (int m = ty; m < M; m += 16) s += to_f32<T>(x[(long)m * ld + c]);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][tx];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// fp32 partial-sum arrays ([<=2048 rows][N], the second stage of every two-stage reduction): 16 columns
// per block so that a 1024-column array still spreads over 64 CUs, 64 row lanes with all loads of a
// thread independent (one memory latency instead of M/16 dependent round trips)
__global__ __launch_bounds__(256) void colsum_small_f32_kernel(const float* __restrict__ x, int M, int N, long ld,
                                                               float* __restrict__ out, int accumulate) {
  __shared__ float red[64][17];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c0 = blockIdx.x * 16 + cl * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
#pragma unroll 8
    for (int m = rl; m < M; m += 64) s += *reinterpret_cast<const f32x4*>(x + (long)m * ld + c0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][cl * 4 + e] = s[e];
  __syncthreads();
  // 256 threads: column = tid & 15, 16 row groups of 4 -> shuffle-free second step through LDS
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  float t = red[4 * g][c] + red[4 * g + 1][c] + red[4 * g + 2][c] + red[4 * g + 3][c];
  __syncthreads();
  red[g][c] = t;
  __syncthreads();
  if (threadIdx.x < 16 && blockIdx.x * 16 + c < N) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v += red[r][c];
    float* o = out + blockIdx.x * 16 + c;
    *o = accumulate ? *o + v : v;
  }
}

__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = threadIdx.x < 16 ? red[threadIdx.x] : 0.f;
    t = wave_sum(t);
    if (threadIdx.x == 0) out[0] = t;
  }
}

__global__ void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      f32x4 a = *reinterpret_cast<const f32x4*>(src + i);
      f32x4 b = *reinterpret_cast<const f32x4*>(src + i + 4);
      bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3],
                  (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
      *reinterpret_cast<bf16x8*>(dst + i) = o;
    } else {
      for (long j = i; j < n; ++j) dst[j] = (bf16_t)src[j];
    }
  }
}

// ------------------------------------------------------------------ activation backward
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ aux, T* __restrict__ dx, long n,
                               int act) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float g = to_f32<T>(dy[i]), a = to_f32<T>(aux[i]);
    float r = g;
    if (act == VG_ACT_RELU) r = a > 0.f ? g : 0.f;
    else if (act == VG_ACT_GELU) r = g * gelu_erf_grad(a);
    else if (act == VG_ACT_SILU) r = g * silu_grad(a);
    dx[i] = from_f32<T>(r);
  }
}

// ------------------------------------------------------------------ row mask (apply_mask on [M][C] rows)
template <typename T>
__global__ void mask_rows_kernel(const T* __restrict__ x, T* __restrict__ y, long n, int C,
                                 const int* __restrict__ lengths, int Tlen) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = row_valid(lengths, Tlen, (int)(i / C)) ? x[i] : from_f32<T>(0.f);
}

// ------------------------------------------------------------------ row gather (packed <-> padded rows)
// dst[i][:] = map[i] >= 0 ? src[map[i]][:] : 0, 16 bytes per lane.  Packing the valid frames of a right-padded batch
// (map = the frame of packed row i) and un-packing them (map = the packed row of frame i, -1 on padded frames) are
// both this gather, and each is the other's backward: no scatter, no atomics, nothing to pre-zero.
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int* __restrict__ map,
                                                          uint4* __restrict__ dst, int n_dst, int vec_per_row) {
  const long total = (long)n_dst * vec_per_row, stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int row = (int)(i / vec_per_row), v = (int)(i - (long)row * vec_per_row);
    const int m = map[row];
    dst[i] = m >= 0 ? src[(long)m * vec_per_row + v] : make_uint4(0, 0, 0, 0);
  }
}

// ------------------------------------------------------------------ token cross-entropy
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const long* __restrict__ targets,
                                                     float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                     int* __restrict__ argmax, int M, int V, long ld,
                                                     const int* __restrict__ lengths, int Tlen) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const T* p = logits + (long)row * ld;
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    const float v = to_f32<T>(p[c]);
    if (v > mx) { mx = v; mi = c; }
  }
  // wave arg-max (lowest index wins ties, like torch.argmax on CPU)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
  }
  float se = 0.f;
  for (int c = lane; c < V; c += 64) se += expf(to_f32<T>(p[c]) - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  if (lane == 0) {
    const bool valid = row_valid(lengths, Tlen, row);
    const long t = targets[row];
    const bool use = valid && t >= 0 && t < V;
    loss_rows[row] = use ? lse - to_f32<T>(p[t]) : 0.f;
    lse_out[row] = lse;
    argmax[row] = mi;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const long* __restrict__ targets,
                                                     const float* __restrict__ lse, const float* __restrict__ gscale,
                                                     T* __restrict__ dlogits, int M, int V, long ld,
                                                     const int* __restrict__ lengths, int Tlen) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const bool valid = row_valid(lengths, Tlen, row);
  const long t = targets[row];
  const bool use = valid && t >= 0 && t < V;
  const float g = gscale[0];
  const float l = lse[row];
  for (int c = lane; c < V; c += 64) {
    float d = 0.f;
    if (use) d = g * (expf(to_f32<T>(logits[(long)row * ld + c]) - l) - (c == t ? 1.f : 0.f));
    dlogits[(long)row * ld + c] = from_f32<T>(d);
  }
}

// ------------------------------------------------------------------ VAE terms
__global__ void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ ls,
                                   const float* __restrict__ eps, float* __restrict__ z, float* __restrict__ log_q,
                                   int M, int D, float temp, const int* __restrict__ lengths, int Tlen) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * D) return;
  const bool valid = row_valid(lengths, Tlen, (int)(i / D));
  const float s = ls[i];
  z[i] = valid ? mu[i] + eps[i] * expf(s) * temp : 0.f;
  log_q[i] = valid ? -s - 0.5f - HALF_LOG_2PI : 0.f;
}

__global__ void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ dlog_q,
                                   const float* __restrict__ ls, const float* __restrict__ eps,
                                   float* __restrict__ dmu, float* __restrict__ dls, int M, int D, float temp,
                                   const int* __restrict__ lengths, int Tlen) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * D) return;
  const bool valid = row_valid(lengths, Tlen, (int)(i / D));
  const float gz = (valid && dz) ? dz[i] : 0.f;
  const float gq = (valid && dlog_q) ? dlog_q[i] : 0.f;
  dmu[i] = gz;
  dls[i] = gz * eps[i] * expf(ls[i]) * temp - gq;
}

__global__ void prior_logp_fwd_kernel(const float* __restrict__ mu_ls, long ld, const float* __restrict__ u,
                                      const float* __restrict__ logdet_sum, const float* __restrict__ log_q,
                                      float* __restrict__ log_p, float* __restrict__ kl_rows, int M, int D,
                                      const int* __restrict__ lengths, int Tlen) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const bool valid = row_valid(lengths, Tlen, m);
  const float ld_term = logdet_sum[m] / (float)D;
  float kl = 0.f;
  for (int d = 0; d < D; ++d) {
    const float mu = mu_ls[(long)m * ld + d], ls = mu_ls[(long)m * ld + D + d];
    const float diff = u[(long)m * D + d] - mu;
    float lp = ld_term - ls - HALF_LOG_2PI - 0.5f * (expf(-2.f * ls) * diff * diff);
    lp = valid ? lp : 0.f;
    log_p[(long)m * D + d] = lp;
    kl += (valid ? log_q[(long)m * D + d] : 0.f) - lp;
  }
  kl_rows[m] = kl / (float)D;
}

// total dlog_p = dlog_p_in - dkl_rows / D ; dlog_q = + dkl_rows / D   (masked)
__global__ void prior_logp_bwd_kernel(const float* __restrict__ dlog_p_in, const float* __restrict__ dkl_rows,
                                      const float* __restrict__ mu_ls, long ld, const float* __restrict__ u,
                                      float* __restrict__ dmu_ls, float* __restrict__ du,
                                      float* __restrict__ dlogdet_sum, float* __restrict__ dlog_q, int M, int D,
                                      const int* __restrict__ lengths, int Tlen) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const bool valid = row_valid(lengths, Tlen, m);
  const float gk = (valid && dkl_rows) ? dkl_rows[m] / (float)D : 0.f;
  float gsum = 0.f;
  for (int d = 0; d < D; ++d) {
    float g = -gk;
    if (valid && dlog_p_in) g += dlog_p_in[(long)m * D + d];
    const float mu = mu_ls[(long)m * ld + d], ls = mu_ls[(long)m * ld + D + d];
    const float diff = u[(long)m * D + d] - mu;
    const float w = expf(-2.f * ls);
    dmu_ls[(long)m * 2 * D + d] = g * w * diff;
    dmu_ls[(long)m * 2 * D + D + d] = g * (-1.f + w * diff * diff);
    du[(long)m * D + d] = -g * w * diff;
    if (dlog_q) dlog_q[(long)m * D + d] = gk;
    gsum += g;
  }
  dlogdet_sum[m] = gsum / (float)D;
}

}  // namespace

// ==================================================================== C ABI
namespace {

int check_row_shape(const char* who, int M, int C, int dtype) {
  const int n = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "%s: bad dtype %d", who, dtype);
  VG_REQUIRE(M > 0 && C > 0, "%s: empty input", who);
  VG_REQUIRE(C % n == 0 && C / n <= 64 * MAXV, "%s: C=%d unsupported (multiple of %d, <= %d)", who, C, n,
             64 * MAXV * n);
  return 0;
}

template <typename T>
void run_rmsnorm_fwd(const void* x, const float* scale, void* y, float* rstd, int M, int C, float eps,
                     const int32_t* lengths, int Tn, hipStream_t stream) {
  // round 5, M = 16000 x 1024 bf16 on cold operands (tools/rows_bench.py, one call): 512 / 1024 / 2048 / 4000 blocks
  // 26.9 / 17.8 / 18.1 / 16.0 us -- one row per wave and no loop beats two rows in flight per wave (the backward, with
  // its three row streams and block-level partial sums, is fastest at 512: 31.4 us against 33.3 / 33.9 / 36.0)
  static const int cap = [] { const char* e = getenv("VG_RMSNORM_FWD_BLOCKS"); return e ? atoi(e) : 4096; }();
  const int nb = (M + 3) / 4 < cap ? (M + 3) / 4 : cap;
  const int nv = (C / Vec<T>::N + 63) / 64;      // 16-byte vectors per lane and row
  auto k = nv <= 1 ? rmsnorm_fwd_kernel<T, 1> : nv <= 2 ? rmsnorm_fwd_kernel<T, 2> : rmsnorm_fwd_kernel<T, MAXV>;
  k<<<dim3(nb), dim3(256), 0, stream>>>((const T*)x, scale, (T*)y, rstd, M, C, eps, lengths, Tn);
}
template <typename T>
void run_rmsnorm_bwd(int nb, const void* dy, const void* x, const float* scale, const float* rstd,
                     const void* dx_add, void* dx, float* dsp, float* csp, int M, int C, const int32_t* lengths, int Tn,
                     hipStream_t stream) {
  const int nv = (C / Vec<T>::N + 63) / 64;
  if (csp != nullptr && nv <= 2) {      // column sums of dx on the side (the widths of the Transformer stack)
    auto k = nv <= 1 ? rmsnorm_bwd_kernel<T, 1, true> : rmsnorm_bwd_kernel<T, 2, true>;
    k<<<dim3(nb), dim3(256), 0, stream>>>((const T*)dy, (const T*)x, scale, rstd, (const T*)dx_add, (T*)dx, dsp, csp, M, C,
                                          lengths, Tn);
    return;
  }
  auto k = nv <= 1 ? rmsnorm_bwd_kernel<T, 1, false> : nv <= 2 ? rmsnorm_bwd_kernel<T, 2, false> : rmsnorm_bwd_kernel<T, MAXV, false>;
  k<<<dim3(nb), dim3(256), 0, stream>>>((const T*)dy, (const T*)x, scale, rstd, (const T*)dx_add, (T*)dx, dsp, nullptr, M, C,
                                        lengths, Tn);
}
template <typename T>
void run_colsum(dim3 grid, const void* x, int M, int N, long ld, float* out, hipStream_t stream) {
  colsum_kernel<T><<<grid, dim3(64, 4), 0, stream>>>((const T*)x, M, N, ld, out);
}
template <typename T>
void run_ce_fwd(const void* logits, const int64_t* targets, float* loss_rows, float* lse, int32_t* argmax, int M,
                int V, long ld, const int32_t* lengths, int Tn, hipStream_t stream) {
  ce_fwd_kernel<T><<<dim3((M + 3) / 4), dim3(256), 0, stream>>>((const T*)logits, (const long*)targets, loss_rows,
                                                                lse, argmax, M, V, ld, lengths, Tn);
}
template <typename T>
void run_ce_bwd(const void* logits, const int64_t* targets, const float* lse, const float* gscale, void* dlogits,
                int M, int V, long ld, const int32_t* lengths, int Tn, hipStream_t stream) {
  ce_bwd_kernel<T><<<dim3((M + 3) / 4), dim3(256), 0, stream>>>((const T*)logits, (const long*)targets, lse, gscale,
                                                                (T*)dlogits, M, V, ld, lengths, Tn);
}

}  // namespace

extern "C" int vg_rmsnorm_fwd(const void* x, const float* scale, void* y, float* rstd, int M, int C, float eps,
                              const int32_t* lengths, int T, int dtype, hipStream_t stream) {
  if (int e = check_row_shape("vg_rmsnorm_fwd", M, C, dtype)) return e;
  const int Tn = T > 0 ? T : 1;
  const double esz = dtype == VG_BF16 ? 2.0 : 4.0;       // read x, write y (+ 4 B of rstd per row)
  const int tok = vg_host::prof_begin(VG_PROF_RMSNORM_FWD, (double)M * (2.0 * C * esz + 4.0), stream);
  if (dtype == VG_BF16) run_rmsnorm_fwd<bf16_t>(x, scale, y, rstd, M, C, eps, lengths, Tn, stream);
  else run_rmsnorm_fwd<float>(x, scale, y, rstd, M, C, eps, lengths, Tn, stream);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_rmsnorm_fwd");
}

extern "C" int vg_rmsnorm_bwd_blocks(int M) {
  const int b = (M + 3) / 4;
  static const int cap = [] { const char* e = getenv("VG_RMSNORM_BWD_BLOCKS"); return e ? atoi(e) : 512; }();
  return b < cap ? b : cap;
}

static int rmsnorm_bwd_any(const char* who, const void* dy, const void* x, const float* scale, const float* rstd,
                           const void* dx_add, void* dx, float* dscale_partial, float* dx_colsum_partial, int M, int C,
                           const int32_t* lengths, int T, int dtype, hipStream_t stream) {
  if (int e = check_row_shape(who, M, C, dtype)) return e;
  const int nb = vg_rmsnorm_bwd_blocks(M), Tn = T > 0 ? T : 1;
  const double esz = dtype == VG_BF16 ? 2.0 : 4.0;       // read dy, x (+ the residual-path gradient), write dx
  const int tok = vg_host::prof_begin(VG_PROF_RMSNORM_BWD, (double)M * ((dx_add ? 4.0 : 3.0) * C * esz + 4.0), stream);
  if (dtype == VG_BF16)
    run_rmsnorm_bwd<bf16_t>(nb, dy, x, scale, rstd, dx_add, dx, dscale_partial, dx_colsum_partial, M, C, lengths, Tn, stream);
  else
    run_rmsnorm_bwd<float>(nb, dy, x, scale, rstd, dx_add, dx, dscale_partial, dx_colsum_partial, M, C, lengths, Tn, stream);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch(who);
}

extern "C" int vg_rmsnorm_bwd(const void* dy, const void* x, const float* scale, const float* rstd,
                              const void* dx_add, void* dx, float* dscale_partial, int M, int C,
                              const int32_t* lengths, int T, int dtype, hipStream_t stream) {
  return rmsnorm_bwd_any("vg_rmsnorm_bwd", dy, x, scale, rstd, dx_add, dx, dscale_partial, nullptr, M, C, lengths, T, dtype,
                         stream);
}

extern "C" int vg_rmsnorm_bwd_colsum(const void* dy, const void* x, const float* scale, const float* rstd,
                                     const void* dx_add, void* dx, float* dscale_partial, float* dx_colsum_partial, int M,
                                     int C, const int32_t* lengths, int T, int dtype, hipStream_t stream) {
  VG_REQUIRE(dx_colsum_partial == nullptr || C / (dtype == VG_BF16 ? 8 : 4) <= 128,
             "vg_rmsnorm_bwd_colsum: the column sums of dx are produced for rows of at most 128 vectors (C=%d)", C);
  return rmsnorm_bwd_any("vg_rmsnorm_bwd_colsum", dy, x, scale, rstd, dx_add, dx, dscale_partial, dx_colsum_partial, M, C,
                         lengths, T, dtype, stream);
}

namespace vg_host {
void colsum_accumulate(const void* x, int M, int N, long ld, float* out, int dtype, hipStream_t stream) {
  dim3 gridf((N + 63) / 64), blockf(64, 16);
  if (dtype == VG_BF16) colsum_final_kernel<bf16_t><<<gridf, blockf, 0, stream>>>((const bf16_t*)x, M, N, ld, out, 1);
  else colsum_final_kernel<float><<<gridf, blockf, 0, stream>>>((const float*)x, M, N, ld, out, 1);
}
}  // namespace vg_host

extern "C" int vg_colsum_blocks(int M) {
  if (M <= 2048) return 1;            // small inputs (partial-sum arrays): one pass, one launch
  const int b = (M + 31) / 32;
  return b < 128 ? b : 128;
}

namespace {
void launch_colsum_small(const float* x, int M, int N, long ld, float* out, int accumulate, hipStream_t stream) {
  if (ld % 4 == 0 && ((uintptr_t)x % 16) == 0)
    colsum_small_f32_kernel<<<dim3((N + 15) / 16), dim3(256), 0, stream>>>(x, M, N, ld, out, accumulate);
  else
    colsum_final_kernel<float><<<dim3((N + 63) / 64), dim3(64, 16), 0, stream>>>(x, M, N, ld, out, accumulate);
}
}  // namespace

extern "C" int vg_colsum(const void* x, int M, int N, int64_t ld, float* ws, float* out, int dtype,
                         int accumulate, hipStream_t stream) {
  VG_REQUIRE(N % 4 == 0 && M > 0, "vg_colsum: N=%d must be a multiple of 4", N);
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_colsum: bad dtype %d", dtype);
  const int nb = vg_colsum_blocks(M);
  dim3 gridf((N + 63) / 64), blockf(64, 16);
  if (nb == 1) {
    if (dtype == VG_BF16)
      colsum_final_kernel<bf16_t><<<gridf, blockf, 0, stream>>>((const bf16_t*)x, M, N, (long)ld, out, accumulate);
    else
      launch_colsum_small((const float*)x, M, N, (long)ld, out, accumulate, stream);
  } else {
    const int vec = dtype == VG_BF16 ? 8 : 4;
    if (N % vec == 0 && ld % vec == 0 && ((uintptr_t)x % 16) == 0) {
      dim3 gridv((N + 64 * vec - 1) / (64 * vec), nb), blockv(64, 4);
      if (dtype == VG_BF16) colsum_vec_kernel<bf16_t><<<gridv, blockv, 0, stream>>>((const bf16_t*)x, M, N, (long)ld, ws);
      else colsum_vec_kernel<float><<<gridv, blockv, 0, stream>>>((const float*)x, M, N, (long)ld, ws);
    } else {
      dim3 grid1((N + 255) / 256, nb);
      if (dtype == VG_BF16) run_colsum<bf16_t>(grid1, x, M, N, (long)ld, ws, stream);
      else run_colsum<float>(grid1, x, M, N, (long)ld, ws, stream);
    }
    launch_colsum_small(ws, nb, N, (long)N, out, accumulate, stream);
  }
  return vg_host::check_launch("vg_colsum");
}

// =====================================================================================
// Masked means of several small fp32 row tensors in ONE launch (round 4): out[k] = sum over valid frames m and
// columns c of f_k(x_k[m][c]) / (cols_k * number of valid frames), f = identity or |.|.  These are the step's
// monitors -- TensorMask.mean() of the prior / posterior mean and log-std, |posterior mean|, log p, log q
// (utils/tensormask.py:135-140 called from models/speech/lvtr.py:210-224 and trainers/speech/lvtr.py:131-145 of the
// reference) -- seven reductions that cost five tiny stock launches each (where, div, sum, sum of lengths, div).
// Two launches: up to 64 blocks (one frame per thread and pass) leave per-block sums, one wave folds them and divides --
// deterministic, no atomics (a first version as ONE block of 1024 threads took ~100 us: 16 dependent passes of
// latency-bound row reads on one CU).
// =====================================================================================
namespace {
struct MeanTasks {
  vg_mean_task t[VG_MEAN_MAX_TASKS];
  int n;
};
constexpr int MEAN_SLOTS = VG_MEAN_MAX_TASKS + 1;          // the tasks' sums + the valid-frame count
__global__ __launch_bounds__(256) void masked_means_partial_kernel(MeanTasks tk, int M, const int* __restrict__ lengths,
                                                                   int T, float* __restrict__ partial) {
  __shared__ float red[4][MEAN_SLOTS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[MEAN_SLOTS];
#pragma unroll
  for (int k = 0; k < MEAN_SLOTS; ++k) acc[k] = 0.f;
  for (int m = blockIdx.x * 256 + tid; m < M; m += gridDim.x * 256) {
    if (!row_valid(lengths, T, m)) continue;
    acc[VG_MEAN_MAX_TASKS] += 1.0f;
#pragma unroll
    for (int k = 0; k < VG_MEAN_MAX_TASKS; ++k) {
      if (k < tk.n) {
        const float* row = tk.t[k].src + (long)m * tk.t[k].ld;
        float a = 0.f;
        for (int c = 0; c < tk.t[k].cols; ++c) a += tk.t[k].absolute ? fabsf(row[c]) : row[c];
        acc[k] += a;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MEAN_SLOTS; ++k) {
    const float w = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = w;
  }
  __syncthreads();
  if (tid < MEAN_SLOTS) partial[blockIdx.x * MEAN_SLOTS + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}
__global__ __launch_bounds__(64) void masked_means_final_kernel(MeanTasks tk, const float* __restrict__ partial, int nblocks,
                                                                float* __restrict__ out) {
  const int lane = threadIdx.x;
  float frames = 0.f;
  for (int b = 0; b < nblocks; ++b) frames += partial[b * MEAN_SLOTS + VG_MEAN_MAX_TASKS];
  if (lane < tk.n) {
    float t = 0.f;
    for (int b = 0; b < nblocks; ++b) t += partial[b * MEAN_SLOTS + lane];
    out[lane] = t / ((float)tk.t[lane].cols * frames);
  }
}
}  // namespace

extern "C" int vg_masked_means_blocks(int M) {
  const int b = (M + 255) / 256;
  return b < 64 ? (b < 1 ? 1 : b) : 64;
}

extern "C" int vg_masked_means(const vg_mean_task* tasks, int n, int M, const int32_t* lengths, int T, float* partial,
                               float* out, hipStream_t stream) {
  VG_REQUIRE(tasks != nullptr && n >= 1 && n <= VG_MEAN_MAX_TASKS && M > 0 && out != nullptr && partial != nullptr,
             "vg_masked_means: n=%d M=%d", n, M);
  MeanTasks tk;
  tk.n = n;
  for (int i = 0; i < n; ++i) {
    VG_REQUIRE(tasks[i].src != nullptr && tasks[i].cols > 0 && tasks[i].cols <= 64 && tasks[i].ld >= tasks[i].cols,
               "vg_masked_means: task %d: cols=%d ld=%ld", i, tasks[i].cols, (long)tasks[i].ld);
    tk.t[i] = tasks[i];
  }
  const int nb = vg_masked_means_blocks(M);
  masked_means_partial_kernel<<<dim3(nb), dim3(256), 0, stream>>>(tk, M, lengths, T > 0 ? T : 1, partial);
  masked_means_final_kernel<<<dim3(1), dim3(64), 0, stream>>>(tk, partial, nb, out);
  return vg_host::check_launch("vg_masked_means");
}

extern "C" int vg_sum_f32(const float* x, int64_t n, float* out, hipStream_t stream) {
  sum_kernel<<<dim3(1), dim3(1024), 0, stream>>>(x, (long)n, out);
  return vg_host::check_launch("vg_sum_f32");
}

extern "C" int vg_cast_f32_to_bf16(const float* src, void* dst, int64_t n, hipStream_t stream) {
  VG_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "vg_cast: unaligned");
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  cast_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(src, (bf16_t*)dst, (long)n);
  return vg_host::check_launch("vg_cast_f32_to_bf16");
}

extern "C" int vg_act_bwd(const void* dy, const void* aux, void* dx, int64_t n, int act, int dtype,
                          hipStream_t stream) {
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_act_bwd: bad dtype %d", dtype);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (dtype == VG_BF16)
    act_bwd_kernel<bf16_t><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>((const bf16_t*)dy, (const bf16_t*)aux,
                                                                             (bf16_t*)dx, (long)n, act);
  else
    act_bwd_kernel<float><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>((const float*)dy, (const float*)aux,
                                                                            (float*)dx, (long)n, act);
  return vg_host::check_launch("vg_act_bwd");
}

extern "C" int vg_mask_rows(const void* x, void* y, int M, int C, const int32_t* lengths, int T, int dtype,
                            hipStream_t stream) {
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_mask_rows: bad dtype %d", dtype);
  const long n = (long)M * C;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (dtype == VG_BF16)
    mask_rows_kernel<bf16_t><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>((const bf16_t*)x, (bf16_t*)y, n, C,
                                                                               lengths, T > 0 ? T : 1);
  else
    mask_rows_kernel<float><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>((const float*)x, (float*)y, n, C,
                                                                              lengths, T > 0 ? T : 1);
  return vg_host::check_launch("vg_mask_rows");
}

extern "C" int vg_gather_rows(const void* src, const int32_t* map, void* dst, int n_dst, int row_bytes, hipStream_t stream) {
  VG_REQUIRE(n_dst > 0 && row_bytes > 0 && row_bytes % 16 == 0, "vg_gather_rows: %d rows of %d bytes (rows must be whole 16-byte vectors)",
             n_dst, row_bytes);
  VG_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "vg_gather_rows: unaligned");
  const int vpr = row_bytes / 16;
  long blocks = ((long)n_dst * vpr + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  gather_rows_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>((const uint4*)src, map, (uint4*)dst, n_dst, vpr);
  return vg_host::check_launch("vg_gather_rows");
}

extern "C" int vg_ce_fwd(const void* logits, const int64_t* targets, float* loss_rows, float* lse, int32_t* argmax,
                         int M, int V, int64_t ld, const int32_t* lengths, int T, int dtype, hipStream_t stream) {
  VG_REQUIRE(M > 0 && V > 0, "vg_ce_fwd: empty");
  const int Tn = T > 0 ? T : 1;
  if (dtype == VG_BF16) run_ce_fwd<bf16_t>(logits, targets, loss_rows, lse, argmax, M, V, (long)ld, lengths, Tn, stream);
  else run_ce_fwd<float>(logits, targets, loss_rows, lse, argmax, M, V, (long)ld, lengths, Tn, stream);
  return vg_host::check_launch("vg_ce_fwd");
}

extern "C" int vg_ce_bwd(const void* logits, const int64_t* targets, const float* lse, const float* gscale,
                         void* dlogits, int M, int V, int64_t ld, const int32_t* lengths, int T, int dtype,
                         hipStream_t stream) {
  VG_REQUIRE(M > 0 && V > 0, "vg_ce_bwd: empty");
  const int Tn = T > 0 ? T : 1;
  if (dtype == VG_BF16) run_ce_bwd<bf16_t>(logits, targets, lse, gscale, dlogits, M, V, (long)ld, lengths, Tn, stream);
  else run_ce_bwd<float>(logits, targets, lse, gscale, dlogits, M, V, (long)ld, lengths, Tn, stream);
  return vg_host::check_launch("vg_ce_bwd");
}

extern "C" int vg_reparam_fwd(const float* mu, const float* logstd, const float* eps, float* z, float* log_q, int M,
                              int D, float temperature, const int32_t* lengths, int T, hipStream_t stream) {
  const long n = (long)M * D;
  reparam_fwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(mu, logstd, eps, z, log_q, M, D,
                                                                                  temperature, lengths, T > 0 ? T : 1);
  return vg_host::check_launch("vg_reparam_fwd");
}

extern "C" int vg_reparam_bwd(const float* dz, const float* dlog_q, const float* logstd, const float* eps,
                              float* dmu, float* dlogstd, int M, int D, float temperature, const int32_t* lengths,
                              int T, hipStream_t stream) {
  const long n = (long)M * D;
  reparam_bwd_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(
      dz, dlog_q, logstd, eps, dmu, dlogstd, M, D, temperature, lengths, T > 0 ? T : 1);
  return vg_host::check_launch("vg_reparam_bwd");
}

extern "C" int vg_prior_logp_fwd(const float* mu_ls, int64_t ld_mu_ls, const float* u, const float* logdet_sum,
                                 const float* log_q, float* log_p, float* kl_rows, int M, int D,
                                 const int32_t* lengths, int T, hipStream_t stream) {
  prior_logp_fwd_kernel<<<dim3((M + 255) / 256), dim3(256), 0, stream>>>(mu_ls, (long)ld_mu_ls, u, logdet_sum, log_q,
                                                                         log_p, kl_rows, M, D, lengths, T > 0 ? T : 1);
  return vg_host::check_launch("vg_prior_logp_fwd");
}

extern "C" int vg_prior_logp_bwd(const float* dlog_p, const float* dkl_rows, const float* mu_ls, int64_t ld_mu_ls,
                                 const float* u, float* dmu_ls, float* du, float* dlogdet_sum, float* dlog_q, int M,
                                 int D, const int32_t* lengths, int T, hipStream_t stream) {
  prior_logp_bwd_kernel<<<dim3((M + 255) / 256), dim3(256), 0, stream>>>(
      dlog_p, dkl_rows, mu_ls, (long)ld_mu_ls, u, dmu_ls, du, dlogdet_sum, dlog_q, M, D, lengths, T > 0 ? T : 1);
  return vg_host::check_launch("vg_prior_logp_bwd");
}

// =====================================================================================
// Training-side input fusion: out[m] = mask(E[id[m]]) + relu(Wf z[m] + bf)
// (Embedding.forward modules/linear/layers.py:150-152, token_fuser Linear + ReLU :184-193 -- NOT masked --,
// LVTR.fuse_inputs models/speech/lvtr.py:390-392).  One wave per frame, lane-strided over the embedding width.
// Backward: dE[id] += mask(dout) by fp32 atomics (200 x 64 table, L2-resident), dz[m] = Wf^T g with
// g = dout * (pre > 0), and per-block partial sums of dWf = g^T z and dbf = sum g that a second pass folds
// (deterministic for Wf / bf; the embedding rows add in arrival order).
// =====================================================================================
namespace {
constexpr int EF_ROWS = 64;        // frames per block (4 waves x 16)
constexpr int EF_MAXD = 8;

__global__ __launch_bounds__(256) void embed_fuse_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ z,
                                                             long ldz, const float* __restrict__ emb, int vocab, int E,
                                                             const float* __restrict__ wf, const float* __restrict__ bf,
                                                             int D, const int* __restrict__ lengths, int T,
                                                             float* __restrict__ out, int M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < EF_ROWS; r += 4) {
    const int m = blockIdx.x * EF_ROWS + r;
    if (m >= M) return;
    const bool valid = row_valid(lengths, T, m);
    const long id = min(max(ids[m], 0L), (long)vocab - 1);
    float zv[EF_MAXD];
#pragma unroll
    for (int d = 0; d < EF_MAXD; ++d) zv[d] = d < D ? z[(long)m * ldz + d] : 0.f;
    for (int c = lane; c < E; c += 64) {
      float a = bf ? bf[c] : 0.f;
#pragma unroll
      for (int d = 0; d < EF_MAXD; ++d)
        if (d < D) a = fmaf(wf[(long)c * D + d], zv[d], a);
      out[(long)m * E + c] = (valid ? emb[id * E + c] : 0.f) + fmaxf(a, 0.f);
    }
  }
}

// part[block][E][D + 1]: columns 0..D-1 = dWf rows, column D = dbf
__global__ __launch_bounds__(256) void embed_fuse_bwd_kernel(const float* __restrict__ dout, const long* __restrict__ ids,
                                                             const float* __restrict__ z, long ldz, int vocab, int E,
                                                             const float* __restrict__ wf, const float* __restrict__ bf,
                                                             int D, const int* __restrict__ lengths, int T,
                                                             float* __restrict__ demb, float* __restrict__ dz, long lddz,
                                                             float* __restrict__ part, int M) {
  __shared__ float acc[4][64][EF_MAXD + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c0 = 0; c0 < E; c0 += 64) {
    const int c = c0 + lane;
    float w[EF_MAXD], pw[EF_MAXD + 1];
#pragma unroll
    for (int d = 0; d < EF_MAXD; ++d) w[d] = (c < E && d < D) ? wf[(long)c * D + d] : 0.f;
#pragma unroll
    for (int d = 0; d <= EF_MAXD; ++d) pw[d] = 0.f;
    const float b = (c < E && bf) ? bf[c] : 0.f;
    for (int r = wave; r < EF_ROWS; r += 4) {
      const int m = blockIdx.x * EF_ROWS + r;
      if (m >= M) break;
      float zv[EF_MAXD];
#pragma unroll
      for (int d = 0; d < EF_MAXD; ++d) zv[d] = d < D ? z[(long)m * ldz + d] : 0.f;
      float pre = b;
#pragma unroll
      for (int d = 0; d < EF_MAXD; ++d) pre = fmaf(w[d], zv[d], pre);
      const float go = c < E ? dout[(long)m * E + c] : 0.f;
      if (c < E && demb != nullptr && go != 0.f && row_valid(lengths, T, m)) {
        const long id = min(max(ids[m], 0L), (long)vocab - 1);
        atomicAdd(demb + id * E + c, go);
      }
      const float g = pre > 0.f ? go : 0.f;
#pragma unroll
      for (int d = 0; d < EF_MAXD; ++d) {
        if (d < D) {
          pw[d] = fmaf(g, zv[d], pw[d]);
          const float t = wave_sum(g * w[d]);           // dz[m][d] = sum_c g[c] Wf[c][d]
          if (lane == 0 && dz != nullptr) {
            if (c0 == 0) dz[(long)m * lddz + d] = t;
            else dz[(long)m * lddz + d] += t;
          }
        }
      }
      pw[EF_MAXD] += g;
    }
#pragma unroll
    for (int d = 0; d <= EF_MAXD; ++d) acc[wave][lane][d] = pw[d];
    __syncthreads();
    if (wave == 0 && c < E) {
      for (int d = 0; d < D; ++d)
        part[((long)blockIdx.x * E + c) * (D + 1) + d] = acc[0][lane][d] + acc[1][lane][d] + acc[2][lane][d] + acc[3][lane][d];
      part[((long)blockIdx.x * E + c) * (D + 1) + D] =
          acc[0][lane][EF_MAXD] + acc[1][lane][EF_MAXD] + acc[2][lane][EF_MAXD] + acc[3][lane][EF_MAXD];
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int vg_embed_fuse_blocks(int M) { return (M + EF_ROWS - 1) / EF_ROWS; }

extern "C" int vg_embed_fuse_fwd(const int64_t* ids, const float* z, int64_t ldz, const float* emb, int vocab, int E,
                                 const float* wf, const float* bf, int D, const int32_t* lengths, int T, float* out,
                                 int M, hipStream_t stream) {
  VG_REQUIRE(M > 0 && E > 0 && vocab > 0 && D >= 1 && D <= EF_MAXD, "vg_embed_fuse_fwd: M=%d E=%d D=%d", M, E, D);
  embed_fuse_fwd_kernel<<<dim3((M + EF_ROWS - 1) / EF_ROWS), dim3(256), 0, stream>>>(
      reinterpret_cast<const long*>(ids), z, ldz, emb, vocab, E, wf, bf, D, lengths, T > 0 ? T : 1, out, M);
  return vg_host::check_launch("vg_embed_fuse_fwd");
}

extern "C" int vg_embed_fuse_bwd(const float* dout, const int64_t* ids, const float* z, int64_t ldz, int vocab, int E,
                                 const float* wf, const float* bf, int D, const int32_t* lengths, int T, float* demb,
                                 float* dz, int64_t lddz, float* part, int M, hipStream_t stream) {
  VG_REQUIRE(M > 0 && E > 0 && vocab > 0 && D >= 1 && D <= EF_MAXD && part != nullptr,
             "vg_embed_fuse_bwd: M=%d E=%d D=%d", M, E, D);
  embed_fuse_bwd_kernel<<<dim3((M + EF_ROWS - 1) / EF_ROWS), dim3(256), 0, stream>>>(
      dout, reinterpret_cast<const long*>(ids), z, ldz, vocab, E, wf, bf, D, lengths, T > 0 ? T : 1, demb, dz, lddz,
      part, M);
  return vg_host::check_launch("vg_embed_fuse_bwd");
}

// =====================================================================================
// Diffusion-decoder loss arithmetic (GaussianDiffusion1D.q_sample / p_losses,
// modules/diffusion/ddpm.py:337-366; masked L1 of training_lib/losses.py:9-27,44-57):
//   q_sample : x_t[m][c] = mask( a[t_b] x0[m][c] + s[t_b] noise[m][c] ),  b = m / T
//   l1 rows  : loss_row[m] = mask( mean_c |pred[m][c] - target[m][c]| )            (summed by vg_sum_f32)
//   l1 bwd   : dpred[m][c] = mask( g sign(pred - target) / C )
// One wave per frame, lane-strided over the channels.
// =====================================================================================
namespace {
__global__ __launch_bounds__(256) void qsample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                      const float* __restrict__ ca, const float* __restrict__ cs,
                                                      const long* __restrict__ t, const int* __restrict__ lengths, int T,
                                                      float* __restrict__ out, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const bool valid = row_valid(lengths, T, m);
  const long tb = t[m / T];
  const float a = ca[tb], sgm = cs[tb];
  for (int c = lane; c < C; c += 64) {
    const long i = (long)m * C + c;
    out[i] = valid ? fmaf(a, x0[i], sgm * noise[i]) : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void l1_rows_fwd_kernel(const T* __restrict__ pred, const float* __restrict__ target,
                                                          const int* __restrict__ lengths, int Tn,
                                                          float* __restrict__ rows, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float acc = 0.f;
  if (row_valid(lengths, Tn, m))
    for (int c = lane; c < C; c += 64) acc += fabsf(to_f32<T>(pred[(long)m * C + c]) - target[(long)m * C + c]);
  acc = wave_sum(acc);
  if (lane == 0) rows[m] = acc / (float)C;
}

template <typename T>
__global__ __launch_bounds__(256) void l1_rows_bwd_kernel(const T* __restrict__ pred, const float* __restrict__ target,
                                                          const float* __restrict__ gscale,
                                                          const int* __restrict__ lengths, int Tn,
                                                          T* __restrict__ dpred, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const bool valid = row_valid(lengths, Tn, m);
  const float g = gscale[0] / (float)C;
  for (int c = lane; c < C; c += 64) {
    const long i = (long)m * C + c;
    const float d = to_f32<T>(pred[i]) - target[i];
    dpred[i] = from_f32<T>(valid ? (d > 0.f ? g : (d < 0.f ? -g : 0.f)) : 0.f);
  }
}
}  // namespace

extern "C" int vg_qsample(const float* x0, const float* noise, const float* coef_x0, const float* coef_noise,
                          const int64_t* t, const int32_t* lengths, int T, float* out, int M, int C,
                          hipStream_t stream) {
  VG_REQUIRE(M > 0 && C > 0 && T > 0 && M % T == 0, "vg_qsample: M=%d C=%d T=%d", M, C, T);
  qsample_kernel<<<dim3((M + 3) / 4), dim3(256), 0, stream>>>(x0, noise, coef_x0, coef_noise,
                                                              reinterpret_cast<const long*>(t), lengths, T, out, M, C);
  return vg_host::check_launch("vg_qsample");
}

extern "C" int vg_l1_rows_fwd(const void* pred, const float* target, const int32_t* lengths, int T, float* rows, int M,
                              int C, int dtype, hipStream_t stream) {
  VG_REQUIRE(M > 0 && C > 0, "vg_l1_rows_fwd: M=%d C=%d", M, C);
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_l1_rows_fwd: bad dtype %d", dtype);
  const int Tn = T > 0 ? T : 1;
  if (dtype == VG_BF16)
    l1_rows_fwd_kernel<bf16_t><<<dim3((M + 3) / 4), dim3(256), 0, stream>>>((const bf16_t*)pred, target, lengths, Tn, rows, M, C);
  else
    l1_rows_fwd_kernel<float><<<dim3((M + 3) / 4), dim3(256), 0, stream>>>((const float*)pred, target, lengths, Tn, rows, M, C);
  return vg_host::check_launch("vg_l1_rows_fwd");
}

extern "C" int vg_l1_rows_bwd(const void* pred, const float* target, const float* gscale, const int32_t* lengths, int T,
                              void* dpred, int M, int C, int dtype, hipStream_t stream) {
  VG_REQUIRE(M > 0 && C > 0 && gscale != nullptr, "vg_l1_rows_bwd: M=%d C=%d", M, C);
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_l1_rows_bwd: bad dtype %d", dtype);
  const int Tn = T > 0 ? T : 1;
  if (dtype == VG_BF16)
    l1_rows_bwd_kernel<bf16_t><<<dim3((M + 3) / 4), dim3(256), 0, stream>>>((const bf16_t*)pred, target, gscale, lengths, Tn,
                                                                             (bf16_t*)dpred, M, C);
  else
    l1_rows_bwd_kernel<float><<<dim3((M + 3) / 4), dim3(256), 0, stream>>>((const float*)pred, target, gscale, lengths, Tn,
                                                                           (float*)dpred, M, C);
  return vg_host::check_launch("vg_l1_rows_bwd");
}

// =====================================================================================
// Several small column sums in ONE launch: the second stage of the two-stage reductions (bias / norm-scale
// gradients from per-block partial arrays) is launch-bound at ~4.5 us each and a backward node produces up to
// five of them.  Tasks travel by value in the kernel arguments; blockIdx.y = task, blockIdx.x = 16 columns.
// =====================================================================================
namespace {
struct ColsumTasks {
  vg_colsum_task t[VG_COLSUM_MAX_TASKS];
};

__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumTasks tk) {
  __shared__ float red[64][17];
  const vg_colsum_task& k = tk.t[blockIdx.y];
  const int N = k.cols, M = k.rows;
  if ((int)blockIdx.x * 16 >= N) return;           // uniform per block
  const long ld = k.ld;
  const float* __restrict__ x = k.src;
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c0 = blockIdx.x * 16 + cl * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
#pragma unroll 8
    for (int m = rl; m < M; m += 64) s += *reinterpret_cast<const f32x4*>(x + (long)m * ld + c0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][cl * 4 + e] = s[e];
  __syncthreads();
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  float t = red[4 * g][c] + red[4 * g + 1][c] + red[4 * g + 2][c] + red[4 * g + 3][c];
  __syncthreads();
  red[g][c] = t;
  __syncthreads();
  if (threadIdx.x < 16 && blockIdx.x * 16 + c < N) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v += red[r][c];
    float* o = k.dst + blockIdx.x * 16 + c;
    *o = k.accumulate ? *o + v : v;
  }
}
}  // namespace

namespace {
// first stage of several LARGE column sums in one launch: blockIdx.x walks the 64-lane column blocks of all tasks,
// blockIdx.y the row blocks; task k's partial sums go to dst[blockIdx.y][cols] (fp32)
struct ColsumBigTasks {
  vg_colsum_task t[VG_COLSUM_MAX_TASKS];
  int first_block[VG_COLSUM_MAX_TASKS + 1];
  int n;
};

template <typename T>
__global__ __launch_bounds__(256) void colsum_partials_multi_kernel(ColsumBigTasks tk) {
  constexpr int V = Vec<T>::N;
  __shared__ float red[4][64][V];
  int k = 0;
#pragma unroll
  for (int i = 1; i < VG_COLSUM_MAX_TASKS; ++i)
    if (i < tk.n && (int)blockIdx.x >= tk.first_block[i]) k = i;
  const vg_colsum_task& t = tk.t[k];
  const T* __restrict__ x = reinterpret_cast<const T*>(t.src);
  const int M = t.rows, N = t.cols;
  const long ld = t.ld;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c0 = (((int)blockIdx.x - tk.first_block[k]) * 64 + tx) * V;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    const int step = gridDim.y * 4;
    int m = blockIdx.y * 4 + ty;
    for (; m + step < M; m += 2 * step) {
      float a[8], b[8];
      Vec<T>::load(x + (long)m * ld + c0, a);
      Vec<T>::load(x + (long)(m + step) * ld + c0, b);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += a[e] + b[e];
    }
    if (m < M) {
      float a[8];
      Vec<T>::load(x + (long)m * ld + c0, a);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += a[e];
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[ty][tx][e] = s[e];
  __syncthreads();
  if (ty == 0 && c0 < N) {
#pragma unroll
    for (int e = 0; e < V; ++e)
      t.dst[(long)blockIdx.y * N + c0 + e] = red[0][tx][e] + red[1][tx][e] + red[2][tx][e] + red[3][tx][e];
  }
}
}  // namespace

namespace {
// per-SEGMENT column sums, first stage: x is [nseg * rows][ld]; part[blockIdx.y][seg][cols] = sum over this row
// block's rows of segment seg.  A plain column sum over the nb rows of part (viewed [nb][nseg * cols]) finishes it.
template <typename T>
__global__ __launch_bounds__(256) void colsum_segments_kernel(const T* __restrict__ x, int rows_, int N, long ld,
                                                              float* __restrict__ part, const int* __restrict__ cu) {
  constexpr int V = Vec<T>::N;
  __shared__ float red[4][64][V];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int seg = blockIdx.z, nseg = gridDim.z;
  const int c0 = (blockIdx.x * 64 + tx) * V;
  // cu: segment seg = rows [cu[seg], cu[seg + 1]) (packed ragged sequences); otherwise rows_ rows each
  const int first = cu ? cu[seg] : seg * rows_, rows = cu ? cu[seg + 1] - first : rows_;
  const T* __restrict__ xs = x + (long)first * ld;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    const int step = gridDim.y * 4;
    int m = blockIdx.y * 4 + ty;
    for (; m + step < rows; m += 2 * step) {
      float a[8], b[8];
      Vec<T>::load(xs + (long)m * ld + c0, a);
      Vec<T>::load(xs + (long)(m + step) * ld + c0, b);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += a[e] + b[e];
    }
    if (m < rows) {
      float a[8];
      Vec<T>::load(xs + (long)m * ld + c0, a);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += a[e];
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[ty][tx][e] = s[e];
  __syncthreads();
  if (ty == 0 && c0 < N) {
#pragma unroll
    for (int e = 0; e < V; ++e)
      part[((long)blockIdx.y * nseg + seg) * N + c0 + e] = red[0][tx][e] + red[1][tx][e] + red[2][tx][e] + red[3][tx][e];
  }
}
}  // namespace

extern "C" int vg_colsum_segments(const void* x, int nseg, int rows, int cols, int64_t ld, float* part, int nb, float* out,
                                  int dtype, hipStream_t stream) {
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_colsum_segments: bad dtype %d", dtype);
  const int vec = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(x != nullptr && part != nullptr && out != nullptr && nseg >= 1 && nseg <= 65535 && rows >= 1 && nb >= 1 && nb <= 64 &&
                 cols > 0 && cols % vec == 0 && ld % vec == 0 && ((uintptr_t)x % 16) == 0,
             "vg_colsum_segments: nseg=%d rows=%d cols=%d ld=%ld nb=%d", nseg, rows, cols, (long)ld, nb);
  dim3 grid((cols + 64 * vec - 1) / (64 * vec), nb, nseg), block(64, 4);
  if (dtype == VG_BF16) colsum_segments_kernel<bf16_t><<<grid, block, 0, stream>>>((const bf16_t*)x, rows, cols, (long)ld, part, nullptr);
  else colsum_segments_kernel<float><<<grid, block, 0, stream>>>((const float*)x, rows, cols, (long)ld, part, nullptr);
  launch_colsum_small(part, nb, nseg * cols, (long)nseg * cols, out, 0, stream);
  return vg_host::check_launch("vg_colsum_segments");
}

// the same sums over ragged segments laid end to end: segment s = rows [cu_rows[s], cu_rows[s + 1])
extern "C" int vg_colsum_segments_cu(const void* x, const int* cu_rows, int nseg, int cols, int64_t ld, float* part, int nb,
                                     float* out, int dtype, hipStream_t stream) {
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_colsum_segments_cu: bad dtype %d", dtype);
  const int vec = dtype == VG_BF16 ? 8 : 4;
  VG_REQUIRE(x != nullptr && cu_rows != nullptr && part != nullptr && out != nullptr && nseg >= 1 && nseg <= 65535 && nb >= 1 &&
                 nb <= 64 && cols > 0 && cols % vec == 0 && ld % vec == 0 && ((uintptr_t)x % 16) == 0,
             "vg_colsum_segments_cu: nseg=%d cols=%d ld=%ld nb=%d", nseg, cols, (long)ld, nb);
  dim3 grid((cols + 64 * vec - 1) / (64 * vec), nb, nseg), block(64, 4);
  if (dtype == VG_BF16) colsum_segments_kernel<bf16_t><<<grid, block, 0, stream>>>((const bf16_t*)x, 0, cols, (long)ld, part, cu_rows);
  else colsum_segments_kernel<float><<<grid, block, 0, stream>>>((const float*)x, 0, cols, (long)ld, part, cu_rows);
  launch_colsum_small(part, nb, nseg * cols, (long)nseg * cols, out, 0, stream);
  return vg_host::check_launch("vg_colsum_segments_cu");
}

extern "C" int vg_colsum_partials_multi(const vg_colsum_task* tasks, int n, int nb, int dtype, hipStream_t stream) {
  VG_REQUIRE(tasks != nullptr && n >= 1 && n <= VG_COLSUM_MAX_TASKS, "vg_colsum_partials_multi: n=%d (1..%d)", n,
             VG_COLSUM_MAX_TASKS);
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_colsum_partials_multi: bad dtype %d", dtype);
  VG_REQUIRE(nb >= 1 && nb <= 1024, "vg_colsum_partials_multi: nb=%d", nb);
  const int vec = dtype == VG_BF16 ? 8 : 4;
  ColsumBigTasks tk;
  tk.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const vg_colsum_task& k = tasks[i];
    VG_REQUIRE(k.src != nullptr && k.dst != nullptr && k.rows > 0 && k.cols > 0 && k.cols % vec == 0 && k.ld % vec == 0 &&
                   ((uintptr_t)k.src % 16) == 0,
               "vg_colsum_partials_multi: task %d: rows=%d cols=%d ld=%ld (16-byte aligned rows, cols %% %d == 0)", i,
               k.rows, k.cols, (long)k.ld, vec);
    tk.t[i] = k;
    tk.first_block[i] = blocks;
    blocks += (k.cols + 64 * vec - 1) / (64 * vec);
  }
  for (int i = n; i <= VG_COLSUM_MAX_TASKS; ++i) tk.first_block[i] = blocks;
  dim3 grid(blocks, nb), block(64, 4);
  if (dtype == VG_BF16) colsum_partials_multi_kernel<bf16_t><<<grid, block, 0, stream>>>(tk);
  else colsum_partials_multi_kernel<float><<<grid, block, 0, stream>>>(tk);
  return vg_host::check_launch("vg_colsum_partials_multi");
}

extern "C" int vg_colsum_multi(const vg_colsum_task* tasks, int n, hipStream_t stream) {
  VG_REQUIRE(tasks != nullptr && n >= 1 && n <= VG_COLSUM_MAX_TASKS, "vg_colsum_multi: n=%d (1..%d)", n,
             VG_COLSUM_MAX_TASKS);
  ColsumTasks tk;
  int maxc = 0;
  for (int i = 0; i < n; ++i) {
    const vg_colsum_task& k = tasks[i];
    VG_REQUIRE(k.src != nullptr && k.dst != nullptr && k.rows > 0 && k.cols > 0 && k.cols % 4 == 0 && k.ld % 4 == 0 &&
                   ((uintptr_t)k.src % 16) == 0,
               "vg_colsum_multi: task %d: rows=%d cols=%d ld=%ld (fp32, 16-byte aligned rows, cols %% 4 == 0)", i, k.rows,
               k.cols, (long)k.ld);
    tk.t[i] = k;
    maxc = k.cols > maxc ? k.cols : maxc;
  }
  for (int i = n; i < VG_COLSUM_MAX_TASKS; ++i) tk.t[i] = tasks[0];
  colsum_multi_kernel<<<dim3((maxc + 15) / 16, n), dim3(256), 0, stream>>>(tk);
  return vg_host::check_launch("vg_colsum_multi");
}

// =====================================================================================
// Per-frame channel norm for NARROW rows (the utterance encoder's 128- and 256-channel layers; the wide layers use
// vg_dwnorm_*): y = act(gamma (x - mean) rstd + beta), unbiased variance (reference modules/norm.py:35-47), optional
// fused ReLU (ConvNormAct, modules/conv/layers.py:543-560).  One wave per frame, lane-strided over C <= 1024.
// Backward: dx, and per-block partial sums of (dgamma | dbeta) for vg_colsum_multi.
// =====================================================================================
namespace {
constexpr int CN_ROWS = 16;            // frames per block (4 waves x 4)
constexpr int CN_MAXE = 16;            // elements per lane: C <= 1024

template <typename T>
__global__ __launch_bounds__(256) void chnorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                         int C, float eps, int relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < CN_ROWS; r += 4) {
    const int m = blockIdx.x * CN_ROWS + r;
    if (m >= M) return;
    float v[CN_MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < CN_MAXE; ++e) {
      const int c = lane + 64 * e;
      v[e] = c < C ? to_f32<T>(x[(long)m * C + c]) : 0.f;
      s += v[e];
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < CN_MAXE; ++e) {
      const int c = lane + 64 * e;
      const float d = c < C ? v[e] - mu : 0.f;
      q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / (float)(C > 1 ? C - 1 : 1) + eps);
#pragma unroll
    for (int e = 0; e < CN_MAXE; ++e) {
      const int c = lane + 64 * e;
      if (c < C) {
        float o = fmaf(gamma[c] * rs, v[e] - mu, beta[c]);
        if (relu) o = fmaxf(o, 0.f);
        y[(long)m * C + c] = from_f32<T>(o);
      }
    }
    if (lane == 0) {
      mean[m] = mu;
      rstd[m] = rs;
    }
  }
}

// part[block][2 C]: dgamma | dbeta partial sums of the block's frames
template <typename T>
__global__ __launch_bounds__(256) void chnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                         const T* __restrict__ y, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         T* __restrict__ dx, float* __restrict__ part, int M, int C,
                                                         int relu) {
  __shared__ float acc[4][2][64 * CN_MAXE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pg[CN_MAXE], pb[CN_MAXE];
#pragma unroll
  for (int e = 0; e < CN_MAXE; ++e) pg[e] = pb[e] = 0.f;
  for (int r = wave; r < CN_ROWS; r += 4) {
    const int m = blockIdx.x * CN_ROWS + r;
    if (m >= M) break;
    const float mu = mean[m], rs = rstd[m];
    float g[CN_MAXE], xh[CN_MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < CN_MAXE; ++e) {
      const int c = lane + 64 * e;
      g[e] = xh[e] = 0.f;
      if (c < C) {
        float d = to_f32<T>(dy[(long)m * C + c]);
        if (relu && !(to_f32<T>(y[(long)m * C + c]) > 0.f)) d = 0.f;
        xh[e] = (to_f32<T>(x[(long)m * C + c]) - mu) * rs;
        pg[e] = fmaf(d, xh[e], pg[e]);
        pb[e] += d;
        g[e] = d * gamma[c];
        s1 += g[e];
        s2 = fmaf(g[e], xh[e], s2);
      }
    }
    // y = gamma xh + beta with xh = (x - mu) rs and rs from the UNBIASED variance:
    // dx = rs (g - mean(g) - xh sum(g xh) / (C - 1))
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)(C > 1 ? C - 1 : 1);
#pragma unroll
    for (int e = 0; e < CN_MAXE; ++e) {
      const int c = lane + 64 * e;
      if (c < C) dx[(long)m * C + c] = from_f32<T>(rs * (g[e] - s1 - xh[e] * s2));
    }
  }
#pragma unroll
  for (int e = 0; e < CN_MAXE; ++e) {
    acc[wave][0][lane + 64 * e] = pg[e];
    acc[wave][1][lane + 64 * e] = pb[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    part[(long)blockIdx.x * 2 * C + c] = acc[0][0][c] + acc[1][0][c] + acc[2][0][c] + acc[3][0][c];
    part[(long)blockIdx.x * 2 * C + C + c] = acc[0][1][c] + acc[1][1][c] + acc[2][1][c] + acc[3][1][c];
  }
}
}  // namespace

extern "C" int vg_chnorm_blocks(int M) { return (M + CN_ROWS - 1) / CN_ROWS; }

extern "C" int vg_chnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                             int M, int C, float eps, int relu, int dtype, hipStream_t stream) {
  VG_REQUIRE(M > 0 && C >= 2 && C <= 64 * CN_MAXE, "vg_chnorm_fwd: M=%d C=%d (2..%d)", M, C, 64 * CN_MAXE);
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_chnorm_fwd: bad dtype %d", dtype);
  const dim3 grid((M + CN_ROWS - 1) / CN_ROWS);
  if (dtype == VG_BF16)
    chnorm_fwd_kernel<bf16_t><<<grid, dim3(256), 0, stream>>>((const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, C, eps, relu);
  else
    chnorm_fwd_kernel<float><<<grid, dim3(256), 0, stream>>>((const float*)x, gamma, beta, (float*)y, mean, rstd, M, C, eps, relu);
  return vg_host::check_launch("vg_chnorm_fwd");
}

extern "C" int vg_chnorm_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* mean,
                             const float* rstd, void* dx, float* part, int M, int C, int relu, int dtype,
                             hipStream_t stream) {
  VG_REQUIRE(M > 0 && C >= 2 && C <= 64 * CN_MAXE && part != nullptr, "vg_chnorm_bwd: M=%d C=%d", M, C);
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_chnorm_bwd: bad dtype %d", dtype);
  VG_REQUIRE(!relu || y != nullptr, "vg_chnorm_bwd: the fused ReLU needs the forward output");
  const dim3 grid((M + CN_ROWS - 1) / CN_ROWS);
  if (dtype == VG_BF16)
    chnorm_bwd_kernel<bf16_t><<<grid, dim3(256), 0, stream>>>((const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)y, gamma, mean,
                                                              rstd, (bf16_t*)dx, part, M, C, relu);
  else
    chnorm_bwd_kernel<float><<<grid, dim3(256), 0, stream>>>((const float*)dy, (const float*)x, (const float*)y, gamma, mean, rstd,
                                                             (float*)dx, part, M, C, relu);
  return vg_host::check_launch("vg_chnorm_bwd");
}

// =====================================================================================
// Window gather of a strided 1-D convolution on channels-last rows (ConvNormAct of the utterance encoder,
// modules/conv/layers.py:543-560: Conv1d(k, stride) = gather + one GEMM):
//   rows[b][to][tap][c] = x[b][to * stride + tap - pad_left][c]   (0 outside the sequence)
// and its adjoint in gather form (no atomics):
//   dx[b][t][c] = sum over taps with (t + pad_left - tap) % stride == 0 of drows[b][(t + pad_left - tap) / stride][tap][c]
// =====================================================================================
namespace {
template <typename T>
__global__ __launch_bounds__(256) void conv_gather_kernel(const T* __restrict__ x, T* __restrict__ rows, int B, int Tn,
                                                          int C, int t_out, int k, int stride, int pl) {
  const long total = (long)B * t_out * k * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long r = i / C;
    const int tap = (int)(r % k);
    r /= k;
    const int to = (int)(r % t_out), b = (int)(r / t_out);
    const int t = to * stride + tap - pl;
    rows[i] = (t >= 0 && t < Tn) ? x[((long)b * Tn + t) * C + c] : from_f32<T>(0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_scatter_kernel(const T* __restrict__ drows, T* __restrict__ dx, int B, int Tn,
                                                           int C, int t_out, int k, int stride, int pl) {
  const long total = (long)B * Tn * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long r = i / C;
    const int t = (int)(r % Tn), b = (int)(r / Tn);
    float acc = 0.f;
    for (int tap = 0; tap < k; ++tap) {
      const int u = t + pl - tap;
      if (u < 0 || u % stride != 0) continue;
      const int to = u / stride;
      if (to < t_out) acc += to_f32<T>(drows[(((long)b * t_out + to) * k + tap) * C + c]);
    }
    dx[i] = from_f32<T>(acc);
  }
}
}  // namespace

extern "C" int vg_conv_gather(const void* x, void* rows, int B, int T, int C, int t_out, int k, int stride, int pad_left,
                              int dtype, hipStream_t stream) {
  VG_REQUIRE(B > 0 && T > 0 && C > 0 && t_out > 0 && k > 0 && stride > 0 && pad_left >= 0,
             "vg_conv_gather: B=%d T=%d C=%d t_out=%d k=%d stride=%d", B, T, C, t_out, k, stride);
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_conv_gather: bad dtype %d", dtype);
  const long total = (long)B * t_out * k * C;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == VG_BF16)
    conv_gather_kernel<bf16_t><<<dim3(grid), dim3(256), 0, stream>>>((const bf16_t*)x, (bf16_t*)rows, B, T, C, t_out, k, stride, pad_left);
  else
    conv_gather_kernel<float><<<dim3(grid), dim3(256), 0, stream>>>((const float*)x, (float*)rows, B, T, C, t_out, k, stride, pad_left);
  return vg_host::check_launch("vg_conv_gather");
}

extern "C" int vg_conv_scatter(const void* drows, void* dx, int B, int T, int C, int t_out, int k, int stride,
                               int pad_left, int dtype, hipStream_t stream) {
  VG_REQUIRE(B > 0 && T > 0 && C > 0 && t_out > 0 && k > 0 && stride > 0 && pad_left >= 0,
             "vg_conv_scatter: B=%d T=%d C=%d t_out=%d k=%d stride=%d", B, T, C, t_out, k, stride);
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_conv_scatter: bad dtype %d", dtype);
  const long total = (long)B * T * C;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == VG_BF16)
    conv_scatter_kernel<bf16_t><<<dim3(grid), dim3(256), 0, stream>>>((const bf16_t*)drows, (bf16_t*)dx, B, T, C, t_out, k, stride, pad_left);
  else
    conv_scatter_kernel<float><<<dim3(grid), dim3(256), 0, stream>>>((const float*)drows, (float*)dx, B, T, C, t_out, k, stride, pad_left);
  return vg_host::check_launch("vg_conv_scatter");
}
