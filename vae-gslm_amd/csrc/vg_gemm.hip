// MFMA GEMM family for the Transformer / head linears.
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(k,n) )
//
// Operand storage modes (no transposed copies are ever materialised):
//   A_ROW : A stored [M][K]  (k contiguous)      activations, upstream grads
//   A_TR  : A stored [K][M]  (m contiguous)      dY in the weight-gradient GEMM
//   B_ROW : B stored [N][K]  (k contiguous)      nn.Linear weight in forward
//   B_TR  : B stored [K][N]  (n contiguous)      nn.Linear weight in dgrad, X in wgrad
//
//   forward   y  = x  W^T       : A_ROW, B_ROW      (reference: modules/transformer/layers.py:82,
//   dgrad     dx = dy W         : A_ROW, B_TR        modules/attention/attention.py:52,79,
//   wgrad     dW = dy^T x       : A_TR,  B_TR        modules/linear/layers.py:192-193)
//
// Tiling: 128x128 block tile, 4 waves (2x2), each wave a 64x64 tile = 2x2
// MFMA 32x32 tiles; BK = 64 (bf16) / 32 (f32) -> every LDS tile is 16 KiB;
// two LDS stages, register-staged global loads issued one tile ahead of the
// MFMAs that hide them (one barrier per K tile).
#include <stdlib.h>

#include "vg_common.h"
#include "../../include/vaegslm_hip.h"
#include "vg_gemm_params.h"

using namespace vg;

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int TILE_BYTES = 128 * 33 * 4;   // largest image (f32 RowTile 128 x (32+1))

template <typename T> struct BKOf;
template <> struct BKOf<bf16_t> { static constexpr int v = 64; };
template <> struct BKOf<float> { static constexpr int v = 32; };

// ---- global -> register staging of one 16 KiB operand tile (4 x 16 B per thread)
template <typename T, bool TR>
VG_DEVICE void stage_load(uint4 (&r)[4], const T* __restrict__ src, long ld, int rc0, int rc_lim,
                          int k0, int k_lim, int tid) {
  constexpr int VEC = Traits<T>::VEC;
  constexpr int BK = BKOf<T>::v;
  if constexpr (!TR) {
    // tile [128 rows][BK] : thread -> (row = it*32 + tid/8, chunk = tid%8)
    const int c16 = tid & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = rc0 + it * 32 + (tid >> 3);
      const int k = k0 + c16 * VEC;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < rc_lim && k < k_lim) v = *reinterpret_cast<const uint4*>(src + (long)row * ld + k);
      r[it] = v;
    }
  } else {
    // tile [BK krows][128 cols] : 128 cols = 128/VEC chunks per krow
    constexpr int CPR = 128 / VEC;            // chunks per krow (16 bf16 / 32 f32)
    constexpr int RPI = NTHREADS / CPR;       // krows per iteration (16 / 8)
    static_assert(RPI * 4 == BK, "tile shape");
    const int c16 = tid % CPR;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int krow = k0 + it * RPI + tid / CPR;
      const int col = rc0 + c16 * VEC;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (krow < k_lim && col < rc_lim) v = *reinterpret_cast<const uint4*>(src + (long)krow * ld + col);
      r[it] = v;
    }
  }
}

template <typename T, bool TR>
VG_DEVICE void stage_store(const uint4 (&r)[4], char* tile, int tid) {
  constexpr int VEC = Traits<T>::VEC;
  constexpr int BK = BKOf<T>::v;
  if constexpr (!TR) {
#pragma unroll
    for (int it = 0; it < 4; ++it) RowTile<T, BK>::store_vec(tile, it * 32 + (tid >> 3), tid & 7, r[it]);
  } else {
    constexpr int CPR = 128 / VEC;
    constexpr int RPI = NTHREADS / CPR;
#pragma unroll
    for (int it = 0; it < 4; ++it) TrTile<T, 128>::store_vec(tile, it * RPI + tid / CPR, tid % CPR, r[it]);
  }
}

template <typename T, bool A_TR, bool B_TR>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmParams p) {
  typedef Traits<T> Tr;
  typedef typename Tr::Frag Frag;
  constexpr int BK = BKOf<T>::v;
  constexpr int KSTEPS = BK / Tr::KSTEP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // stage c: A image at smem + 2c * TILE_BYTES, B image right behind it
  auto tileA = [&](int c) -> char* { return smem + (2 * c) * TILE_BYTES; };
  auto tileB = [&](int c) -> char* { return smem + (2 * c + 1) * TILE_BYTES; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nkt = (kend - kbeg + BK - 1) / BK;
  const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(p.B);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

  uint4 ra[4], rb[4];
  stage_load<T, A_TR>(ra, A, p.lda, m0, p.M, kbeg, kend, tid);
  stage_load<T, B_TR>(rb, B, p.ldb, n0, p.N, kbeg, kend, tid);
  stage_store<T, A_TR>(ra, tileA(0), tid);
  stage_store<T, B_TR>(rb, tileB(0), tid);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < nkt;
    if (more) {
      stage_load<T, A_TR>(ra, A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, tid);
      stage_load<T, B_TR>(rb, B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, tid);
    }
    const char* ta = tileA(cur);
    const char* tb = tileB(cur);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      Frag fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (A_TR) fa[i] = TrTile<T, 128>::template frag<false>(ta, 0, wm * 64 + i * 32, s, lane);
        else fa[i] = RowTile<T, BK>::frag(ta, wm * 64 + i * 32 + (lane & 31), s, lane);
        if constexpr (B_TR) fb[i] = TrTile<T, 128>::template frag<false>(tb, 0, wn * 64 + i * 32, s, lane);
        else fb[i] = RowTile<T, BK>::frag(tb, wn * 64 + i * 32 + (lane & 31), s, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Tr::mfma(fa[i], fb[j], acc[i][j]);
    }
    if (more) {
      stage_store<T, A_TR>(ra, tileA(cur ^ 1), tid);
      stage_store<T, B_TR>(rb, tileB(cur ^ 1), tid);
    }
    __syncthreads();
  }

  // ------------------------------------------------------------ epilogue
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
  const T* __restrict__ auxi = reinterpret_cast<const T*>(p.aux_in);
  T* __restrict__ auxo = reinterpret_cast<T*>(p.aux_out);
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + acc_row(r, lane);
      if (m >= p.M) continue;
      const bool valid = row_valid(p.lengths, p.T, p.m_base + m);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + acc_col(lane);
        if (n >= p.N) continue;
        const long idx = (long)m * p.ldc + n;
        float v = acc[i][j][r] * p.alpha;
        if (split) {   // partial sums: raw fp32 accumulation, no epilogue
          atomicAdd(reinterpret_cast<float*>(p.C) + idx, v);
          continue;
        }
        if (p.bias) v += p.bias[n];
        if (p.pre_add) v += to_f32<T>(reinterpret_cast<const T*>(p.pre_add)[idx]);
        const int act = p.act & 15;
        if (p.act & VG_ACT_SAVE_DERIV) {
          float d = 1.f;
          if (act == VG_ACT_RELU) d = v > 0.f ? 1.f : 0.f;
          else if (act == VG_ACT_GELU) d = gelu_erf_grad(v);
          else if (act == VG_ACT_SILU) d = silu_grad(v);
          if (auxo) auxo[idx] = from_f32<T>(d);
        } else if (auxo) {
          auxo[idx] = from_f32<T>(v);
        }
        if (act == VG_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == VG_ACT_GELU) v = gelu_erf(v);
        else if (act == VG_ACT_SILU) v = silu(v);
        if (p.dact == VG_ACT_RELU) v = (to_f32<T>(auxi[idx]) > 0.f) ? v : 0.f;
        else if (p.dact == VG_ACT_GELU) v *= gelu_erf_grad(to_f32<T>(auxi[idx]));
        else if (p.dact == VG_ACT_SILU) v *= silu_grad(to_f32<T>(auxi[idx]));
        else if (p.dact == VG_ACT_STORED) v *= to_f32<T>(auxi[idx]);
        if (res) v += to_f32<T>(res[idx]);
        if (!valid) v = 0.f;
        if (p.out_f32) {
          float* c = reinterpret_cast<float*>(p.C) + idx;
          *c = p.accumulate ? (*c + v) : v;
        } else {
          reinterpret_cast<T*>(p.C)[idx] = from_f32<T>(v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ thin products (latent side)
// The model's 4-wide latent makes a few Linears extremely thin (4 -> 64, 512 -> 4, 4 -> 8): 0.01-0.07 GFLOP
// each, for which a 128x128 MFMA tile launch costs 30-40 us (one column of tiles, exact-f32 MFMA at 2 k per
// instruction).  They are plain dot products: one thread per output element, the reduction index walked
// serially (<= 512 long) or, for the weight gradient (reduction over all frames), cut into chunks that meet
// through fp32 atomics.  Element (i, j) = sum_r A[i*sai + r*sar] * B[j*sbj + r*sbr]; same epilogue subset as
// the tile kernels (alpha, bias, ReLU/GELU/SiLU, residual, row mask, accumulate, fp32-or-T output).
template <typename T>
__global__ __launch_bounds__(256) void thin_gemm_kernel(GemmParams p, long sai, long sar, long sbj, long sbr, int R,
                                                        int rchunk) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * p.N) return;
  const int i = (int)(idx / p.N), j = (int)(idx % p.N);
  const T* __restrict__ a = reinterpret_cast<const T*>(p.A) + i * sai;
  const T* __restrict__ b = reinterpret_cast<const T*>(p.B) + j * sbj;
  const int r0 = blockIdx.y * rchunk, r1 = min(R, r0 + rchunk);
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc = fmaf(to_f32<T>(a[r * sar]), to_f32<T>(b[r * sbr]), acc);
  float v = acc * p.alpha;
  const long o = (long)i * p.ldc + j;
  if (gridDim.y > 1) {       // partial sums of a chunked reduction: raw fp32 accumulation
    atomicAdd(reinterpret_cast<float*>(p.C) + o, v);
    return;
  }
  if (p.bias) v += p.bias[j];
  if (p.act == VG_ACT_RELU) v = fmaxf(v, 0.f);
  else if (p.act == VG_ACT_GELU) v = gelu_erf(v);
  else if (p.act == VG_ACT_SILU) v = silu(v);
  if (p.residual) v += to_f32<T>(reinterpret_cast<const T*>(p.residual)[o]);
  if (!row_valid(p.lengths, p.T, p.m_base + i)) v = 0.f;
  if (p.out_f32) {
    float* c = reinterpret_cast<float*>(p.C) + o;
    *c = p.accumulate ? (*c + v) : v;
  } else {
    reinterpret_cast<T*>(p.C)[o] = from_f32<T>(v);
  }
}

// true if the launch was taken by the thin kernel
template <typename T>
bool launch_thin(const GemmParams& p, int a_tr, int b_tr, int splits, hipStream_t stream) {
  if (a_tr && !b_tr) return false;
  if (p.aux_in || p.aux_out || p.pre_add || p.dact != VG_ACT_NONE || (p.act & VG_ACT_SAVE_DERIV) || p.colsum_out ||
      p.colpart)
    return false;
  const bool thin = (p.N <= 16 || p.K <= 16 || (a_tr && p.M <= 16)) && (long)p.M * p.N <= (1L << 24);
  if (!thin) return false;
  if (!a_tr && (long)p.K > 1024) return false;                 // serial reduction per thread: keep it short
  if (splits > 1 && !p.out_f32) return false;
  // strides of element (i, j) / reduction index r in A and B for the three operand modes
  const long sai = a_tr ? 1 : p.lda, sar = a_tr ? p.lda : 1;
  const long sbj = b_tr ? 1 : p.ldb, sbr = b_tr ? p.ldb : 1;
  int chunks = 1, rchunk = p.K;
  if (a_tr) {                                                   // reduction over frames: chunk + atomics
    if (!p.out_f32) return false;
    rchunk = 64;          // few outputs, long reduction: many short chunks keep the whole chip busy
    chunks = (p.K + rchunk - 1) / rchunk;
    if (chunks == 1 && !(splits > 1)) rchunk = p.K;
    else if (!(splits > 1 || p.accumulate)) return false;       // atomics need a destination that already holds a value
  }
  GemmParams q = p;
  if (a_tr && chunks > 1) q.accumulate = 1;
  const long outs = (long)p.M * p.N;
  dim3 grid((unsigned)((outs + 255) / 256), chunks);
  VG_LAUNCH(thin_gemm_kernel<T>, grid, dim3(256), 0, stream, q, sai, sar, sbj, sbr, p.K, rchunk);
  return true;
}

// algorithmic bytes of one product: both operands once, the result once (twice when it accumulates into C), every
// optional M x N stream of the epilogue once
double gemm_algorithmic_bytes(const GemmParams& p, int esz) {
  const double mn = (double)p.M * p.N;
  double b = ((double)p.M * p.K + (double)p.K * p.N) * esz + mn * (p.out_f32 ? 4 : esz) * (p.accumulate ? 2 : 1);
  if (p.residual) b += mn * esz;
  if (p.pre_add) b += mn * esz;
  if (p.aux_in) b += mn * (p.aux_u8 ? 1 : esz);
  if (p.aux_out) b += mn * (p.aux_u8 ? 1 : esz);
  return b;
}

template <typename T>
int launch(const GemmParams& p, int a_tr, int b_tr, int splits, int tile_cfg, hipStream_t stream) {
  const int kind = sizeof(T) == 4 ? VG_PROF_GEMM_F32
                    : (a_tr ? VG_PROF_GEMM_BF16_TN : (b_tr ? VG_PROF_GEMM_BF16_NN : VG_PROF_GEMM_BF16_NT));
  if (sizeof(T) == 2 && tile_cfg > 0) {   // LDS-DMA pipelined variant (vg_gemm_dma.hip)
    static const int sig_debug = [] { const char* e = getenv("VG_DEBUG_GEMM"); return e ? atoi(e) : 0; }();
    if (sig_debug == 2)          // one line per launch: shape, tile configuration and epilogue options (tools/gemm_census.py)
      fprintf(stderr, "[vg_gemm] cfg=%d M=%d N=%d K=%d a_tr=%d b_tr=%d splits=%d act=%d dact=%d bias=%d res=%d pre=%d aux_out=%d "
                      "lens=%d f32=%d acc=%d colpart=%d colsum=%d alpha=%g\n",
              tile_cfg, p.M, p.N, p.K, a_tr, b_tr, splits, p.act, p.dact, p.bias != nullptr, p.residual != nullptr,
              p.pre_add != nullptr, p.aux_out != nullptr, p.lengths != nullptr, p.out_f32, p.accumulate,
              p.colpart != nullptr, p.colsum_out != nullptr, (double)p.alpha);
    const int tok = vg_host::prof_begin(kind, 2.0 * p.M * p.N * p.K, stream, gemm_algorithmic_bytes(p, sizeof(T)));
    int rc = -1;
    hipEvent_t lab0 = nullptr, lab1 = nullptr;
    if (sig_debug == 3) {        // lab: one line per launch WITH its duration (synchronises: eager launches only)
      hipEventCreate(&lab0);
      hipEventCreate(&lab1);
      hipEventRecord(lab0, stream);
    }
    if (tile_cfg >= 10) {     // phase-pipelined 256x256 tile (vg_gemm_ph.hip); shapes it does not take run on cfg 3
      rc = vg_host::gemm_ph_launch(p, a_tr, b_tr, tile_cfg, splits, stream);
      if (rc != 0) rc = vg_host::gemm_dma_launch(p, a_tr, b_tr, 3, splits, stream);
    } else {
      rc = vg_host::gemm_dma_launch(p, a_tr, b_tr, tile_cfg, splits, stream);
    }
    vg_host::prof_end(tok, stream);
    if (sig_debug == 3) {
      float ms = 0.f;
      hipEventRecord(lab1, stream);
      hipEventSynchronize(lab1);
      hipEventElapsedTime(&ms, lab0, lab1);
      fprintf(stderr, "[vg_gemm_t] us=%.1f cfg=%d M=%d N=%d K=%d a_tr=%d b_tr=%d splits=%d act=%d dact=%d res=%d aux_out=%d f32=%d acc=%d "
                      "colpart=%d\n", ms * 1e3, tile_cfg, p.M, p.N, p.K, a_tr, b_tr, splits, p.act, p.dact, p.residual != nullptr,
              p.aux_out != nullptr, p.out_f32, p.accumulate, p.colpart != nullptr);
      hipEventDestroy(lab0);
      hipEventDestroy(lab1);
    }
    if (rc == 0) return vg_host::check_launch("vg_gemm(dma)");
  }
  if (launch_thin<T>(p, a_tr, b_tr, splits, stream)) return vg_host::check_launch("vg_gemm(thin)");
  static const bool debug = getenv("VG_DEBUG_GEMM") != nullptr;
  if (debug)
    fprintf(stderr, "[vg_gemm] register-staged kernel: %s M=%d N=%d K=%d a_tr=%d b_tr=%d splits=%d act=%d dact=%d aux_out=%d res=%d\n",
            sizeof(T) == 4 ? "f32" : "bf16", p.M, p.N, p.K, a_tr, b_tr, splits, p.act, p.dact, p.aux_out != nullptr,
            p.residual != nullptr);
  if (p.colsum_out)   // register-staged path: the bias gradient is a separate pass over A = dY [K][lda]
    vg_host::colsum_accumulate(p.A, p.K, p.M, p.lda, p.colsum_out, sizeof(T) == 4 ? VG_F32 : VG_BF16, stream);
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, splits);
  dim3 block(NTHREADS);
  const size_t lds = 4 * TILE_BYTES;
  void (*k)(GemmParams) = nullptr;
  if (!a_tr && !b_tr) k = gemm_kernel<T, false, false>;
  else if (!a_tr && b_tr) k = gemm_kernel<T, false, true>;
  else if (a_tr && b_tr) k = gemm_kernel<T, true, true>;
  else k = gemm_kernel<T, true, false>;
  static bool attr_done[2][4] = {};
  const int ti = sizeof(T) == 4 ? 0 : 1, ki = a_tr * 2 + b_tr;
  if (!attr_done[ti][ki]) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done[ti][ki] = true;
  }
  const int tok = vg_host::prof_begin(kind, 2.0 * p.M * p.N * p.K, stream, gemm_algorithmic_bytes(p, sizeof(T)));
  VG_LAUNCH(k, grid, block, lds, stream, p);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_gemm");
}

}  // namespace

namespace {
// tile configuration the bf16 LDS-DMA path will use for this problem (-1: register-staged kernel)
int pick_cfg(const vg_gemm_desc* d, double* cost_out = nullptr) {
  if (cost_out) *cost_out = -1.0;          // (set below where the cost model chose between tile shapes)
  int cfg = d->tile_cfg;
  // K tails are zero-filled by the DMA (rows past K of a k-major operand, 16-byte chunks past the row end of a
  // k-contiguous one), so any K that keeps the 16-byte chunks whole qualifies
  const bool dma_ok = d->dtype == VG_BF16 && (d->K % 8 == 0 || (d->a_tr && d->b_tr)) && !(d->a_tr && !d->b_tr) &&
                      (long)(d->a_tr ? d->K : d->M) * d->lda * 2 < 0x7ffffff0L &&
                      (long)(d->b_tr ? d->K : d->N) * d->ldb * 2 < 0x7ffffff0L;
  if (!dma_ok) return -1;
  if (cfg == 15) {                 // the 192-row long-phase tile takes row-image A operands and whole 64-deep K tiles only
    const int sp = d->split_k > 0 ? d->split_k : 1;
    const int kp = (((d->K + sp - 1) / sp + 63) / 64) * 64;
    if (d->a_tr || d->K % 64 != 0 || kp % 64 != 0) cfg = 3;
  }
  if (cfg != 0) return cfg;
  // Forward / dgrad products: the tile shape with the least estimated time.  A launch runs in rounds of
  // 256 x blocks-per-CU tiles; a round costs its tile area times a measured per-shape factor, a partly filled last
  // round a little less.  With whole 64-deep K tiles the choice is between 128x128 (two blocks per CU) and the
  // long-phase 256x256 schedule; other K keep the round-1 model (128x128 / 2-stage 256x256 / 192x256).
  // Weight gradients: 256x256 once the caller's split fills a good part of the chip, else 128x128 + split-K.
  cfg = 1;
  // phase-pipelined 256x256 kernels (vg_gemm_ph.hip) need whole 64-deep K tiles in every split
  const int splits = d->split_k > 0 ? d->split_k : 1;
  int kps = (d->K + splits - 1) / splits;
  kps = ((kps + 63) / 64) * 64;
  const bool ph_ok = d->K % 64 == 0 && kps % 64 == 0;
  if (!d->a_tr) {
    auto cost = [&](int rows, int cols, int per_cu, double shape, double part) {
      const long tiles = (long)((d->M + rows - 1) / rows) * ((d->N + cols - 1) / cols), slots = 256L * per_cu;
      const long full = tiles / slots, rem = tiles % slots;
      const double rounds = (double)full + (rem ? part + (1.0 - part) * (double)rem / (double)slots : 0.0);
      return rounds * rows * cols * per_cu * shape;
    };
    const double longk_pen = 0.10 * fmin(1.0, fmax(0.0, (d->K - 1024) / 3072.0));   // 128x128 falls behind at long K
    if (ph_ok) {
      // the long-phase 256x256 schedule with the lean epilogues against 128x128 (tools/tile_cold_sweep.py at M = 5120 /
      // 8000 / 10240 / 16000 / 32000, CFGS=1,9,12,13): it wins wherever its tiles cover about half the CUs or more
      // (M = 10240: 160 tiles, 25 % faster than 128x128 or 192x256; M = 8000, N = 1024: 128 tiles, a tie; M = 5120,
      // N = 1024: 80 tiles, 128x128 is 15 % faster).  A partly filled round of one-block-per-CU tiles costs a whole one.
      const double c1 = cost(128, 128, 2, 1.08 + longk_pen, 0.85), c13 = cost(256, 256, 1, 0.62, 0.93);
      cfg = c13 <= c1 ? 13 : 1;
      // round 3: the same schedule on 192 x 256 tiles (tile_cfg 15) where its rounds are cheaper -- 216 tiles instead
      // of 160 at M = 10240 (the yaml's 2 x 8 x 640 frames), 168 instead of 128 at M = 8000: three quarters of a round's
      // time for the N = 1024 products.  It has to win clearly (the narrower tile streams a third more B per FLOP).
      static const int no15 = [] { const char* e = getenv("VG_NO_CFG15"); return e ? atoi(e) : 0; }();
      const double c15 = cost(192, 256, 1, 0.70, 0.93);     // per-area factor from tools/tile_cold_sweep.py CFGS=1,13,15 at M = 8000 / 10240
      if (!no15 && c15 < 0.93 * fmin(c1, c13)) cfg = 15;
      if (cost_out) *cost_out = cfg == 15 ? c15 : (cfg == 13 ? c13 : c1);
    } else {
      const double c1 = cost(128, 128, 2, 1.08 + longk_pen, 0.85), c3 = cost(256, 256, 1, 1.0, 0.85),
                   c9 = cost(192, 256, 1, 0.97, 0.85);
      cfg = c3 <= c1 ? 3 : 1;
      if (c9 < 0.95 * fmin(c1, c3)) cfg = 9;      // the odd shape must win clearly (model error ~5 %)
    }
  } else if (d->b_tr && ph_ok && !d->colsum_out) {
    // weight gradients: 256x256 ring tiles once they can fill a good part of the chip with the caller's split
    const long tiles = (long)((d->M + 255) / 256) * ((d->N + 255) / 256) * splits;
    if (tiles >= 96) cfg = 13;
  }
  static const int longk = [] { const char* e = getenv("VG_CFG_LONGK"); return e ? atoi(e) : 5; }();
  if (cfg == 1 && !d->a_tr && !d->b_tr && d->K >= 4096 && d->N <= 1024 && d->M >= 2048 && longk > 0) cfg = longk;
  if (d->a_tr && d->b_tr && d->colsum_out) cfg = 1;
  static const int no_ph = [] { const char* e = getenv("VG_NO_PH"); return e ? atoi(e) : 0; }();
  if (no_ph && cfg >= 10) cfg = d->a_tr ? 1 : 3;     // A/B switch: the round-1 kernels
  return cfg;
}
int cfg_tile_rows(int cfg) { return (cfg == 9 || cfg == 15) ? 192 : (cfg == 2 || cfg == 3 || cfg == 5 || cfg == 6 || cfg >= 10) ? 256 : 128; }
}  // namespace

extern "C" int vg_gemm_tile_rows(const vg_gemm_desc* d) {
  if (d == nullptr) return 0;
  const int cfg = pick_cfg(d);
  return cfg > 0 ? cfg_tile_rows(cfg) : 0;
}

namespace {
// checks a descriptor and fills the kernel parameter block; returns 0 or an error code (message set)
int fill_params(const vg_gemm_desc* d, GemmParams& p, int& splits, const char* who) {
  VG_REQUIRE(d != nullptr, "%s: null descriptor", who);
  VG_REQUIRE(d->dtype == VG_F32 || d->dtype == VG_BF16, "%s: bad dtype %d", who, d->dtype);
  VG_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "%s: empty problem %d %d %d", who, d->M, d->N, d->K);
  const int vec = d->dtype == VG_BF16 ? 8 : 4;
  const int bk = d->dtype == VG_BF16 ? 64 : 32;
  VG_REQUIRE(d->lda % vec == 0 && d->ldb % vec == 0, "%s: lda/ldb must be multiples of %d", who, vec);
  if (!d->a_tr || !d->b_tr) VG_REQUIRE(d->K % vec == 0, "%s: K must be a multiple of %d", who, vec);
  if (d->a_tr) VG_REQUIRE(d->M % vec == 0, "%s: M must be a multiple of %d for a_tr", who, vec);
  if (d->b_tr) VG_REQUIRE(d->N % vec == 0, "%s: N must be a multiple of %d for b_tr", who, vec);
  VG_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0, "%s: A/B must be 16-byte aligned", who);
  splits = d->split_k > 0 ? d->split_k : 1;
  VG_REQUIRE(splits == 1 || d->out_f32, "%s: split-K needs an fp32 (pre-zeroed) C", who);
  p.A = d->A; p.B = d->B; p.C = d->C;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.bias = d->bias; p.residual = d->residual; p.aux_in = d->aux_in; p.aux_out = d->aux_out;
  p.pre_add = d->pre_add;
  p.lengths = d->lengths; p.T = d->T > 0 ? d->T : 1;
  p.act = d->act & ~VG_ACT_DERIV_U8; p.dact = d->dact & ~VG_ACT_DERIV_U8; p.out_f32 = d->out_f32; p.accumulate = d->accumulate;
  p.m_base = 0;
  p.aux_u8 = ((d->act | d->dact) & VG_ACT_DERIV_U8) != 0;
  // (needs whole 16-byte chunks of codes per row image: N % 16 == 0 and 16-byte aligned rows -- what the lean 8-bit
  // epilogue needs anyway -- and the 32-bit piece offsets of the ring: the code array below 2 GiB)
  static const int aux_ring = [] { const char* e = getenv("VG_AUX_RING"); return e ? atoi(e) : 1; }();
  p.aux_ring = aux_ring && p.aux_u8 && (long)d->M * d->ldc < 0x7ffffff0L;
  if (p.aux_u8) {     // one byte per element of the stored GELU derivative (bf16 launches; include/vaegslm_hip.h)
    VG_REQUIRE(d->dtype == VG_BF16 && splits == 1 && !d->a_tr, "%s: VG_ACT_DERIV_U8 needs a bf16 forward / dgrad product without split-K", who);
    const bool save8 = (d->act & VG_ACT_DERIV_U8) != 0, load8 = (d->dact & VG_ACT_DERIV_U8) != 0;
    VG_REQUIRE(!save8 || (p.act == (VG_ACT_GELU | VG_ACT_SAVE_DERIV) && d->aux_out != nullptr),
               "%s: VG_ACT_DERIV_U8 in act goes with VG_ACT_GELU | VG_ACT_SAVE_DERIV and an aux_out", who);
    VG_REQUIRE(!load8 || (p.dact == VG_ACT_STORED && d->aux_in != nullptr), "%s: VG_ACT_DERIV_U8 in dact goes with VG_ACT_STORED and an aux_in", who);
    VG_REQUIRE(save8 != load8, "%s: one launch either stores or reads the 8-bit derivative", who);
    VG_REQUIRE(d->ldc % 8 == 0 && d->N % 8 == 0 && (((uintptr_t)d->aux_in | (uintptr_t)d->aux_out) & 7) == 0,
               "%s: VG_ACT_DERIV_U8 moves 8 codes at a time (ldc, N multiples of 8, 8-byte aligned aux)", who);
  }
  p.alpha = d->alpha;
  p.colsum_out = d->colsum_out;
  static const int cs_rr = [] { const char* e = getenv("VG_COLSUM_RR"); return e ? atoi(e) : 1; }();
  p.colsum_rr = cs_rr;
  // tile order of a plain launch.  Products up to 8 column-tiles wide: bands of 4 row-tiles walked m-fastest (tools/gemm_rotate.py
  // + bench.py, round 3: +0.9 % end to end over n-fastest, 8: +0.6 %).  N >= 3072 (QKV, FFN-in and the dgrads of their shape):
  // n-fastest, an XCD's 32 blocks = two or three whole rows of tiles -- round 5, alternating bench.py runs of one call
  // (profiles/r05/labs/tile_order_wide_products.txt): NT launches 66.1 -> 64.3 us on average, NN 69.8 -> 69.2, step 28.24 ->
  // 28.08 ms; the same pair on cold operands in isolation measures level (131 - 133 us either way at N = 4096).
  static const int group_m = [] { const char* e = getenv("VG_GEMM_GROUP_M"); return e ? atoi(e) : 4; }();
  static const int group_m_wide = [] { const char* e = getenv("VG_GEMM_GROUP_M_WIDE"); return e ? atoi(e) : 0; }();
  static const int wide_n = [] { const char* e = getenv("VG_GEMM_WIDE_N"); return e ? atoi(e) : 3072; }();
  p.group_m = (group_m_wide >= 0 && !d->a_tr && d->N >= wide_n) ? group_m_wide : group_m;
  VG_REQUIRE(d->colsum_out == nullptr || (d->a_tr && d->b_tr), "%s: colsum_out needs a_tr = b_tr = 1", who);
  int kps = (d->K + splits - 1) / splits;
  kps = ((kps + bk - 1) / bk) * bk;
  splits = (d->K + kps - 1) / kps;
  p.k_per_split = kps;
  p.colpart = d->colpart;
  p.split_ws = nullptr;
  p.split_cnt = nullptr;
  return 0;
}
}  // namespace

namespace {
// Round 6: partial rounds -- split M, not K.  A forward / dgrad product on 256 x 256 tiles whose last
// round of 256 CUs is badly filled (M = 13,312 packed rows x N = 4096: 832 tiles = 3.25 rounds, paid as 4) runs as TWO
// launches: the row-tiles that make whole rounds on the 256 x 256 schedule, and the remaining row band as a product of its
// own on whatever tile shape the cost model likes best for it (128 x 128 at two blocks per CU, 192 x 256, ...).  No slab,
// no atomics; every output row is computed by one launch exactly as the one-launch form would compute it on that tile
// shape.  Returns the rows of the first launch (a multiple of 256), or 0: no split.
// MEASURED AND NOT ADOPTED (profiles/r06/labs/split_m_pieces_cold_operands.txt, one call, cold operands): the premise does
// not hold on this kernel -- a quarter-full fourth round costs 18 us, not a round's 31 (the CUs that get no fourth tile
// leave the memory system to the others), and the row band as a launch of its own costs the same 17 us:
//   M = 13,312 x N = 4096 x K = 1024: one launch 110.5 us; 12,288 rows 92.5 + 1,024 rows 17.2 (128 x 128) = 109.7 us
//   M = 13,312 x N = 3072:            one launch  82.3 us; 10,752 rows 61.5 + 2,560 rows 20.2             =  81.7 us
//   M = 10,240 x N = 4096:            one launch  82.6 us;  8,192 rows 62.9 + 2,048 rows 21.5             =  84.4 us
// and bench.py --ragged / --seq-len 640 measure the same with and without it.  VG_GEMM_SPLIT_M: 0 (default) off, 1 = by the
// cost model below, 2 = whenever the last round is under 0.7 full (what tests/test_parity_round6_gpu.py forces).  Read
// per call (not cached) so that one process can measure both.
int split_rows(const vg_gemm_desc* d, int cfg, double cost_whole) {
  const char* env = getenv("VG_GEMM_SPLIT_M");
  const int on = env ? atoi(env) : 0;
  if (!on || d->tile_cfg != 0 || cfg != 13 || d->a_tr || (d->split_k > 1) || cost_whole <= 0.0) return 0;
  const long ntn = (d->N + 255) / 256, ntm = (d->M + 255) / 256, tiles = ntm * ntn;
  const long full = tiles / 256, rem = tiles % 256;
  if (full < 1 || rem == 0 || rem * 10 >= 256 * 7) return 0;             // the last round is at least 0.7 full: leave it
  const long ntm1 = full * 256 / ntn;                                       // row-tiles of the whole rounds
  if (ntm1 < 1 || ntm1 >= ntm) return 0;
  vg_gemm_desc d1 = *d, d2 = *d;
  d1.M = (int)(ntm1 * 256);
  d2.M = d->M - d1.M;
  double c1 = -1.0, c2 = -1.0;
  (void)pick_cfg(&d1, &c1);
  const int cfg2 = pick_cfg(&d2, &c2);
  if (c1 <= 0.0 || c2 <= 0.0 || cfg2 <= 0) return 0;
  // the second launch pays its own launch + prologue + epilogue (~6-9 us of a ~100 us product: 0.06 of a round's cost)
  const double extra = 0.06 * 256.0 * 256.0 * 0.62;
  return (on >= 2 || c1 + c2 + extra < 0.97 * cost_whole) ? d1.M : 0;
}
}  // namespace

/* rows of the fp32 [rows][N] array a launch with `colpart` fills: one per row-tile of each of its (one or two) launches */
extern "C" int vg_gemm_colpart_rows(const vg_gemm_desc* d) {
  if (d == nullptr) return 0;
  double cw = -1.0;
  const int cfg = pick_cfg(d, &cw);
  if (cfg <= 0) return 0;
  const int m1 = split_rows(d, cfg, cw);
  if (m1 == 0) return (d->M + cfg_tile_rows(cfg) - 1) / cfg_tile_rows(cfg);
  vg_gemm_desc d2 = *d;
  d2.M = d->M - m1;
  const int r2 = cfg_tile_rows(pick_cfg(&d2));
  return m1 / 256 + (d2.M + r2 - 1) / r2;
}

extern "C" int vg_gemm(const vg_gemm_desc* d, hipStream_t stream) {
  GemmParams p;
  int splits = 1;
  if (int e = fill_params(d, p, splits, "vg_gemm")) return e;
  // tile_cfg: 0 = auto, -1 = force the register-staged kernel, 1.. = LDS-DMA tile shapes
  double cost_whole = -1.0;
  const int cfg = pick_cfg(d, &cost_whole);
  if (const int m1 = (d->dtype == VG_BF16 && splits == 1 && !d->split_ws) ? split_rows(d, cfg, cost_whole) : 0) {
    // two launches over disjoint row ranges; the second one's row mask counts from the whole product's first row
    vg_gemm_desc d2 = *d;
    d2.M = d->M - m1;
    const int cfg2 = pick_cfg(&d2);
    GemmParams p1 = p, p2 = p;
    p1.M = m1;
    p2.M = d->M - m1;
    p2.m_base = m1;
    const size_t esz = 2;
    p2.A = (const char*)p.A + (size_t)m1 * p.lda * esz;
    p2.C = (char*)p.C + (size_t)m1 * p.ldc * (p.out_f32 ? 4 : esz);
    if (p.residual) p2.residual = (const char*)p.residual + (size_t)m1 * p.ldc * esz;
    if (p.pre_add) p2.pre_add = (const char*)p.pre_add + (size_t)m1 * p.ldc * esz;
    if (p.aux_in) p2.aux_in = (const char*)p.aux_in + (size_t)m1 * p.ldc * (p.aux_u8 ? 1 : esz);
    if (p.aux_out) p2.aux_out = (char*)p.aux_out + (size_t)m1 * p.ldc * (p.aux_u8 ? 1 : esz);
    if (p.colpart) p2.colpart = p.colpart + (size_t)(m1 / 256) * p.N;
    if (int e = launch<bf16_t>(p1, d->a_tr, d->b_tr, 1, cfg, stream)) return e;
    return launch<bf16_t>(p2, d->a_tr, d->b_tr, 1, cfg2, stream);
  }
  if (splits > 1 && cfg > 0 && d->split_ws != nullptr && d->split_cnt != nullptr) {
    // in-launch reduction of the K slices through fp32 slabs: needs splits * tiles * tile floats of workspace
    const int rows = cfg_tile_rows(cfg), cols = (cfg == 3 || cfg == 4 || (cfg >= 10 && cfg != 14)) ? 256 : 128;
    const long tiles = (long)((d->M + rows - 1) / rows) * ((d->N + cols - 1) / cols);
    if (tiles <= 4096 && tiles * rows * cols * (long)splits <= d->split_ws_floats && d->ldc % 4 == 0 &&
        ((uintptr_t)d->C % 16) == 0) {
      p.split_ws = d->split_ws;
      p.split_cnt = d->split_cnt;
      if (hipMemsetAsync(d->split_cnt, 0, (size_t)tiles * sizeof(int), stream) != hipSuccess) {
        vg_host::set_error("vg_gemm: could not zero the split-K counters");
        return 2;
      }
    }
  }
  VG_REQUIRE(!p.aux_u8 || cfg > 0, "vg_gemm: VG_ACT_DERIV_U8 needs the bf16 LDS-DMA path (K %% 8 == 0, operands below 2 GiB)");
  VG_REQUIRE(d->colpart == nullptr || (cfg > 0 && splits == 1),
             "vg_gemm: colpart needs the bf16 LDS-DMA path without split-K (ask vg_gemm_tile_rows first)");
  if (d->dtype == VG_BF16) return launch<bf16_t>(p, d->a_tr, d->b_tr, splits, cfg, stream);
  return launch<float>(p, d->a_tr, d->b_tr, splits, -1, stream);
}

extern "C" int vg_gemm_grouped(const vg_gemm_desc* descs, int n, hipStream_t stream) {
  VG_REQUIRE(descs != nullptr && n >= 1 && n <= VG_GROUP_MAX, "vg_gemm_grouped: n = %d (1..%d)", n, VG_GROUP_MAX);
  GemmParams ps[VG_GROUP_MAX];
  int splits[VG_GROUP_MAX];
  double work = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const vg_gemm_desc* d = descs + i;
    if (int e = fill_params(d, ps[i], splits[i], "vg_gemm_grouped")) return e;
    VG_REQUIRE(d->dtype == VG_BF16 && d->a_tr && d->b_tr && d->out_f32, "vg_gemm_grouped: problem %d is not a bf16 TN product with fp32 C", i);
    VG_REQUIRE(d->K % 64 == 0 && ps[i].k_per_split % 64 == 0, "vg_gemm_grouped: problem %d: K = %d / split %d is not in whole 64-deep tiles",
               i, d->K, splits[i]);
    VG_REQUIRE(!d->bias && !d->residual && !d->aux_in && !d->aux_out && !d->pre_add && !d->lengths && d->act == VG_ACT_NONE &&
                   d->dact == VG_ACT_NONE && !d->colsum_out && !d->colpart,
               "vg_gemm_grouped: problem %d carries an epilogue", i);
    VG_REQUIRE(splits[i] == 1, "vg_gemm_grouped: problem %d: split_k = %d (must be 1: the launch divides the reduction itself)", i, splits[i]);
    VG_REQUIRE(d->alpha == 1.0f && d->N % 8 == 0 && d->ldc % 4 == 0 && ((uintptr_t)d->C % 16) == 0,
               "vg_gemm_grouped: problem %d: C += A^T B into 16-byte aligned fp32 rows only (alpha = 1, N %% 8 == 0)", i);
    VG_REQUIRE((long)d->K * d->lda * 2 < 0x7ffffff0L && (long)d->K * d->ldb * 2 < 0x7ffffff0L, "vg_gemm_grouped: problem %d is too large", i);
    work += 2.0 * d->M * d->N * d->K;
    bytes += gemm_algorithmic_bytes(ps[i], 2);
  }
  const int tok = vg_host::prof_begin(VG_PROF_GEMM_BF16_TN, work, stream, bytes);
  static const int lab_debug = [] { const char* e = getenv("VG_DEBUG_GEMM"); return e ? atoi(e) : 0; }();
  hipEvent_t lab0 = nullptr, lab1 = nullptr;
  if (lab_debug == 3) {
    hipEventCreate(&lab0);
    hipEventCreate(&lab1);
    hipEventRecord(lab0, stream);
  }
  const int rc = vg_host::gemm_group_launch(ps, splits, n, stream);
  vg_host::prof_end(tok, stream);
  if (lab_debug == 3) {
    float ms = 0.f;
    hipEventRecord(lab1, stream);
    hipEventSynchronize(lab1);
    hipEventElapsedTime(&ms, lab0, lab1);
    fprintf(stderr, "[vg_gemm_t] us=%.1f cfg=group M=%d N=%d K=%d a_tr=1 b_tr=1 splits=%d gflop=%.1f\n", ms * 1e3, ps[0].M, ps[0].N,
            ps[0].K, n, work * 1e-9);
    hipEventDestroy(lab0);
    hipEventDestroy(lab1);
  }
  if (rc != 0) {
    vg_host::set_error("vg_gemm_grouped: launch failed");
    return 3;
  }
  return vg_host::check_launch("vg_gemm_grouped");
}
