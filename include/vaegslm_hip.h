/* libvaegslm_hip -- C ABI of the MI355X (gfx950) VAE-GSLM training hot path.
 *
 * The reference (b04901014/vae-gslm) has no native layer: its hot path is a
 * chain of ATen calls inside Python nn.Modules.  Each entry point below
 * replaces one such call site (cited as reference file:line); the Python
 * binding a maintainer adds is shown in INTEGRATION.md (ctypes).
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; the message is
 *    retrievable with vg_last_error() (thread-local); nothing throws;
 *  - all pointers are DEVICE pointers owned by the caller unless noted;
 *    no function allocates, frees or synchronises -> safe under stream
 *    capture (hipGraph);
 *  - `stream` is the hipStream_t the work is enqueued on;
 *  - sequences are right-padded: frame (b, t) is valid iff t < lengths[b]
 *    (utils/tensormask.py:45-54).  `lengths` may be NULL (= all valid).
 *    Rows are frames in (b, t) order: row m = b * T + t;
 *  - dtype selects the STORAGE type of activations: VG_F32 runs the
 *    exact-f32 MFMA path (parity), VG_BF16 the bf16 MFMA path (speed);
 *    accumulation, softmax statistics, norms and losses are always fp32.
 */
#ifndef VAEGSLM_HIP_H
#define VAEGSLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vg_stream_t; /* == hipStream_t */

enum { VG_F32 = 0, VG_BF16 = 1 };
enum { VG_ACT_NONE = 0, VG_ACT_RELU = 1, VG_ACT_GELU = 2, VG_ACT_SILU = 3,
       /* dact only: multiply by aux_in, a derivative stored by a forward launch with VG_ACT_SAVE_DERIV */
       VG_ACT_STORED = 4,
       /* flag OR-ed into act: aux_out receives act'(pre-activation) instead of the pre-activation, so the
          backward epilogue is one multiply instead of re-evaluating erf/exp per element */
       VG_ACT_SAVE_DERIV = 16,
       /* flag OR-ed into act (with VG_ACT_GELU | VG_ACT_SAVE_DERIV) and into dact (with VG_ACT_STORED), bf16 launches
          only (round 6): the stored derivative is ONE BYTE per element, aux_out / aux_in = uint8 [M][ldc] (ldc counted in
          elements = bytes, ldc % 8 == 0, 8-byte aligned base), code = round((GELU' - VG_DERIV_U8_LO) / VG_DERIV_U8_STEP).
          GELU' lies in [-0.1290, 1.1290], so 256 codes of step 0.005 from -0.13 cover it: absolute error <= 0.0025, what
          bf16 keeps near 1 (0.002 - 0.004) -- and half the bytes of the second M x N stream of both FFN-in launches. */
       VG_ACT_DERIV_U8 = 32 };
#define VG_DERIV_U8_STEP 0.005f
#define VG_DERIV_U8_LO (-0.13f)

int vg_version(void);
/* copies the calling thread's last error message (NUL terminated) */
int vg_last_error(char* buf, int buflen);

/* ---------------------------------------------------------------- GEMM
 * C[M,N] = epi( alpha * sum_k A(m,k) B(k,n) )
 * Replaces nn.Linear forward/backward at modules/attention/attention.py:52,79
 * (in_proj/out_proj), modules/transformer/layers.py:82 (FFN linear1/2 with
 * GELU, modules/activations.py:11), :151-154 (stack input projection) and
 * modules/linear/layers.py:192-193 (Linear heads, +ReLU), :87-109 (Gaussian
 * head mean/logstd projections).
 *   a_tr = 0: A stored [M][lda] (k contiguous);  a_tr = 1: A stored [K][lda] (m contiguous)
 *   b_tr = 0: B stored [N][ldb] (k contiguous);  b_tr = 1: B stored [K][ldb] (n contiguous)
 * Epilogue order: +bias[n] -> +pre_add -> (aux_out = value, or act'(value) with VG_ACT_SAVE_DERIV) -> act ->
 * *dact'(aux_in) (or *aux_in for VG_ACT_STORED) ->
 * +residual -> row mask (zero rows t >= lengths[b]) -> store.
 * split_k > 1: fp32 C must be pre-zeroed; partial sums are added atomically
 * and the epilogue is skipped (used for weight gradients only).
 */
typedef struct vg_gemm_desc {
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int a_tr, b_tr;
  int dtype;            /* storage type of A, B, residual, aux_*, and of C unless out_f32 */
  const float* bias;    /* [N] fp32 or NULL */
  const void* residual; /* [M][ldc] or NULL */
  const void* aux_in;   /* [M][ldc] input of the activation derivative (dact) */
  void* aux_out;        /* [M][ldc] receives the pre-activation value or NULL */
  const int32_t* lengths;
  int T;
  int act, dact;
  int out_f32;
  int accumulate;       /* C += (fp32 C, split_k == 1) */
  int split_k;
  float alpha;
  const void* pre_add;  /* [M][ldc] added before the activation (conditioning term of the conv blocks) or NULL */
  int tile_cfg;         /* 0 = auto; -1 = register-staged 128x128; 1..5 = LDS-DMA 128x128 / 256x128 / 256x256 / 128x256 / 256x128 with a 3-stage ring;
                           11 / 12 / 13 = phase-pipelined 256x256 (ring / complementary / complementary with long phases and
                           compile-time epilogues: what auto picks for whole 64-deep K tiles), 14 = two 256x128 blocks per CU,
                           15 = the long-phase schedule on 192x256 tiles (auto: where 256-row tiles leave a round of CUs
                           badly filled, e.g. M = 10240 or 8000 with N = 1024) (vg_gemm_ph.hip) */
  float* colsum_out;    /* a_tr = b_tr = 1 only (weight gradient dY^T X): colsum_out[m] += sum_k A(m,k), i.e. the bias
                           gradient of the same Linear (modules/linear/layers.py:192) from the tiles already in LDS; NULL = off */
  float* colpart;       /* [ceil(M / vg_gemm_tile_rows(desc))][N] fp32 or NULL: per-row-tile column sums of the stored result
                           (the dgrad that writes a Linear's input gradient also reduces it for the bias gradient of
                           the layer below); bf16 LDS-DMA path, split_k == 1 only */
  float* split_ws;      /* optional workspace for split_k > 1 on the LDS-DMA path: the K slices meet through fp32 slabs and */
  int32_t* split_cnt;   /* one arrival counter per output tile (wait-free in-launch reduction, plain 16-byte traffic) */
  int64_t split_ws_floats; /* instead of fp32 atomics; used when split_k * tiles * tile size <= split_ws_floats (4096 counters
                           suffice for every tile grid of <= 4096 tiles), otherwise the launch falls back to atomics */
} vg_gemm_desc;
int vg_gemm(const vg_gemm_desc* desc, vg_stream_t stream);
/* rows per output tile the launch for `desc` will use (128 or 256), 0 if it takes the register-staged kernel */
int vg_gemm_tile_rows(const vg_gemm_desc* desc);
/* rows of the fp32 [rows][N] array `colpart` a call with this descriptor fills (0: the call cannot produce it).  Round 6:
 * a forward / dgrad product whose last round of 256 x 256 tiles would be badly filled runs as two launches over disjoint
 * row ranges (whole rounds + the remaining row band on the tile shape that suits it), each writing its own row-tiles'
 * column sums: the count is no longer ceil(M / vg_gemm_tile_rows). */
int vg_gemm_colpart_rows(const vg_gemm_desc* desc);
/* Several weight-gradient products in ONE launch.  Contract (anything else is refused with a message in
 * vg_last_error and nothing is launched -- run those through vg_gemm): every desc bf16, a_tr = b_tr = 1, fp32 C
 * (out_f32 = 1) in 16-byte aligned rows (N % 8 == 0, ldc % 4 == 0), K a multiple of 64, split_k = 1,
 * alpha = 1 (C += A^T B: C must hold its initial value; the launch divides the reduction among its blocks itself
 * and combines whole-K segments with plain adds, head / tail pieces with fp32 atomics), no epilogue fields.
 * accumulate = 0 (round 4) is the caller's statement that C holds ZEROS -- the first contribution since the optimizer
 * cleared the gradient: whole-K segments then store without reading C; pieces still add atomically.  Replaces
 * the four dW = dY^T X launches of one Transformer layer's backward (the nn.Linear weight gradients autograd computes
 * for modules/transformer/layers.py:52,79,82,151 of the reference) and the two or three of a conv bottleneck block
 * (modules/conv/layers.py): one persistent grid of 256 blocks with equal (tile, K tile) unit counts.  Round 4: up to
 * VG_GROUP_MAX = 48 products per launch (the host side queues the weight gradients of several backward nodes -- four
 * Transformer layers are 768 tiles = three whole rounds of 256 CUs -- and flushes them together). */
enum { VG_GROUP_MAX = 48 };
int vg_gemm_grouped(const vg_gemm_desc* descs, int n, vg_stream_t stream);

/* ---------------------------------------------------------------- RMSNorm
 * modules/norm.py:28-32 fused with the re-mask of
 * modules/transformer/layers.py:52-54:  y = mask ? scale * x * rsqrt(mean(x^2)+eps) : 0
 * x, y: [M][C] (dtype); scale: fp32 [C]; rstd: fp32 [M] (saved for backward).
 */
int vg_rmsnorm_fwd(const void* x, const float* scale, void* y, float* rstd, int M, int C, float eps,
                   const int32_t* lengths, int T, int dtype, vg_stream_t stream);
/* dx = (dx_add ? dx_add : 0) + d(rmsnorm)/dx . dy ; dscale_partial: fp32 [nblocks][C]
 * (nblocks = vg_rmsnorm_bwd_blocks(M)); reduce with vg_colsum_f32. */
int vg_rmsnorm_bwd_blocks(int M);
int vg_rmsnorm_bwd(const void* dy, const void* x, const float* scale, const float* rstd, const void* dx_add,
                   void* dx, float* dscale_partial, int M, int C, const int32_t* lengths, int T, int dtype,
                   vg_stream_t stream);
/* the same launch, which also leaves the column sums of the dx it stores: dx_colsum_partial fp32 [nblocks][C] (C at most
 * 128 16-byte vectors; NULL = vg_rmsnorm_bwd).  dx of a layer's first norm is the incoming gradient of the layer below,
 * whose FFN-out bias gradient is the column sum of exactly that tensor (modules/transformer/layers.py:82-86). */
int vg_rmsnorm_bwd_colsum(const void* dy, const void* x, const float* scale, const float* rstd, const void* dx_add,
                          void* dx, float* dscale_partial, float* dx_colsum_partial, int M, int C,
                          const int32_t* lengths, int T, int dtype, vg_stream_t stream);

/* ---------------------------------------------------------------- attention
 * Causal multi-head self-attention with in-kernel ALiBi, replacing the mask
 * materialisation + F.scaled_dot_product_attention at
 * modules/attention/attention.py:60-77 and modules/position/alibi.py:9-33.
 * qkv: [B*T][3*H*64] = in_proj output (q | k | v, head h = columns 64h..64h+63,
 * attention.py:12-18,52); out: [B*T][H*64] (heads merged, :78); rows
 * t >= lengths[b] of `out` are written as zeros.  head_dim must be 64.
 * slopes: fp32 [H] (positive; bias = -slope * (i - j)).  lse: fp32 [H][B*T] (head-major over the rows of qkv;
 * only passed on to vg_attn_bwd).
 */
int vg_attn_fwd(const void* qkv, void* out, float* lse, const float* slopes, int B, int T, int H,
                const int32_t* lengths, int dtype, vg_stream_t stream);
/* dqkv: [B*T][3*H*64]; delta: fp32 workspace [H][B*T]. */
int vg_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const float* slopes,
                void* dqkv, float* delta, int B, int T, int H, const int32_t* lengths, int dtype,
                vg_stream_t stream);
/* The same two calls on PACKED rows (the valid frames of a right-padded batch stored back to back, as
 * utils/tensormask.py:63-67's apply_mask makes the padded ones irrelevant): sequence b owns rows
 * cu_rows[b] .. cu_rows[b + 1] - 1 of qkv / out / dout / dqkv (`rows` rows in all; lengths[b] of them are attended,
 * normally all), Tmax >= every length sizes the grid.  A sequence with lengths[b] = 0 has its rows zero-filled: the
 * caller appends such pseudo sequences (each <= Tmax rows) to cover rows it padded the packed tensors with.
 * lse / delta: fp32 [H][rows]. */
int vg_attn_fwd_varlen(const void* qkv, void* out, float* lse, const float* slopes, int B, int Tmax, int H,
                       const int32_t* lengths, const int32_t* cu_rows, int rows, int dtype, vg_stream_t stream);
int vg_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse, const float* slopes,
                       void* dqkv, float* delta, int B, int Tmax, int H, const int32_t* lengths,
                       const int32_t* cu_rows, int rows, int dtype, vg_stream_t stream);
/* The same calls with the ALiBi window of the backward pass (round 5).  `stats`: fp32 workspace of
 * vg_attn_stats_floats(B, T, H) floats per call pair, written by the bf16 forward (per (batch, head): max |k|^2, and per
 * 32-query group max |q|^2 and max -logsumexp; plain stores, no initialisation needed) and read by the backward, which then
 * does not stream key / query tiles whose every probability is provably below 2^-20 of its row (the same threshold the
 * kernels already drop products at; fp32 launches ignore it and keep every tile, as does VG_ATTN_WINDOW=0).  The
 * reference (modules/attention/attention.py:60-77) computes the dense masked softmax; under ALiBi
 * (modules/position/alibi.py:9-33) those entries are exactly the ones that underflow bf16.  cu_rows = NULL (and
 * rows = 0): padded layout as vg_attn_fwd; otherwise packed rows as vg_attn_fwd_varlen.  stats = NULL: no window. */
int vg_attn_stats_floats(int B, int T, int H);
int vg_attn_fwd_stats(const void* qkv, void* out, float* lse, const float* slopes, int B, int T, int H,
                      const int32_t* lengths, const int32_t* cu_rows, int rows, float* stats, int dtype,
                      vg_stream_t stream);
int vg_attn_bwd_stats(const void* qkv, const void* out, const void* dout, const float* lse, const float* slopes,
                      void* dqkv, float* delta, int B, int T, int H, const int32_t* lengths, const int32_t* cu_rows,
                      int rows, const float* stats, int dtype, vg_stream_t stream);
/* dst[i] = map[i] >= 0 ? src[map[i]] : 0 for rows of row_bytes (a multiple of 16) bytes: packing the valid frames of a
 * padded batch, un-packing them, and each other's backward. */
int vg_gather_rows(const void* src, const int32_t* map, void* dst, int n_dst, int row_bytes, vg_stream_t stream);
/* Single-query decode step against a pre-allocated KV cache (attention.py:56-73
 * with past_kv).  q: [B][H*64]; kcache/vcache: [B][Tmax][H*64]; pos[b] = number
 * of valid cache rows INCLUDING the current one; out: [B][H*64]. */
int vg_attn_decode(const void* q, const void* kcache, const void* vcache, void* out, const float* slopes,
                   const int32_t* pos, int B, int Tmax, int H, int dtype, vg_stream_t stream);

/* ---------------------------------------------------------------- token cross-entropy
 * training_lib/losses.py:30-41 (masked_ce_loss, reduction="sum", ignore -100).
 * logits: [M][V] (dtype); targets: int64 [M]; loss_rows: fp32 [M] (0 on padded rows);
 * lse: fp32 [M]; argmax: int32 [M].  bwd: dlogits = gscale * (softmax - onehot), 0 on padded rows.
 */
int vg_ce_fwd(const void* logits, const int64_t* targets, float* loss_rows, float* lse, int32_t* argmax, int M,
              int V, int64_t ld, const int32_t* lengths, int T, int dtype, vg_stream_t stream);
int vg_ce_bwd(const void* logits, const int64_t* targets, const float* lse, const float* gscale /*device scalar*/,
              void* dlogits, int M, int V, int64_t ld, const int32_t* lengths, int T, int dtype,
              vg_stream_t stream);

/* ---------------------------------------------------------------- VAE terms
 * Posterior reparameterisation (modules/linear/layers.py:110-128,
 * models/speech/lvtr.py:157-160):
 *   z = mask ? mu + eps * exp(logstd) * temperature : 0 ; log_q = mask ? -logstd - 0.5 - 0.5 ln 2pi : 0
 * mu/logstd/eps/z/log_q: fp32 [M][D].
 */
int vg_reparam_fwd(const float* mu, const float* logstd, const float* eps, float* z, float* log_q, int M, int D,
                   float temperature, const int32_t* lengths, int T, vg_stream_t stream);
int vg_reparam_bwd(const float* dz, const float* dlog_q, const float* logstd, const float* eps, float* dmu,
                   float* dlogstd, int M, int D, float temperature, const int32_t* lengths, int T,
                   vg_stream_t stream);
/* Prior log-density under the flow (models/speech/lvtr.py:182-191):
 *   log_p[m,d] = mask ? logdet_sum[m]/D - ls - 0.5 ln 2pi - 0.5 exp(-2 ls) (u - mu)^2 : 0
 * and the KL reduction (training_lib/losses.py:16-18,27 via trainers/speech/lvtr.py:122-124):
 *   kl_rows[m] = mean_d(log_q - log_p)  (0 on padded rows)
 * mu_ls: [M][ld_mu_ls] fp32 with mu in columns 0..D-1 and logstd in D..2D-1 (fused prior head). */
int vg_prior_logp_fwd(const float* mu_ls, int64_t ld_mu_ls, const float* u, const float* logdet_sum,
                      const float* log_q, float* log_p, float* kl_rows, int M, int D, const int32_t* lengths,
                      int T, vg_stream_t stream);
/* Backward of the pair (log_p, kl_rows): total dlog_p = dlog_p (may be NULL) - dkl_rows/D;
 * outputs d(mu_ls) [M][2D], du [M][D], dlogdet_sum [M], dlog_q [M][D] (= +dkl_rows/D, may be NULL). */
int vg_prior_logp_bwd(const float* dlog_p, const float* dkl_rows, const float* mu_ls, int64_t ld_mu_ls,
                      const float* u, float* dmu_ls, float* du, float* dlogdet_sum, float* dlog_q, int M, int D,
                      const int32_t* lengths, int T, vg_stream_t stream);

/* ---------------------------------------------------------------- reductions / casts
 * deterministic fp32 sum of n values -> out[0] (single block tree) */
int vg_sum_f32(const float* x, int64_t n, float* out, vg_stream_t stream);
/* column sums of a [M][N] matrix (dtype) -> fp32 [N]; bias gradients and the
 * second stage of the RMSNorm scale gradient.  ws: fp32 [vg_colsum_blocks(M)][N].
 * accumulate != 0: out[n] += sum (gradient accumulation straight into param.grad). */
int vg_colsum_blocks(int M);
int vg_colsum(const void* x, int M, int N, int64_t ld, float* ws, float* out, int dtype, int accumulate,
              vg_stream_t stream);
/* up to VG_COLSUM_MAX_TASKS independent fp32 column sums (dst[cols] (+)= sum over rows of src[rows][ld]) in one
 * launch: the partial-sum arrays behind the bias / norm-scale gradients of one backward node */
enum { VG_COLSUM_MAX_TASKS = 32 };
typedef struct vg_colsum_task {
  const float* src;
  int rows, cols;
  int64_t ld;
  float* dst;
  int accumulate;
} vg_colsum_task;
int vg_colsum_multi(const vg_colsum_task* tasks, int n, vg_stream_t stream);
/* Masked means of up to VG_MEAN_MAX_TASKS small fp32 row tensors in one launch: out[k] = sum over valid frames and
 * columns of x_k (or |x_k|) / (cols_k * valid frames) -- TensorMask.mean() (utils/tensormask.py:135-140) of the step's
 * monitors (models/speech/lvtr.py:210-224, trainers/speech/lvtr.py:131-145).  lengths may be NULL (every frame valid).
 * Two small launches (per-block sums, then one wave folds them): deterministic, no atomics. */
enum { VG_MEAN_MAX_TASKS = 8 };
typedef struct vg_mean_task {
  const float* src;       /* [M][ld] fp32 */
  int64_t ld;
  int32_t cols;           /* <= 64 */
  int32_t absolute;       /* 1: mean of |x| */
} vg_mean_task;
int vg_masked_means_blocks(int M);     /* partial: fp32 [vg_masked_means_blocks(M)][VG_MEAN_MAX_TASKS + 1] scratch */
int vg_masked_means(const vg_mean_task* tasks, int n, int M, const int32_t* lengths, int T, float* partial, float* out,
                    vg_stream_t stream);
/* first stage of up to VG_COLSUM_MAX_TASKS LARGE column sums in one launch (the bias gradients of one Transformer
 * layer: column sums of the incoming gradient, of the attention-output gradient and of dQKV): src is [rows][ld] of
 * `dtype` (passed through the float* field), dst receives nb rows of cols fp32 partial sums (nb as vg_colsum_blocks),
 * which vg_colsum_multi then folds together with the node's other partial arrays. */
int vg_colsum_partials_multi(const vg_colsum_task* tasks, int n, int nb, int dtype, vg_stream_t stream);
/* out[seg][cols] = sum over the `rows` rows of segment seg of x[nseg * rows][ld] (the time-embedding gradient of a
 * conv block: the block's input gradient summed over the frames of each sequence; `.float().sum(1)` on a [B, T, C]
 * view in the reference-shaped code).  part: nb * nseg * cols floats of scratch, nb <= 64 row blocks per segment. */
int vg_colsum_segments(const void* x, int nseg, int rows, int cols, int64_t ld, float* part, int nb, float* out, int dtype,
                       vg_stream_t stream);
/* the same over ragged segments laid end to end (packed rows): segment s = rows [cu_rows[s], cu_rows[s + 1]) */
int vg_colsum_segments_cu(const void* x, const int32_t* cu_rows, int nseg, int cols, int64_t ld, float* part, int nb,
                          float* out, int dtype, vg_stream_t stream);
/* dx = dy * act'(aux): ReLU takes aux = activation output, GELU (erf) takes aux = pre-activation
 * (modules/activations.py:5-18 backward, for Linear+activation heads with several consumers). */
int vg_act_bwd(const void* dy, const void* aux, void* dx, int64_t n, int act, int dtype, vg_stream_t stream);
/* y = rows t < lengths[b] of x, zeros elsewhere (utils/tensormask.py:63-67 on [M][C] rows; backward of a
 * masked Linear when the incoming gradient is not already zero on padded frames) */
int vg_mask_rows(const void* x, void* y, int M, int C, const int32_t* lengths, int T, int dtype, vg_stream_t stream);
/* fp32 -> bf16 copy (weights shadow) */
int vg_cast_f32_to_bf16(const float* src, void* dst, int64_t n, vg_stream_t stream);

/* ---------------------------------------------------------------- conv-block row kernels (channels-last)
 * Depthwise conv along time (taps k = 0..taps-1 read input frame t + k - shift; shift = taps-1 causal,
 * 0 look-ahead; zero padding inside each sequence) + conv bias + per-sequence time embedding, followed by
 * the per-frame channel norm with UNBIASED variance -- the fused front half of the reference's
 * ResidualBlock / TemporalResidualBlock / TCResidualBlock (modules/conv/layers.py:114-124,243-251,
 * 277-286; modules/norm.py:43-47) on [M = B*T][C] rows.  taps = 0: the norm alone (final_norm).
 * w: fp32 [C][taps]; cbias, gamma, beta: fp32 [C]; temb: fp32 [B][C] or NULL; mean/rstd: fp32 [M].
 * Backward: du = dL/d(conv output) [M][C]; dx = dx_add + conv^T(du) (taps > 0); norm_part: fp32
 * [vg_dwnorm_blocks(M)][2][C] (gamma, beta partial sums), w_part: fp32 [blocks][C][taps]. */
int vg_dwnorm_blocks(int M);
int vg_dwnorm_fwd(const void* x, const float* w, const float* cbias, const float* temb, const float* gamma,
                  const float* beta, void* y, float* mean, float* rstd, int M, int C, int T, int taps, int shift,
                  float eps, int dtype, vg_stream_t stream);
int vg_dwnorm_bwd(const void* dy, const void* x, const float* w, const float* cbias, const float* temb,
                  const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du, void* dx,
                  float* norm_part, float* w_part, int M, int C, int T, int taps, int shift, int dtype,
                  vg_stream_t stream);
/* Packed rows (ragged batches without their padding; the reference pads, utils/helpers.py:80-135, and convolves the
 * padding, modules/conv/layers.py:70-135): the same two operators on sequences laid end to end -- sequence s = rows
 * [cu_rows[s], cu_rows[s + 1]), s < nseq <= 64, every sequence with its own zero padding on both sides; temb row
 * min(s, nbatch - 1) (sequences past nbatch are the zero-length pseudo sequences that cover a bucket's spare rows).
 * bf16, C = 512, taps = 7 only (every conv block of vae-gslm.yaml); other shapes are refused. */
int vg_dwnorm_fwd_seg(const void* x, const float* w, const float* cbias, const float* temb, const float* gamma,
                      const float* beta, void* y, float* mean, float* rstd, int M, int C, const int32_t* cu_rows, int nseq,
                      int nbatch, int taps, int shift, float eps, int dtype, vg_stream_t stream);
int vg_dwnorm_bwd_seg(const void* dy, const void* x, const float* w, const float* cbias, const float* temb,
                      const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du, void* dx,
                      float* norm_part, float* w_part, int M, int C, const int32_t* cu_rows, int nseq, int nbatch, int taps,
                      int shift, int dtype, vg_stream_t stream);
/* Round 6: the conditioning of a conv block merged into its first 1x1 convolution.  The reference concatenates the
 * condition channels to the normalised activations (modules/conv/layers.py:112-113, "concat" conditioning) and runs one
 * Conv1d(k=1) over C + cond channels; until round 5 this build ran a K = C product with a pre-activation operand that a
 * K = cond_cols (32) product had written first (65 MB out, 65 MB back in, two launches of 38 - 43 us for 2 GFLOP).
 * vg_dwnorm_fwd_cat writes y rows `ldy` elements apart (ldy >= C + 64) and appends to every row its condition channels
 * and zeros up to column C + 64 (cond: bf16 [M][ldcond], cond_cols <= 64, a multiple of 8; NULL: no tail), so the 1x1
 * convolution is ONE vg_gemm over K = C + 64 against the weight zero-padded to the same width, and its dgrad
 * (N = C + 64) yields d(norm output) and d(cond) side by side; vg_dwnorm_bwd_ld is vg_dwnorm_bwd / _seg reading the
 * incoming gradient rows `ldy` elements apart.  bf16, C = 512, 7 taps; cu_rows = NULL: M / T sequences of T rows, else
 * packed rows as in the _seg entry points. */
int vg_dwnorm_fwd_cat(const void* x, const float* w, const float* cbias, const float* temb, const float* gamma,
                      const float* beta, void* y, int64_t ldy, const void* cond, int64_t ldcond, int cond_cols,
                      float* mean, float* rstd, int M, int C, int T, const int32_t* cu_rows, int nseq, int nbatch,
                      int taps, int shift, float eps, int dtype, vg_stream_t stream);
int vg_dwnorm_bwd_ld(const void* dy, int64_t ldy, const void* x, const float* w, const float* cbias, const float* temb,
                     const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du, void* dx,
                     float* norm_part, float* w_part, int M, int C, int T, const int32_t* cu_rows, int nseq, int nbatch,
                     int taps, int shift, int dtype, vg_stream_t stream);
/* Round 6: the same backward (modules/conv/layers.py:93-110 + modules/norm.py:43-47 under autograd) in ONE launch: a block
 * owns 26 consecutive frames of one sequence, recomputes du for them and the 6 frames its dx needs beyond them, keeps du
 * in LDS and takes dx and the tap gradients from there (the two-launch form writes du to HBM and reads it back).  du and dx
 * are bitwise those of vg_dwnorm_bwd / _seg / _ld; norm_part [nblk][2C] and w_part [nblk][C * taps] have
 * nblk = vg_dwnorm_bwd_fused_blocks(nseq, max_len) rows (other frames per row than the two-launch form: same column sums up
 * to fp32 rounding).  du may be NULL (not stored); du_part (or NULL): fp32 [nblk][C], the column sums of each block's own du
 * rows as stored (bf16) -- blocks [s * nblk / nseq, (s + 1) * nblk / nseq) belong to sequence s, so the per-sequence sums the
 * time-embedding and conv-bias gradients need (modules/conv/layers.py:96) come from nblk small rows instead of a pass over du.  bf16, C = 512, 7 taps, 0 <= shift <= 6; cu_rows = NULL: nseq = M / T
 * sequences of T rows (max_len ignored), else packed rows, every sequence at most max_len rows. */
int vg_dwnorm_bwd_fused_blocks(int nseq, int max_len);
int vg_dwnorm_bwd_fused(const void* dy, int64_t ldy, const void* x, const float* w, const float* cbias, const float* temb,
                        const float* gamma, const float* mean, const float* rstd, const void* dx_add, void* du, void* dx,
                        float* norm_part, float* w_part, float* du_part, int M, int C, int T, const int32_t* cu_rows, int nseq,
                        int nbatch, int max_len, int taps, int shift, int dtype, vg_stream_t stream);

/* ---------------------------------------------------------------- autoregressive decode step
 * LVTR.step (models/speech/lvtr.py:227-286): one new frame per sequence.
 * vg_gemm_rows: y[M][N] = act(x[M][K] W[N][K]^T + bias) + residual for M <= 64 rows, in groups of 8 (HBM-bound on W; exact
 *   fp32 accumulation; replaces nn.Linear at modules/attention/attention.py:52,79,
 *   modules/transformer/layers.py:82, modules/linear/layers.py:192 on the decode path).
 *   x, W, residual in dtype; y in dtype or fp32 (out_f32); K, ldx, ldw multiples of 8.  norm_scale (fp32 [K]
 *   or NULL): fuse the preceding RMSNorm (modules/norm.py:28-32), y = act(rmsnorm(x; scale, eps) W^T + b) + r.
 * vg_embed_fuse: frame [B][ldf] = (token id as float, latent z); out[b] = E[id] + relu(Wf z + bf) in dtype
 *   (token_embedding + token_fuser, models/speech/lvtr.py:161-168).
 * vg_sample_token: categorical draw from softmax(logits / temperature) by inverse CDF with uniform[b] in
 *   [0,1); writes the id to frame[b][0] and, if pos != NULL, pos[b] += 1 (lvtr.py:276-284).
 * vg_attn_decode_append: qkv [B][3*H*64] of the new frame; the key/value rows are written into the
 *   pre-allocated caches [B][Tmax][H*64] at index pos[b], then the query attends over pos[b]+1 frames with
 *   the ALiBi bias of modules/position/alibi.py:9-33 (query position = last; attention.py:56-73).
 * vg_advance: pos[i] += by (device-side frame counter; keeps the step replayable from a hipGraph).
 * Fused layer path (round 3; the residual stream of the step is fp32 [B][d]):
 * vg_attn_layer_decode: the whole attention sub-layer of modules/transformer/layers.py:41-56 for one new frame in ONE
 *   launch (one block per head and sequence): RMSNorm (norm_scale, norm_eps) of x[b], this head's rows of the QKV
 *   projection (attention.py:52), cache append at pos[b] + attention over the cache (attention.py:56-77), this head's
 *   64-column band of the out-projection (attention.py:79) ADDED to x1[b] with fp32 atomics; the h = 0 block adds
 *   x[b] + bo.  x1 must be zero on entry (accumulation target); zero_buf (fp32 [B][d] or NULL) is cleared by this
 *   launch -- pass the buffer the next layer accumulates into.  d = 64 H, a multiple of 256, at most 1024; weights and
 *   caches in dtype, biases / slopes / norm_scale fp32.  The sum order of the H contributions, and with it the last bit
 *   of x1, varies from run to run.
 * vg_gemm_rows_mixed: vg_gemm_rows with fp32 input rows and fp32 residual (the fused path's residual stream) against
 *   weights in dtype; zero_buf / zero_n: an fp32 buffer this launch clears (or NULL / 0).
 */
/* vg_decode_noise: the random draws of one decode step in one launch, from a counter-based generator (Philox 4x32-10) keyed
 *   by (seed, epoch[0], b, pos[b]): normal[b][0..n_normal) ~ N(0, 1) (Box-Muller), uniform[b] in [0, 1).  pos is the
 *   device-side frame counter, so a replayed hipGraph draws fresh numbers (replaces torch.randn / torch.rand of
 *   lvtr.py:262-284 on that path); epoch (one device int32, or NULL = 0) is read by the launch, so the caller bumps it
 *   whenever pos is rewound (a new prompt on the same session) and the replayed graph leaves the old stream of draws. */
int vg_decode_noise(uint64_t seed, const int32_t* pos, const int32_t* epoch, float* normal, int n_normal, float* uniform,
                    int B, vg_stream_t stream);
int vg_attn_layer_decode(const float* x, const float* norm_scale, float norm_eps, const void* wqkv, const float* bqkv,
                         const void* wo, const float* bo, void* kcache, void* vcache, const float* slopes,
                         const int32_t* pos, float* x1, float* zero_buf, int B, int Tmax, int H, int dtype,
                         vg_stream_t stream);
int vg_gemm_rows_mixed(const float* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const float* residual,
                       int64_t ldr, void* y, int64_t ldy, int M, int N, int K, int act, int out_f32,
                       const float* norm_scale, float norm_eps, float* zero_buf, int zero_n, int dtype, vg_stream_t stream);
int vg_gemm_rows(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const void* residual,
                 int64_t ldr, void* y, int64_t ldy, int M, int N, int K, int act, int out_f32,
                 const float* norm_scale, float norm_eps, int dtype, vg_stream_t stream);
/* vg_gemm_rows_acc (round 5): y[M][N] (fp32, ZERO on entry) += x[M][K] W[N][K]^T + bias + residual for bf16 x / W and an
 * fp32 residual, the reduction split over `splits` groups of blocks that meet in y through fp32 atomics: the two
 * N = d_model products of a decode layer (modules/attention/attention.py:79, modules/transformer/layers.py:82-86) at the
 * reference's inference batch, where one block per 16 columns would pull every input row through a single CU.  zero_buf
 * (zero_n floats, not y): cleared by the launch -- the accumulator of a LATER launch.
 * NOT bitwise repeatable: the K slices meet in y through fp32 atomics, whose order varies from run to run, so
 * two runs of one session differ in the last bits of the residual stream and a near-tie of the token draw can flip.
 * vg_gemm_rows / vg_gemm_rows_mixed are repeatable; a decode session uses them for these products with VG_DECODE_ACC=0
 * (inference/speech/session.py, "reproducible mode").  Also: with bf16 weights and M >= VG_ROWS_MFMA rows (default 1),
 * vg_gemm_rows_mixed rounds its fp32 input rows -- times the norm scale -- to bf16 for the matrix-core kernel before the
 * product (VG_ROWS_MFMA=17 keeps fp32 input rows on the exact dot-product kernel up to 16 rows). */
int vg_gemm_rows_acc(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, const float* residual,
                     int64_t ldr, float* y, int64_t ldy, int M, int N, int K, int splits, float* zero_buf, int zero_n,
                     vg_stream_t stream);
int vg_attn_decode_append(const void* qkv, void* kcache, void* vcache, void* out, const float* slopes,
                          const int32_t* pos, int B, int Tmax, int H, int dtype, vg_stream_t stream);
int vg_advance(int32_t* pos, int n, int by, vg_stream_t stream);
/* Read `bytes` (a multiple of 16, 16-byte aligned) with `blocks` narrow workgroups and discard them: brings a weight
 * range into the Infinity Cache ahead of the latency-bound kernels of the decode step that stream it (lab switch of
 * inference/speech/session.py, VG_DECODE_PREFETCH; reference loop: trainers/speech/sampler.py:50-62). */
int vg_touch(const void* ptr, int64_t bytes, int blocks, vg_stream_t stream);
int vg_embed_fuse(const float* frame, int ldf, const float* emb, int vocab, int E, const float* wf, const float* bf,
                  int latent, void* out, int B, int dtype, vg_stream_t stream);
int vg_sample_token(const float* logits, int V, float temperature, const float* uniform, float* frame, int ldf,
                    int32_t* pos, int B, vg_stream_t stream);

/* ---------------------------------------------------------------- optimizer
 * AdamW (torch.optim.AdamW semantics: decoupled weight decay, bias correction; reference
 * training_lib/optimizer.py:18-25, stepped at trainers/speech/lvtr.py:150-157) over one flat gradient
 * bucket: param / grad / exp_avg / exp_avg_sq are fp32 [n] with the SAME layout, n a multiple of 256 and
 * every parameter starting on a 256-element boundary; group_of_chunk[n/256] selects the parameter group
 * (lr[g], weight_decay[g], host arrays, ngroups <= 4) of each chunk.  In the same pass: shadow_bf16 (or
 * NULL) receives the bf16 copy of the updated parameters, grad is multiplied by *grad_scale (device
 * scalar, e.g. a clipping coefficient, or NULL) before use and cleared afterwards if zero_grad.
 * step is the 1-based optimizer step (bias corrections are computed on the host).
 */
int vg_adamw(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
             const uint8_t* group_of_chunk, int64_t n, const float* lr, const float* weight_decay, int ngroups,
             float beta1, float beta2, float eps, int step, const float* grad_scale, int zero_grad,
             vg_stream_t stream);

/* ---------------------------------------------------------------- coupling flow on the latent
 * The conditional affine-coupling stack of the prior (modules/flow/layers.py:15-98 LinearCoupling with
 * flip = true, :199-245 the stack; models/speech/lvtr.py:182-191 call site) as one row kernel per direction:
 * latent dim 4, hidden 64, LayerNorm(eps) -> FiLM -> erf-GELU -> Linear(64->4), log-scale =
 * log(sigmoid(.) * (hi - lo) + lo).  All tensors fp32.
 *   z, u: [M][4]; wb: [M][ldw] FiLM rows, layer l uses columns l*128 .. l*128+63 (scale) and +64.. (shift);
 *   params: [L][580] packed per layer as W1[64][2], b1[64], ln_weight[64], ln_bias[64], W2[4][64], b2[4];
 *   logdet_sum: [M] sum of the log-scales over layers and dims (0 on frames t >= lengths[b]);
 *   states: [M][L][4] layer inputs saved for the backward (NULL to skip, e.g. inference).
 * Backward: dparams_partial is [vg_flow_blocks(M)][L*580] per-block sums (reduce with vg_colsum).
 */
int vg_flow_blocks(int M);
int vg_flow_fwd(const float* z, const float* wb, int64_t ldw, const float* params, int L, float* u,
                float* logdet_sum, float* states, int M, float eps, float hi, float lo,
                const int32_t* lengths, int T, vg_stream_t stream);
/* reverse: z rows are written with stride ldz (>= 4); with mu_ls != NULL (rows [mean(4) | logstd(4)], stride
 * ld_mu_ls) u is unit noise and the kernel first draws mean + temperature * exp(logstd) * u. */
int vg_flow_reverse(const float* u, const float* wb, int64_t ldw, const float* params, int L, float* z, int64_t ldz,
                    int M, float eps, float hi, float lo, const float* mu_ls, int64_t ld_mu_ls, float temperature,
                    vg_stream_t stream);
int vg_flow_bwd(const float* states, const float* wb, int64_t ldw, const float* params, int L,
                const float* du, const float* dlogdet_sum, float* dz, float* dwb, float* dparams_partial,
                int M, float eps, float hi, float lo, const int32_t* lengths, int T, vg_stream_t stream);

/* ---------------------------------------------------------------- input fusion of the training step
 * out[m][E] (fp32) = mask(E[ids[m]]) + relu(Wf z[m] + bf): Embedding.forward (modules/linear/layers.py:150-152,
 * masked) + token_fuser Linear(D -> E) + ReLU (:184-193, not masked) + LVTR.fuse_inputs
 * (models/speech/lvtr.py:390-392).  ids int64 [M], z fp32 [M][ldz] (D <= 8 columns used), emb [vocab][E],
 * Wf [E][D], bf [E] or NULL.  Backward: demb[vocab][E] += mask(dout) (fp32 atomics; NULL to skip),
 * dz[m][0..D) = Wf^T (dout * (pre > 0)) (NULL to skip), part[vg_embed_fuse_blocks(M)][E][D+1] = per-block
 * partial sums of (dWf | dbf), to be folded by vg_colsum. */
int vg_embed_fuse_blocks(int M);
int vg_embed_fuse_fwd(const int64_t* ids, const float* z, int64_t ldz, const float* emb, int vocab, int E,
                      const float* wf, const float* bf, int D, const int32_t* lengths, int T, float* out, int M,
                      vg_stream_t stream);
int vg_embed_fuse_bwd(const float* dout, const int64_t* ids, const float* z, int64_t ldz, int vocab, int E,
                      const float* wf, const float* bf, int D, const int32_t* lengths, int T, float* demb, float* dz,
                      int64_t lddz, float* part, int M, vg_stream_t stream);

/* ---------------------------------------------------------------- diffusion-decoder loss arithmetic
 * GaussianDiffusion1D.q_sample / p_losses (modules/diffusion/ddpm.py:337-366) with the masked L1 of
 * training_lib/losses.py:9-27,44-57 (objective pred_noise, loss_type l1).  Rows are frames (b, t), b = m / T.
 * vg_qsample: x_t = mask(coef_x0[t_b] x0 + coef_noise[t_b] noise), fp32 [M][C]; t int64 [M / T].
 * vg_l1_rows_fwd: rows[m] = mask(mean_c |pred - target|) (sum them with vg_sum_f32); pred in dtype, target fp32.
 * vg_l1_rows_bwd: dpred = mask(gscale[0] sign(pred - target) / C) in dtype; gscale is a device scalar. */
int vg_qsample(const float* x0, const float* noise, const float* coef_x0, const float* coef_noise, const int64_t* t,
               const int32_t* lengths, int T, float* out, int M, int C, vg_stream_t stream);
int vg_l1_rows_fwd(const void* pred, const float* target, const int32_t* lengths, int T, float* rows, int M, int C,
                   int dtype, vg_stream_t stream);
int vg_l1_rows_bwd(const void* pred, const float* target, const float* gscale, const int32_t* lengths, int T,
                   void* dpred, int M, int C, int dtype, vg_stream_t stream);

/* ---------------------------------------------------------------- channel norm of narrow rows
 * y = act(gamma (x - mean) rstd + beta) per frame over C channels, UNBIASED variance (reference modules/norm.py:35-47)
 * with an optional fused ReLU (ConvNormAct, modules/conv/layers.py:543-560), for widths the vg_dwnorm_* kernels do not
 * take (2 <= C <= 1024; the utterance encoder's 128 / 256-channel layers).  x, y, dy, dx [M][C] in dtype; mean, rstd
 * fp32 [M].  Backward: dx and part[vg_chnorm_blocks(M)][2 C] = per-block partial sums of (dgamma | dbeta). */
int vg_chnorm_blocks(int M);
int vg_chnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M, int C,
                  float eps, int relu, int dtype, vg_stream_t stream);
int vg_chnorm_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* mean, const float* rstd,
                  void* dx, float* part, int M, int C, int relu, int dtype, vg_stream_t stream);

/* ---------------------------------------------------------------- strided-conv window gather
 * Conv1d(k, stride, padding) on channels-last rows = window gather + one vg_gemm (ConvNormAct of the utterance
 * encoder, modules/conv/layers.py:543-560): rows[b][to][tap][c] = x[b][to*stride + tap - pad_left][c] (zero outside
 * the sequence), rows viewed as [B*t_out][k*C]; vg_conv_scatter is the adjoint (dx from drows, gather form). */
int vg_conv_gather(const void* x, void* rows, int B, int T, int C, int t_out, int k, int stride, int pad_left, int dtype,
                   vg_stream_t stream);
int vg_conv_scatter(const void* drows, void* dx, int B, int T, int C, int t_out, int k, int stride, int pad_left,
                    int dtype, vg_stream_t stream);

/* ---------------------------------------------------------------- gradient exchange (RCCL)
 * The one collective of the path: the mean of the gradients over the data-parallel ranks, which the reference
 * gets from Lightning's DDP wrapper (training_lib/trainer.py:37-65 builds the strategy, the reduce happens inside
 * manual_backward at trainers/speech/lvtr.py:144).  One process per GPU, one communicator per process.  Rank 0
 * draws an id with vg_comm_unique_id and hands its VG_COMM_ID_BYTES bytes to the other ranks by any side channel
 * (a torch.distributed store, a file, MPI); every rank then calls vg_comm_init with its current HIP device set.
 * vg_allreduce_bucket reduces one flat gradient bucket IN PLACE on comm_stream (sum, or mean with average=1) and
 * returns immediately; ordering against the producers / consumers of the bucket is by stream, like every other
 * launch of this library.  RCCL is bound at run time (dlopen), so the library loads on a box without it. */
enum { VG_COMM_ID_BYTES = 128 };
int vg_comm_unique_id(void* out, int nbytes);
int vg_comm_init(int rank, int world, const void* unique_id, int nbytes);
int vg_comm_world(void);       /* ranks of the live communicator, 0 if none */
int vg_allreduce_bucket(void* buf, int64_t n, int dtype, int average, vg_stream_t comm_stream);
int vg_comm_destroy(void);

/* ---------------------------------------------------------------- measurement hooks
 * Optional HIP-event timing of the GEMM / attention launches (bench.py's roofline
 * figure).  vg_prof_enable(1) clears and starts recording, vg_prof_enable(0)
 * stops; vg_prof_read() synchronises and returns, for one kind, the summed
 * launch durations (ms), the summed ALGORITHMIC work (FLOPs) and the launch count. */
enum { VG_PROF_GEMM_BF16_NT = 0, VG_PROF_GEMM_BF16_NN = 1, VG_PROF_GEMM_BF16_TN = 2,
       VG_PROF_GEMM_F32 = 3, VG_PROF_ATTN_FWD = 4, VG_PROF_ATTN_BWD = 5,
       /* HBM-bound row kernels: `work` is their ALGORITHMIC byte count */
       VG_PROF_RMSNORM_FWD = 6, VG_PROF_RMSNORM_BWD = 7, VG_PROF_ADAMW = 8,
       VG_PROF_DWNORM_FWD = 9, VG_PROF_DWNORM_BWD = 10, VG_PROF_KINDS = 11 };
int vg_prof_enable(int on);
int vg_prof_read(int kind, double* total_ms, double* total_work, int* launches);
/* Scope tags: vg_prof_tag(t) marks every launch recorded from now on with t and returns the previous tag (the layer
 * function brackets its forward and backward with VG_PROF_TAG_LAYER, so bench.py can report the north_star's
 * "attention + FFN path" -- the Transformer layers' GEMMs and attention kernels -- apart from the conv stacks and
 * heads); vg_prof_read_tag() is vg_prof_read() restricted to one tag. */
enum { VG_PROF_TAG_NONE = 0, VG_PROF_TAG_LAYER = 1 };
int vg_prof_tag(int tag);
int vg_prof_read_tag(int kind, int tag, double* total_ms, double* total_work, int* launches);
/* summed ALGORITHMIC bytes (operands and results once each) of the recorded launches of one kind (GEMM kinds) */
int vg_prof_read_bytes(int kind, double* total_bytes);

/* Peak probes for the box the measurement runs on ("measured peaks, do not hard-code"); the caller
 * times the launch with events on `stream`.
 *   vg_probe_mfma: `blocks` x 256 threads, every wave issues 4 independent chains of `iters` x 4
 *                  v_mfma_f32_32x32x16_bf16 from registers = blocks * 4 waves * iters * 4 * 32768 FLOP;
 *                  `out` takes blocks * 256 floats (keeps the chains alive).
 *   vg_probe_copy: streaming copy of `bytes` (multiple of 16) with 16-byte accesses = 2 * bytes of HBM traffic. */
int vg_probe_mfma(float* out, int blocks, int iters, vg_stream_t stream);
int vg_probe_copy(const void* src, void* dst, int64_t bytes, int blocks, vg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VAEGSLM_HIP_H */
