// Kernel-side parameter block shared by the GEMM variants (vg_gemm.hip, vg_gemm_dma.hip).
#pragma once
#include <hip/hip_runtime.h>

struct GemmParams {
  const void* A; const void* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  const float* bias;        // [N] fp32 or null
  const void* residual;     // [M][ldc] (type T) or null
  const void* aux_in;       // [M][ldc] (type T): input of the activation derivative
  void* aux_out;            // [M][ldc] (type T): pre-activation copy
  const void* pre_add;      // [M][ldc] (type T) added BEFORE the activation, or null
  const int* lengths; int T;
  int act;                  // VG_ACT_*
  int dact;                 // VG_ACT_* derivative applied to the result (uses aux_in)
  int out_f32;              // C is fp32 regardless of T
  int accumulate;           // C += result (fp32 C only)
  int k_per_split;          // K range handled by one blockIdx.z (multiple of BK)
  float alpha;
  float* colsum_out;        // TN mode: [M] fp32, += sum_k A(m,k) (unscaled), or null
  float* colpart;           // [tiles_m][N] fp32: per-m-tile column sums of the stored result (bias gradient of the
                            // NEXT layer's Linear produced by the dgrad that writes its input gradient), or null
  float* split_ws;          // split-K slabs [splits][tiles][BM*BN] fp32 (in-launch reduction), or null: fp32 atomics
  int* split_cnt;           // [tiles] arrival counters, zeroed ahead of the launch
  int colsum_rr;            // 1: deal the row-sum MFMAs round-robin over the blocks of an m-panel
  int group_m;              // > 0: tiles are walked m-fastest inside bands of group_m row-tiles (L2 blocking)
  int m_base;               // rows of the whole product above this launch's first row (0 unless vg_gemm split the rows over two
                            // launches): only the row mask needs it (sequence index and frame of a row)
  int aux_ring;             // EPI_DACT8 on the long-phase 256 x 256 schedule: the 8-bit derivative tile arrives through the LDS-DMA
                            // ring as "K tile nkt" instead of by global loads at the end of the main loop (VG_AUX_RING=0: off)
  int aux_u8;               // the stored derivative (aux_out of GELU | SAVE_DERIV, aux_in of dact = STORED) is uint8 [M][ldc]:
                            // VG_ACT_DERIV_U8 (the flag itself is stripped from act / dact)
};

namespace vg_host {
// LDS-DMA pipelined bf16 variant; returns 0 if launched, -1 if not applicable
int gemm_dma_launch(const GemmParams& p, int a_tr, int b_tr, int cfg, int splits, hipStream_t stream);
// phase-pipelined 256x256 tile (cfg 11: ring schedule, 12: complementary schedule); same return convention
int gemm_ph_launch(const GemmParams& p, int a_tr, int b_tr, int cfg, int splits, hipStream_t stream);
// grouped TN products on the ring schedule (weight gradients)
int gemm_group_launch(const GemmParams* ps, const int* splits, int n, hipStream_t stream);
}
