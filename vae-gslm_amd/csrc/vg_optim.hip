// AdamW over a flat gradient bucket in one launch (reference
// training_lib/optimizer.py:18-25 builds torch.optim.AdamW with two parameter groups,
// call site trainers/speech/lvtr.py:150-157).
//
// Parameters, both moments and the gradient bucket share one layout (training_lib/dp.py):
// every parameter starts on a 256-element boundary, so a wave owns one 256-element chunk
// (4 floats per lane, 16-byte accesses on all seven streams) and the chunk's parameter-group
// id selects lr / weight decay.  The same pass writes the bf16 copy of the updated weights
// that the MFMA GEMMs consume and clears the gradient for the next accumulation window:
// 28 B read+written per parameter instead of the separate optimizer, cast and zero passes.
#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

using namespace vg;

namespace {

struct AdamGroups {
  float lr[4];
  float wd[4];
};

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow,
                                                    const unsigned char* __restrict__ group_of_chunk, long nchunks,
                                                    AdamGroups G, float beta1, float beta2, float eps, float inv_bc1,
                                                    float inv_sqrt_bc2, const float* __restrict__ grad_scale,
                                                    int zero_grad) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  const float gs = grad_scale ? *grad_scale : 1.0f;
  for (long c = wave; c < nchunks; c += nwaves) {
    const int gi = group_of_chunk[c] & 3;
    const float lr = G.lr[gi], decay = 1.0f - lr * G.wd[gi], step = lr * inv_bc1;
    const long i = c * 256 + lane * 4;
    // the seven fp32 streams are touched once per step: non-temporal loads / stores (338 -> 321 us per 57 M-parameter
    // bucket in the step, round 4; the same switch on the row kernels -- RMSNorm, channel norms -- measured 0.4 % slower
    // end to end and was not kept).  The bf16 copy, which the next step's GEMMs read, keeps the default policy.
#ifndef VG_ADAM_TEMPORAL
#define VG_AD_LD(ptr) __builtin_nontemporal_load(ptr)
#define VG_AD_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define VG_AD_LD(ptr) (*(ptr))
#define VG_AD_ST(ptr, val) (*(ptr) = (val))
#endif
    f32x4 pv = VG_AD_LD(reinterpret_cast<const f32x4*>(p + i));
    f32x4 gv = VG_AD_LD(reinterpret_cast<const f32x4*>(g + i));
    f32x4 mv = VG_AD_LD(reinterpret_cast<const f32x4*>(m + i));
    f32x4 vv = VG_AD_LD(reinterpret_cast<const f32x4*>(v + i));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gv[e] * gs;
      mv[e] = fmaf(beta1, mv[e], (1.0f - beta1) * ge);
      vv[e] = fmaf(beta2, vv[e], (1.0f - beta2) * ge * ge);
      const float denom = fmaf(sqrtf(vv[e]), inv_sqrt_bc2, eps);
      pv[e] = fmaf(-step, mv[e] / denom, pv[e] * decay);
    }
    VG_AD_ST(reinterpret_cast<f32x4*>(p + i), pv);
    VG_AD_ST(reinterpret_cast<f32x4*>(m + i), mv);
    VG_AD_ST(reinterpret_cast<f32x4*>(v + i), vv);
    if (shadow) {
      bf16x4 sv = {(bf16_t)pv[0], (bf16_t)pv[1], (bf16_t)pv[2], (bf16_t)pv[3]};
      *reinterpret_cast<bf16x4*>(shadow + i) = sv;
    }
    if (zero_grad) VG_AD_ST(reinterpret_cast<f32x4*>(g + i), (f32x4{0.f, 0.f, 0.f, 0.f}));
  }
}

}  // namespace

extern "C" int vg_adamw(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                        const uint8_t* group_of_chunk, int64_t n, const float* lr, const float* weight_decay,
                        int ngroups, float beta1, float beta2, float eps, int step, const float* grad_scale,
                        int zero_grad, hipStream_t stream) {
  VG_REQUIRE(n > 0 && n % 256 == 0, "vg_adamw: n=%ld must be a positive multiple of 256", (long)n);
  VG_REQUIRE(ngroups >= 1 && ngroups <= 4 && step >= 1, "vg_adamw: ngroups=%d step=%d", ngroups, step);
  VG_REQUIRE(((uintptr_t)param % 16) == 0 && ((uintptr_t)grad % 16) == 0 && ((uintptr_t)exp_avg % 16) == 0 &&
                 ((uintptr_t)exp_avg_sq % 16) == 0 && ((uintptr_t)shadow_bf16 % 8) == 0,
             "vg_adamw: unaligned buffers");
  AdamGroups G;
  for (int i = 0; i < 4; ++i) {
    G.lr[i] = i < ngroups ? lr[i] : 0.f;
    G.wd[i] = i < ngroups ? weight_decay[i] : 0.f;
  }
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const long nchunks = n / 256;
  long blocks = (nchunks + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  // 28 B per parameter: p, g, m, v read; p, m, v written (16 + 12 B) + the bf16 copy (2 B) + the cleared gradient (4 B)
  const int tok = vg_host::prof_begin(VG_PROF_ADAMW, (double)n * (28.0 + (shadow_bf16 ? 2.0 : 0.0) + (zero_grad ? 4.0 : 0.0)), stream);
  adamw_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(param, grad, exp_avg, exp_avg_sq, (bf16_t*)shadow_bf16,
                                                                group_of_chunk, nchunks, G, beta1, beta2, eps,
                                                                (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale,
                                                                zero_grad);
  vg_host::prof_end(tok, stream);
  return vg_host::check_launch("vg_adamw");
}
