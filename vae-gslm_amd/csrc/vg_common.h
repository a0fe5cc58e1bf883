// Shared device-side building blocks for the gfx950 (MI355X / CDNA4) kernels.
//
// Everything here is written for wave64 + the 32x32 MFMA family:
//   * bf16 : v_mfma_f32_32x32x16_bf16  (8 bf16 per lane per operand, 16-deep k-step)
//   * f32  : v_mfma_f32_32x32x2_f32    (1 f32 per lane per operand, 2-deep k-step,
//            bit-exact fmaf chain -> used by the fp32 parity path)
// Both share ONE accumulator layout (16 f32 registers per lane):
//     col = lane & 31,  row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5),  i = 0..15
// so every kernel body (epilogues, online softmax, masking) is written once and
// templated on the storage type.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define VG_DEVICE __device__ __forceinline__
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

namespace vg {

// ---------------------------------------------------------------- accumulator map
VG_DEVICE int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }
VG_DEVICE int acc_col(int lane) { return lane & 31; }

VG_DEVICE f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// ---------------------------------------------------------------- scalar conversions
template <typename T> VG_DEVICE float to_f32(T v);
template <> VG_DEVICE float to_f32<float>(float v) { return v; }
template <> VG_DEVICE float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> VG_DEVICE T from_f32(float v);
template <> VG_DEVICE float from_f32<float>(float v) { return v; }
template <> VG_DEVICE bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---------------------------------------------------------------- per-type traits
template <typename T> struct Traits;

template <> struct Traits<bf16_t> {
  typedef bf16x8 Frag;                 // one MFMA operand fragment
  static constexpr int KSTEP = 16;     // reduction depth of one MFMA
  static constexpr int VEC = 8;        // elements per 16-byte vector
  static VG_DEVICE f32x16 mfma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <> struct Traits<float> {
  typedef float Frag;
  static constexpr int KSTEP = 2;
  static constexpr int VEC = 4;
  static VG_DEVICE f32x16 mfma(Frag a, Frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

// ---------------------------------------------------------------- LDS tile images
//
// RowTile<T, KD>: a [rows][KD] tile whose reduction index k is contiguous
// ("K-contiguous": activations [M][K], nn.Linear weights [N][K], K/Q/V rows
// [t][d]).  An MFMA operand fragment is 8 consecutive k (bf16) / one k (f32)
// of one row.
//   bf16: KD must be 64 -> 128-byte rows; the 16-byte chunk index is XORed
//         with (row >> 1) & 7 so each ds_read_b128 lane group touches 16
//         distinct 16-byte slots of the 256-byte bank row (conflict-free).
//   f32 : pitch KD + 1 floats (odd pitch -> conflict-free ds_read_b32 column reads).
template <typename T, int KD> struct RowTile;

template <int KD> struct RowTile<bf16_t, KD> {
  static_assert(KD == 64, "bf16 RowTile is specialised for 128-byte rows");
  static VG_DEVICE int bytes(int rows) { return rows * 128; }
  // byte offset of 16-byte chunk c16 (0..7) of `row`
  static VG_DEVICE int chunk_off(int row, int c16) { return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4); }
  // store 8 elements (one 16-byte vector) fetched from global
  static VG_DEVICE void store_vec(char* base, int row, int c16, uint4 v) {
    *reinterpret_cast<uint4*>(base + chunk_off(row, c16)) = v;
  }
  // fragment of k-step s (16 deep): this lane's row `row`, k = 16 s + 8 h + j
  static VG_DEVICE bf16x8 frag(const char* base, int row, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(base + chunk_off(row, 2 * s + (lane >> 5)));
  }
  // 16x16x32 operand fragment of k-step s (32 deep): row = row0 + (lane & 15), k = 32 s + 8 (lane >> 4) + j
  static VG_DEVICE bf16x8 frag16(const char* base, int row0, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(base + chunk_off(row0 + (lane & 15), 4 * s + (lane >> 4)));
  }
  // TRANSPOSED fragment out of the same image (the tile's ROW is the reduction index: what TrTile<bf16, 64>::frag
  // reads from its own image, same element order): ds_read_b64_tr_b16 only needs each lane's four elements to be
  // contiguous, which any 16-byte-chunk swizzle keeps.  Rows q and q + 2 of a 16-lane group fall on the same banks
  // (2-way conflict, twice the LDS cycles of TrTile) -- the price of not staging a second copy of the tile.
  template <bool PERM>
  static VG_DEVICE bf16x8 tr_frag(const char* base, int k0, int col0, int s, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = g >> 1;
    const int col = col0 + 16 * (g & 1) + 4 * p;
    const int ka = k0 + 16 * s + (PERM ? 4 * h : 8 * h) + q;
    const int kb = ka + (PERM ? 8 : 4);
    const int oa = ka * 128 + ((((col >> 3) ^ ((ka >> 1) & 7))) << 4) + ((col & 7) << 1);
    const int ob = kb * 128 + ((((col >> 3) ^ ((kb >> 1) & 7))) << 4) + ((col & 7) << 1);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + oa));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + ob));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
};

template <int KD> struct RowTile<float, KD> {
  static constexpr int PITCH = KD + 1;
  static VG_DEVICE int bytes(int rows) { return rows * PITCH * 4; }
  static VG_DEVICE void store_vec(char* base, int row, int c16, uint4 v) {
    float* p = reinterpret_cast<float*>(base) + row * PITCH + c16 * 4;
    p[0] = __uint_as_float(v.x); p[1] = __uint_as_float(v.y);
    p[2] = __uint_as_float(v.z); p[3] = __uint_as_float(v.w);
  }
  // fragment of k-step s (2 deep): k = 2 s + h
  static VG_DEVICE float frag(const char* base, int row, int s, int lane) {
    return reinterpret_cast<const float*>(base)[row * PITCH + 2 * s + (lane >> 5)];
  }
};

// TrTile<T, COLS>: a [krows][COLS] tile whose reduction index k is the ROW
// (the operand's own row/col index is contiguous): dY and X in the weight-
// gradient GEMM, W [N][K] in the data-gradient GEMM, V (and K, Q, dO) in the
// attention products that sum over time.  bf16 fragments are fetched with
// ds_read_b64_tr_b16 (hardware transpose, 4 k x 16 columns per 16-lane group).
//   bf16, COLS = 128: 256-byte rows, 64-byte granule index ^= (krow & 3)
//   bf16, COLS =  64: 128-byte rows, 64-byte granule index ^= (krow >> 1) & 1
//   -> the 4 consecutive k-rows x 64 bytes a 32-lane half reads tile one
//      256-byte bank row exactly (conflict-free).
//   f32: plain [krows][COLS] (lanes read 32 consecutive floats).
template <typename T, int COLS> struct TrTile;

template <int COLS> struct TrTile<bf16_t, COLS> {
  static_assert(COLS == 128 || COLS == 64, "bf16 TrTile supports 128 or 64 columns");
  static constexpr int PITCH = COLS * 2;
  static VG_DEVICE int bytes(int krows) { return krows * PITCH; }
  static VG_DEVICE int swz(int krow) { return COLS == 128 ? (krow & 3) : ((krow >> 1) & 1); }
  // byte offset of element (krow, col); col multiple of 4 for vector access
  // COLS = 128 additionally XORs the 32-byte half inside a granule with bit 3 of krow, so that the
  // two 16-lane groups of a half-wave that read the SAME 16 columns 8 k-rows apart (the
  // 16x16x32 operand pattern) also land on different banks.
  static VG_DEVICE int off(int krow, int col) {
    if constexpr (COLS == 128)
      return krow * PITCH + ((((col >> 5) ^ (krow & 3))) << 6) + ((((col >> 4) & 1) ^ ((krow >> 3) & 1)) << 5) +
             ((col & 15) << 1);
    else
      return krow * PITCH + ((((col >> 5) ^ swz(krow))) << 6) + ((col & 31) << 1);
  }
  // 16x16x32 operand fragment (natural k order): element j <-> k = 32 s + 8 (lane >> 4) + j,
  // operand row/col index = col0 + (lane & 15)
  static VG_DEVICE bf16x8 frag16(const char* base, int k0, int col0, int s, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int col = col0 + 4 * p;
    const int ka = k0 + 32 * s + 8 * g + q;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + off(ka, col)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + off(ka + 4, col)));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
  static VG_DEVICE void store_vec(char* base, int krow, int c16, uint4 v) {
    *reinterpret_cast<uint4*>(base + off(krow, c16 * 8)) = v;
  }
  // fragment for k-step s: element j <-> k = kperm(s, h, j), operand index = col0 + (lane & 31)
  //   natural order  : k = 16 s + 8 h + j                    (PERM = false)
  //   accumulator order: k = 16 s + 8 (j >> 2) + 4 h + (j & 3) (PERM = true; pairs with a
  //                      32x32 accumulator tile used as the other operand)
  template <bool PERM>
  static VG_DEVICE bf16x8 frag(const char* base, int k0, int col0, int s, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = g >> 1;
    const int col = col0 + 16 * (g & 1) + 4 * p;
    const int ka = k0 + 16 * s + (PERM ? 4 * h : 8 * h) + q;
    const int kb = ka + (PERM ? 8 : 4);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + off(ka, col)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + off(kb, col)));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
};

template <int COLS> struct TrTile<float, COLS> {
  static constexpr int PITCH = COLS;   // floats
  static VG_DEVICE int bytes(int krows) { return krows * PITCH * 4; }
  static VG_DEVICE void store_vec(char* base, int krow, int c16, uint4 v) {
    *reinterpret_cast<uint4*>(base + (krow * PITCH + c16 * 4) * 4) = v;
  }
  // k-step s (2 deep).  natural: k = 2 s + h ; accumulator order: k = acc_row(s, lane)
  template <bool PERM>
  static VG_DEVICE float frag(const char* base, int k0, int col0, int s, int lane) {
    const int k = k0 + (PERM ? acc_row(s, lane) : 2 * s + (lane >> 5));
    return reinterpret_cast<const float*>(base)[k * PITCH + col0 + (lane & 31)];
  }
};

// A 32x32 accumulator tile X (rows = reduction index) as the B operand of the
// next product (Y = A . X).  bf16: registers 8s..8s+7 packed; f32: register s.
template <typename T> struct AccOperand;
template <> struct AccOperand<bf16_t> {
  static constexpr int STEPS = 2;
  static VG_DEVICE bf16x8 get(const f32x16& x, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)x[8 * s + j];
    return r;
  }
};
template <> struct AccOperand<float> {
  static constexpr int STEPS = 16;
  static VG_DEVICE float get(const f32x16& x, int s) { return x[s]; }
};

// ---------------------------------------------------------------- misc math
VG_DEVICE float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
VG_DEVICE float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// bf16 path: Phi(x) and x*phi(x) from ONE transcendental, at the precision the bf16 results can show (round 5).
// Phi(-|x|) = exp(-x^2/2) * Q(|x|), Q(a) = erfcx(a / sqrt(2)) / 2 as a degree-6 polynomial in a = min(|x|, 6) (a Remez
// fit weighted by what the two consumers can see: h = x Phi and GELU' = Phi + x phi stay within 0.76 of
// max(1 bf16 ulp of the exact value, 2^-17 for h / 2^-15 for GELU') over [-9, 9] BEFORE their rounding to bf16, i.e.
// |Phi error| <= ~2^-11 where Phi matters; tests/test_kernels_gpu.py::test_gelu_epilogue_within_one_bf16_ulp).  The
// round-2..4 form carried a degree-8 erfcx polynomial good to 1e-6 -- three orders below what the stores keep -- and
// cost 32 vector instructions per pair of values (+ hazard slots) in an epilogue that is VALU-bound; this one is 21:
// exp(-x^2/2) is shared with the density, the sign is put back with one v_bfi (no compare / select pair), no division.
// The fp32 parity path (vg_gemm.hip) keeps erff.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#ifndef VG_LAB_GELU_R4
#define VG_LAB_GELU_R4 0      // 1: the round-4 form (degree-8 erfcx in z = |x| / sqrt(2) on [0, 4.3], compare + select), A/B only
#endif
VG_DEVICE void gelu_parts_pk(f32x2_t x, f32x2_t& cdf, f32x2_t& pdf_x) {
#if VG_LAB_GELU_R4
  const f32x2_t ax = {fabsf(x[0]), fabsf(x[1])};
  f32x2_t z = ax * 0.70710678118654752440f;
  z = f32x2_t{fminf(z[0], 4.3f), fminf(z[1], 4.3f)};
  const f32x2_t t = z * (2.0f / 4.3f) - 1.0f;
  f32x2_t P = t * 5.537286610e-01f + 1.473028107e+00f;
  P = P * t + 1.759649855e+00f;
  P = P * t + 8.490489436e-01f;
  P = P * t + 2.434840076e-01f;
  P = P * t + -1.777284163e-01f;
  P = P * t + 1.400074079e-01f;
  P = P * t + -2.071878611e-01f;
  P = P * t + 2.402888274e-01f;
  const f32x2_t xx = x * x * -0.72134752044448170368f;
  const f32x2_t e = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
  const f32x2_t half = e * P * 0.5f;
  cdf = f32x2_t{x[0] >= 0.f ? 1.0f - half[0] : half[0], x[1] >= 0.f ? 1.0f - half[1] : half[1]};
  pdf_x = x * e * 0.39894228040143267794f;
#else
  const f32x2_t a = {fminf(fabsf(x[0]), 6.0f), fminf(fabsf(x[1]), 6.0f)};
  f32x2_t P = a * 2.8386106714606285e-04f + -4.2407736182212830e-03f;
  P = P * a + 2.6878884062170982e-02f;
  P = P * a + -9.7037091851234436e-02f;
  P = P * a + 2.3003296554088593e-01f;
  P = P * a + -3.9402547478675842e-01f;
  P = P * a + 4.9973824620246887e-01f;
  const f32x2_t xx = x * x * -0.72134752044448170368f;            // -x^2/2 * log2(e)
  const f32x2_t e = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
  const f32x2_t t = 0.5f - e * P;                                   // 1/2 - Phi(-|x|) >= 0
  cdf = f32x2_t{__builtin_copysignf(t[0], x[0]), __builtin_copysignf(t[1], x[1])} + 0.5f;
  pdf_x = x * e * 0.39894228040143267794f;
#endif
}
VG_DEVICE void gelu_parts_fast(float x, float& cdf, float& pdf_x) {
  f32x2_t c, d;
  gelu_parts_pk(f32x2_t{x, x}, c, d);
  cdf = c[0];
  pdf_x = d[0];
}
VG_DEVICE float gelu_fast(float x) {
  float cdf, px;
  gelu_parts_fast(x, cdf, px);
  return x * cdf;
}
VG_DEVICE float gelu_grad_fast(float x) {
  float cdf, px;
  gelu_parts_fast(x, cdf, px);
  return cdf + px;
}

// ---- the stored GELU derivative as one byte per element (round 6; VG_ACT_DERIV_U8 in include/vaegslm_hip.h).
// GELU'(x) = Phi(x) + x phi(x) lies in [-0.128904, 1.128904] (extrema at x = -+sqrt 2): 256 codes of step 0.005 from -0.13
// (code 26 = 0, code 226 = 1, code 255 = 1.145).  Absolute error <= 0.0025 -- what bf16 keeps for values in [0.5, 2) is
// 0.002 - 0.004 -- at half the bytes.  Encode: one fma + v_cvt_pk_u8_f32 per value (the bf16 form: half a v_cvt_pk_bf16_f32);
// decode: v_cvt_f32_ubyteN + one fma.  The forward's GELU output and the fp32 parity path are untouched.
// v_cvt_pk_u8_f32 rounds to nearest even and saturates to [0, 255] (NaN -> 0) on gfx950: tools/lab/u8_cvt_probe.hip,
// profiles/r06/labs/u8_cvt_probe.txt; the three builds below measured max |error| 0.00274 / 0.00274 / 0.00512 (mean
// 0 / 0 / +0.0025) against float64 GELU' through the C ABI.
#ifndef VG_U8_CVT
#define VG_U8_CVT 1           // how the code is rounded: 0 = v_rndne_f32 in front of the conversion (does not depend on the
#endif                        // conversion's own rounding), 1 = the conversion rounds to nearest (it does), 2 = it truncates (+0.5: lab)
constexpr float DERIV_U8_INV_STEP = 200.0f, DERIV_U8_ZERO = 26.0f, DERIV_U8_STEP = 0.005f, DERIV_U8_LO = -0.13f;
VG_DEVICE unsigned deriv_u8_put(float gp, int byte, unsigned word) {     // `byte` of `word` <- code of the derivative gp
#if VG_U8_CVT == 0
  const float t = __builtin_rintf(fmaf(gp, DERIV_U8_INV_STEP, DERIV_U8_ZERO));
#elif VG_U8_CVT == 1
  const float t = fmaf(gp, DERIV_U8_INV_STEP, DERIV_U8_ZERO);
#else
  const float t = fmaf(gp, DERIV_U8_INV_STEP, DERIV_U8_ZERO + 0.5f);
#endif
  return __builtin_amdgcn_cvt_pk_u8_f32(t, (unsigned)byte, word);          // saturates to [0, 255]
}
VG_DEVICE unsigned dpp_quad_swap1(unsigned v) {      // lane l <- lane l ^ 1 (quad_perm [1, 0, 3, 2]); every lane must be active
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}
VG_DEVICE float deriv_u8_get(unsigned word, int byte) {                  // (v_cvt_f32_ubyteN: the byte select is free)
  return fmaf((float)((word >> (8 * byte)) & 0xffu), DERIV_U8_STEP, DERIV_U8_LO);
}

VG_DEVICE float silu(float x) { return x / (1.0f + expf(-x)); }
VG_DEVICE float silu_grad(float x) {
  const float sg = 1.0f / (1.0f + expf(-x));
  return sg * (1.0f + x * (1.0f - sg));
}

// Wave-wide reductions on the DPP path: quad swaps, half-row / row mirrors and the two row broadcasts of the gfx9
// family fold into the VALU instruction itself (v_add_f32_dpp ...), six dependent steps of a few cycles each, and the
// total leaves lane 63 through v_readlane (uniform result).  The __shfl_xor butterfly these replace is six dependent
// ds_bpermute round trips through the LDS crossbar (~100 cycles each): the row kernels (RMSNorm, channel norms, CE,
// flow) run one wave per frame with one or two such reductions on each frame's critical path.
template <int CTRL, int ROW_MASK>
VG_DEVICE float dpp_move(float v, float ident) {     // lanes the row mask disables receive `ident`
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
VG_DEVICE float wave_sum(float v) {
  v += dpp_move<0xB1, 0xF>(v, 0.f);      // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xF>(v, 0.f);      // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xF>(v, 0.f);     // row_half_mirror: 8 lanes
  v += dpp_move<0x140, 0xF>(v, 0.f);     // row_mirror: 16 lanes
  v += dpp_move<0x142, 0xA>(v, 0.f);     // row_bcast:15 into rows 1, 3
  v += dpp_move<0x143, 0xC>(v, 0.f);     // row_bcast:31 into rows 2, 3: row 3 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
VG_DEVICE float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1, 0xF>(v, -INFINITY));
  v = fmaxf(v, dpp_move<0x4E, 0xF>(v, -INFINITY));
  v = fmaxf(v, dpp_move<0x141, 0xF>(v, -INFINITY));
  v = fmaxf(v, dpp_move<0x140, 0xF>(v, -INFINITY));
  v = fmaxf(v, dpp_move<0x142, 0xA>(v, -INFINITY));
  v = fmaxf(v, dpp_move<0x143, 0xC>(v, -INFINITY));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// row predicate: frame (b = m / T, t = m % T) is valid iff t < lengths[b]
VG_DEVICE bool row_valid(const int* __restrict__ lengths, int T, int m) {
  if (lengths == nullptr) return true;
  const int b = m / T;
  return (m - b * T) < lengths[b];
}

}  // namespace vg

// ---------------------------------------------------------------- host-side error plumbing
namespace vg_host {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// optional HIP-event timing (vg_prof.hip); kinds are the VG_PROF_* enum of the public header
int prof_begin(int kind, double work, hipStream_t stream, double bytes = 0.0);
void prof_end(int token, hipStream_t stream);
// Inside an open prof_begin / prof_end bracket of this thread: a fresh (start, stop) event pair for ONE kernel dispatch
// (VG_LAUNCH hands it to hipExtLaunchKernelGGL, which stamps the events with the dispatch's own start and end -- the
// kernel's execution time as rocprofv3 reports it, without the gap to the previous dispatch that a recorded event pair
// around the launch includes).  false: no bracket is open, or VG_PROF_EXT is not set (the default: see vg_prof.hip): launch plainly.
bool prof_kernel_events(hipEvent_t* start, hipEvent_t* stop);
// out[n] += sum_m x[m][n] in one launch without workspace (vg_rows.hip; slow path of fused bias gradients)
void colsum_accumulate(const void* x, int M, int N, long ld, float* out, int dtype, hipStream_t stream);
}  // namespace vg_host

#define VG_LAUNCH(kernel, grid, block, lds, stream, ...)                                                   \
  do {                                                                                                     \
    hipEvent_t vg_ev0_, vg_ev1_;                                                                           \
    if (vg_host::prof_kernel_events(&vg_ev0_, &vg_ev1_))                                                   \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, vg_ev0_, vg_ev1_, 0, __VA_ARGS__);           \
    else                                                                                                   \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                   \
  } while (0)

#define VG_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      vg_host::set_error(__VA_ARGS__);        \
      return 1;                               \
    }                                         \
  } while (0)
