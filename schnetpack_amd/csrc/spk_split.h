// Split-precision matrix path (round 6): fp32-quality products on the f16 matrix instructions of gfx950.
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 cycles per 32 x 32 x 2 product); v_mfma_f32_32x32x16_f16 runs 16 x that
// (32 cycles per 32 x 32 x 16).  Every fp32 operand is written as   x = x_h + 2^-11 x_l   with x_h = fp16(x) and
// x_l = fp16((x - x_h) 2^11), both round-to-nearest: the residual of the first rounding is exact in fp32 and at most half an fp16
// ulp, so the second rounding leaves |x - x_h - 2^-11 x_l| <= 2^-24 |x| -- the pair carries the fp32 significand.  A product of two
// such operands keeps the three leading terms
//        a b  ~  a_h b_h  +  2^-11 (a_h b_l + a_l b_h)            (dropped: 2^-22 a_l b_l <= 2^-24 |a b|)
// accumulated in fp32 by the matrix core: the hh products into one accumulator, the two cross products into a second one that
// is folded in with weight 2^-11 at the end (SP_FOLD) -- three f16 instructions per 16 k instead of eight f32 ones per 16 k, 3/16 of
// the time.  Measured on the device (scripts/split_mfma_probe.hip, profiles/r06_split_mfma.md): K = 128 products of filter-network
// operands 2.6e-7 of max |C| against a float64 product (v_mfma_f32_32x32x2_f32: 6.2e-7 -- its 128-term fp32 chain rounds more often),
// 578 logical TFLOP/s against 154.
// Range: |x| must stay below 65504 (fp16); operands here are radial-basis values, activations, filter outputs and weights.  Low
// parts are scaled by 2^11 so that they are normal fp16 numbers whenever the high part is; where an operand is exactly
// representable (0/1 incidence matrices) the weight 2^-11 goes into IT instead of a second accumulator.
#pragma once
#include "spk_common.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

#define SP_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)
#define SP_UP 2048.0f
#define SP_DOWN (1.0f / 2048.0f)

// (x is pinned as ONE rounded fp32 value first.  Under the default -ffp-contract=fast the compiler otherwise fuses a product that
//  feeds the split into the conversions -- v_fma_mixlo_f16 rounds the EXACT product a*b to fp16 for the residual while the operand
//  image gets fp16(fp32(a*b)): the two high parts differ by one fp16 ulp wherever the double rounding crosses a tie, and the pair no
//  longer adds up -- 2^-11 relative on that element.  Found on the device: 3.7e-5 outliers in y of single atoms, everything else 2e-7.)
__device__ __forceinline__ void sp_split(float x, _Float16& h, _Float16& l) {
  asm("" : "+v"(x));
  h = (_Float16)x;
  l = (_Float16)((x - (float)h) * SP_UP);
}
// eight consecutive values -> the (high, low) operand pair of one k-step
__device__ __forceinline__ void sp_split8(const float (&x)[8], h16x8& h, h16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) { _Float16 a, b; sp_split(x[e], a, b); h[e] = a; l[e] = b; }
}
__device__ __forceinline__ void sp_split4(const f32x4 x, h16x4& h, h16x4& l) {
  _Float16 a, b;
  sp_split(x.x, a, b); h[0] = a; l[0] = b;
  sp_split(x.y, a, b); h[1] = a; l[1] = b;
  sp_split(x.z, a, b); h[2] = a; l[2] = b;
  sp_split(x.w, a, b); h[3] = a; l[3] = b;
}
__device__ __forceinline__ h16x8 sp_cat(const h16x4 a, const h16x4 b) { return h16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// one k-step of a split product: main += A_h B_h, cross += A_h B_l + A_l B_h
#define SP_STEP(AH, AL, BH, BL, MAIN, CROSS) \
  do { (MAIN) = SP_MFMA((AH), (BH), (MAIN)); (CROSS) = SP_MFMA((AH), (BL), (CROSS)); (CROSS) = SP_MFMA((AL), (BH), (CROSS)); } while (0)
// main += 2^-11 cross
#define SP_FOLD(MAIN, CROSS) \
  do { _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) (MAIN)[r_] = fmaf((CROSS)[r_], SP_DOWN, (MAIN)[r_]); } while (0)

// Accumulator order of the contraction index: register r of lane half `hi` of a 32 x 32 accumulator holds row (r & 3) + 8 (r >> 2) + 4 hi.
// When the 16 values a lane holds become the B (or A) operand of the NEXT product, k-step s' in {0, 1} takes its registers 8 s' .. 8 s' + 7
// as they lie; the other operand must then be laid out so that slot (s', hi, e) of a 32-wide k block is index
//        sp_acc_k(s', hi, e) = (e & 3) + 8 (2 s' + (e >> 2)) + 4 hi
// -- two runs of four consecutive indices, 8 apart.
__host__ __device__ __forceinline__ int sp_acc_k(int sp, int hi, int e) { return (e & 3) + 8 * (2 * sp + (e >> 2)) + 4 * hi; }
