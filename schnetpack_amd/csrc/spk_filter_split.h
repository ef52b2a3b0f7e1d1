// Split-precision operand images of the SchNet filter network (spk_split.h), shared by the molecule-resident kernels
// (spk_schnet_mol.hip) and the pair kernels of the general driver (spk_cfconv.hip).  n_filters = 128.
#pragma once
#include "spk_common.h"
#include "spk_split.h"

// radial basis from parameters staged in LDS (nn/radial.py:11-15 gaussian, :105-110 bessel; hardware transcendentals as
// spk_rbf_eval_fast)
__device__ __forceinline__ void ml_rbf(int kind, int n_rbf, const float* __restrict__ p0, const float* __restrict__ p1, int k, float d,
                                       float& phi, float& dphi) {
  if (k >= n_rbf) { phi = 0.f; dphi = 0.f; return; }
  if (kind == SPK_RBF_GAUSSIAN) {
    const float w = p1[k];
    const float c = -0.5f * __builtin_amdgcn_rcpf(w * w);
    const float t = d - p0[k];
    phi = __builtin_amdgcn_exp2f(1.4426950408889634f * c * t * t);
    dphi = 2.0f * c * t * phi;
  } else {
    const float om = p0[k];
    const float rev = om * d * 0.15915494309189535f;
    const float s = __builtin_amdgcn_sinf(rev), co = __builtin_amdgcn_cosf(rev);
    if (d == 0.0f) { phi = s; dphi = 0.f; }
    else { const float inv = __builtin_amdgcn_rcpf(d); phi = s * inv; dphi = (om * co - phi) * inv; }
  }
}


// ------------------------------------------------------------------------------------------ split-precision images (spk_split.h)
// Filter-network weights as (high, low) fp16 operand images in LDS, made from the raw fp32 tensors while they are staged.
// W2 [NF][NF]: slot (channel tile t, k-step s, lane) = eight values W2[32 t + el][32 (s >> 1) + sp_acc_k(s & 1, hi, e)] -- the contraction
// index in ACCUMULATOR order, so that the hidden activations go from the accumulator registers of GEMM 1 into GEMM 2 as they lie
// (backward) or are written to LDS as two 16-byte pieces per image (forward).  The same image is the B operand of the forward
// (columns = channels) and the A operand of the backward (rows = channels).
template <int NTHREADS>
__device__ __forceinline__ void ml_stage_w2_split(h16x8* __restrict__ dh, h16x8* __restrict__ dl, const float* __restrict__ w2, int tid) {
  // 2 048 slots; batches of four per thread while they last (any workgroup size: the row-tile forward runs 1 024 threads)
#pragma unroll 1
  for (int p0 = 0; tid + p0 * NTHREADS < 2048; p0 += 4) {
    f32x4 va[4], vb[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int sl = tid + (p0 + p) * NTHREADS;
      if (sl < 2048) {
        const int lane = sl & 63, s = (sl >> 6) & 7, t = sl >> 9;
        const float* src = w2 + (32 * t + (lane & 31)) * 128 + 32 * (s >> 1) + 16 * (s & 1) + 4 * (lane >> 5);
        va[p] = *(const f32x4*)src;
        vb[p] = *(const f32x4*)(src + 8);
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int sl = tid + (p0 + p) * NTHREADS;
      if (sl < 2048) {
        h16x4 ah, al, bh, bl;
        sp_split4(va[p], ah, al);
        sp_split4(vb[p], bh, bl);
        dh[sl] = sp_cat(ah, bh);
        dl[sl] = sp_cat(al, bl);
      }
    }
  }
}
// W1 [NF][n_rbf]: slot (hidden tile t, lane): k-step 0 = basis functions 8 hi + e; k-step 1 (n_rbf > 16) = 16 + 4 hi + e for e < 4 and,
// for n_rbf > 24 only, 24 + 4 hi + (e - 4) for e >= 4 -- the live functions of a 20-wide basis are spread evenly over the two lane
// halves (12 evaluations per lane).  Image = [256 slots] h16x8 (k-step 0), then [256 slots] h16x4 (KPB = 3) or h16x8 (KPB = 4).
template <int KPB>
struct MlW1Image {
  static constexpr int BYTES = 4096 + (KPB > 2 ? (KPB == 4 ? 4096 : 2048) : 0);     // per image (high or low)
};
__device__ __forceinline__ int ml_w1_k(int s, int hi, int e) { return s == 0 ? 8 * hi + e : (e < 4 ? 16 + 4 * hi + e : 20 + 4 * hi + e); }
template <int KPB>
__device__ __forceinline__ void ml_stage_w1_split(char* __restrict__ ih, char* __restrict__ il, const float* __restrict__ w1, int n_rbf, int slot) {
  if (slot >= 256) return;
  const int lane = slot & 63, t = slot >> 6, hi = lane >> 5;
  const float* src = w1 + (32 * t + (lane & 31)) * n_rbf;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const int k = ml_w1_k(0, hi, e); x[e] = k < n_rbf ? src[k] : 0.f; }
  h16x8 h, l;
  sp_split8(x, h, l);
  ((h16x8*)ih)[slot] = h; ((h16x8*)il)[slot] = l;
  if (KPB > 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int k = ml_w1_k(1, hi, e); x[e] = (k < n_rbf && (KPB == 4 || e < 4)) ? src[k] : 0.f; }
    sp_split8(x, h, l);
    if (KPB == 4) { ((h16x8*)(ih + 4096))[slot] = h; ((h16x8*)(il + 4096))[slot] = l; }
    else {
      ((h16x4*)(ih + 4096))[slot] = h16x4{h[0], h[1], h[2], h[3]};
      ((h16x4*)(il + 4096))[slot] = h16x4{l[0], l[1], l[2], l[3]};
    }
  }
}
template <int KPB>
__device__ __forceinline__ void ml_w1_operand(const char* __restrict__ ih, const char* __restrict__ il, int s, int slot, h16x8& h, h16x8& l) {
  if (s == 0) { h = ((const h16x8*)ih)[slot]; l = ((const h16x8*)il)[slot]; }
  else if (KPB == 4) { h = ((const h16x8*)(ih + 4096))[slot]; l = ((const h16x8*)(il + 4096))[slot]; }
  else {
    const h16x4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    h = sp_cat(((const h16x4*)(ih + 4096))[slot], z);
    l = sp_cat(((const h16x4*)(il + 4096))[slot], z);
  }
}
// the radial basis of one pair (and its d-derivative) as split B operands: this lane's slots of the k-steps, ml_w1_k() order
template <int KPB, bool DERIV>
__device__ __forceinline__ void ml_basis_split(int kind, int n_rbf, const float* __restrict__ p0, const float* __restrict__ p1, int hi, float d,
                                               h16x8 (&ph)[2], h16x8 (&pl)[2], h16x8 (&dh)[2], h16x8 (&dl)[2]) {
#pragma unroll
  for (int s = 0; s < (KPB > 2 ? 2 : 1); ++s) {
    float v[8], dv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = 0.f; dv[e] = 0.f;
      if (s == 0 || KPB == 4 || e < 4) ml_rbf(kind, n_rbf, p0, p1, ml_w1_k(s, hi, e), d, v[e], dv[e]);
    }
    sp_split8(v, ph[s], pl[s]);
    if (DERIV) sp_split8(dv, dh[s], dl[s]);
  }
}

