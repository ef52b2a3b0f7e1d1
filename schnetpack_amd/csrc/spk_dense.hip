// Dense layers (nn/base.py:52-55) on the fp32 matrix cores of gfx950.
//
// "T-GEMM" convention used by every MFMA kernel of this library (v_mfma_f32_32x32x2_f32,
// exact fp32, 64 lanes):
//   the wave computes a 32x32 tile of OUT^T:  rows = output features i, columns = samples m.
//   A operand (weights):  lane l holds A[i = i0 + (l & 31)][kk],
//   B operand (samples):  lane l holds B[kk][m = m0 + (l & 31)] = in[m][kk],
//   where for k-step s = 4u + v the lane half hi = l >> 5 supplies kk = 8u + 4hi + v
//   (any bijection between (step, half) and kk is legal as long as A and B agree; this one lets
//   each lane fetch 4 consecutive kk with one 16-byte load),
//   C/D accumulator:      acc[r] <-> row i = i0 + (r & 3) + 8 (r >> 2) + 4 hi, column m0 + (l & 31).
//   Because the accumulator rows use the same (r>>2, r&3, hi) -> index map as the k-steps, an
//   accumulator tile can be fed straight back as the B operand of the next layer (chained GEMMs
//   without leaving registers) -- used by the fused cfconv kernels.
#include "spk_common.h"
#include "spk_gemm_tn.h"

// One chunk = 8 k-blocks (64 contraction indices): 8 A + 8 B 16-byte operands per lane.  All loads
// of a chunk are issued before its 32 MFMAs, and the next chunk is requested before the current one
// is consumed (two register sets), so a wave pays the memory latency once instead of per k-block.
#define DCH 8
template <bool TRANS, int PRO, int CH>
__device__ __forceinline__ void dense_load_chunk(f32x4 (&av)[CH], f32x4 (&bv)[CH], int c, int nug,
                                                 const float* __restrict__ inrow,
                                                 const float* __restrict__ prow,
                                                 const float* __restrict__ w, int KC, int NW, int t,
                                                 int el, int hi) {
  // KC and NW are multiples of 4 (not necessarily of 8 / 32): contraction groups past KC and weight rows past NW read as 0
  const bool rin = 32 * t + el < NW;
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int ug = c * CH + u;
    if (ug < nug) {
      const int kk0 = 8 * ug + 4 * hi;
      const bool kin = kk0 < KC;
      f32x4 b{0.f, 0.f, 0.f, 0.f};
      if (kin) b = *(const f32x4*)(inrow + kk0);
      if (PRO != SPK_ACT_NONE && kin) {
        const f32x4 pv = *(const f32x4*)(prow + kk0);
        b.x *= spk_act_grad<PRO>(pv.x); b.y *= spk_act_grad<PRO>(pv.y);
        b.z *= spk_act_grad<PRO>(pv.z); b.w *= spk_act_grad<PRO>(pv.w);
      }
      bv[u] = b;
      f32x4 a4{0.f, 0.f, 0.f, 0.f};
      if (kin && rin) {
        if (!TRANS) {
          a4 = *(const f32x4*)(w + (int64_t)(32 * t + el) * KC + kk0);
        } else {
          const float* wp = w + (int64_t)kk0 * NW + 32 * t + el;
          a4.x = wp[0]; a4.y = wp[NW]; a4.z = wp[2 * (int64_t)NW]; a4.w = wp[3 * (int64_t)NW];
        }
      }
      av[u] = a4;
    }
  }
}

template <int CH>
__device__ __forceinline__ f32x16 dense_mfma_chunk(const f32x4 (&av)[CH], const f32x4 (&bv)[CH], int c,
                                                   int nug, f32x16 acc) {
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    if (c * CH + u < nug) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
    }
  }
  return acc;
}

// one wave walks the 32 x 32 output tiles first, first + stride, ...
template <int ACT, bool TRANS, int PRO>
__device__ __forceinline__ void dense_tiles(
    const float* __restrict__ in, const float* __restrict__ pre_in, const float* __restrict__ w,
    const float* __restrict__ b, const float* res, float* out,
    float* __restrict__ pre_out, int64_t M, int KC, int NW, int64_t ntasks, int64_t first, int64_t stride) {
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 5, el = lane & 31;
  // chunk of k-blocks per register set: the transposed form gathers its A operand with four strided scalar loads per k-block --
  // eight blocks per set kept 32 loads and their addresses in flight twice over and pushed the kernels into scratch (k_gemm_pair)
  // or to one wave per SIMD (k_dense_mfma); four blocks per set fit
  constexpr int CH = TRANS ? 4 : DCH;
  const int tcount = (NW + 31) / 32;
  const int nug = (KC + 7) / 8;
  const int nch = (nug + CH - 1) / CH;
  for (int64_t task = first; task < ntasks; task += stride) {
    const int64_t mt = task / tcount;
    const int t = (int)(task % tcount);
    const int64_t m = mt * 32 + el;
    const bool valid = m < M;
    const int64_t mc = valid ? m : (M - 1);
    const float* inrow = in + mc * KC;
    const float* prow = PRO != SPK_ACT_NONE ? pre_in + mc * KC : nullptr;
    f32x4 a0[CH], b0[CH], a1[CH], b1[CH];
    dense_load_chunk<TRANS, PRO>(a0, b0, 0, nug, inrow, prow, w, KC, NW, t, el, hi);
    f32x4 rv[4];  // residual rows, requested ahead of the MFMAs
#pragma unroll
    for (int q = 0; q < 4; ++q)
      rv[q] = (res && valid && 32 * t + 8 * q + 4 * hi < NW) ? *(const f32x4*)(res + m * NW + 32 * t + 8 * q + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
      acc[r] = (b && col < NW) ? b[col] : 0.f;
    }
    for (int c = 0; c < nch; c += 2) {
      if (c + 1 < nch) dense_load_chunk<TRANS, PRO>(a1, b1, c + 1, nug, inrow, prow, w, KC, NW, t, el, hi);
      acc = dense_mfma_chunk(a0, b0, c, nug, acc);
      if (c + 2 < nch) dense_load_chunk<TRANS, PRO>(a0, b0, c + 2, nug, inrow, prow, w, KC, NW, t, el, hi);
      if (c + 1 < nch) acc = dense_mfma_chunk(a1, b1, c + 1, nug, acc);
    }
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (32 * t + 8 * q + 4 * hi >= NW) continue;
        const int64_t off = m * NW + 32 * t + 8 * q + 4 * hi;
        f32x4 o;
        o.x = acc[4 * q]; o.y = acc[4 * q + 1]; o.z = acc[4 * q + 2]; o.w = acc[4 * q + 3];
        if (pre_out) *(f32x4*)(pre_out + off) = o;
        o.x = spk_act<ACT>(o.x); o.y = spk_act<ACT>(o.y); o.z = spk_act<ACT>(o.z); o.w = spk_act<ACT>(o.w);
        if (res) o += rv[q];
        *(f32x4*)(out + off) = o;
      }
    }
  }
}

template <int ACT, bool TRANS, int PRO>
__global__ __launch_bounds__(256) void k_dense_mfma(
    const float* __restrict__ in, const float* __restrict__ pre_in, const float* __restrict__ w,
    const float* __restrict__ b, const float* res, float* out,
    float* __restrict__ pre_out, int64_t M, int KC, int NW, int64_t ntasks) {
  dense_tiles<ACT, TRANS, PRO>(in, pre_in, w, b, res, out, pre_out, M, KC, NW, ntasks, blockIdx.x * 4 + (threadIdx.x >> 6), (int64_t)gridDim.x * 4);
}

// Two independent GEMMs of a Dense backward in ONE launch: out = a w (TRANS) or a w^T, and (G, gb) = (U^T X, column sums of U).
// The first nblk_dense workgroups (8 waves = 8 tile walkers) take the Dense tiles, the rest the (tile, slice) blocks of gemm_tn.
template <bool TRANS>
__global__ __launch_bounds__(64 * TN_WAVES) void k_gemm_pair(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out, int64_t M,
                                                             int KC, int NW, int64_t ntasks, int nblk_dense, GemmTnArgs tn) {
  if ((int)blockIdx.x < nblk_dense) {
    dense_tiles<SPK_ACT_NONE, TRANS, SPK_ACT_NONE>(in, nullptr, w, nullptr, nullptr, out, nullptr, M, KC, NW, ntasks,
                                                   (int64_t)blockIdx.x * TN_WAVES + (threadIdx.x >> 6), (int64_t)nblk_dense * TN_WAVES);
  } else {
    const int r = (int)blockIdx.x - nblk_dense;
    gemm_tn_block(tn, r % tn.n_tiles, r / tn.n_tiles);
  }
}

// Long contractions with few output tiles (the input gradient of PaiNN's filter layer: [E, 1152] x [1152, 20] -- one column
// tile, 18 chunks): the four waves of a workgroup share ONE 32 x 32 tile, wave w takes chunks w, w + 4, ..., the partial tiles
// meet in LDS in wave order (deterministic).  Linear layers only (bias, no activation / prologue / residual).
template <bool TRANS>
__global__ __launch_bounds__(256) void k_dense_mfma_splitk(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                                                           float* __restrict__ out, int64_t M, int KC, int NW) {
  __shared__ float red[3][32][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  const int tcount = (NW + 31) / 32;
  const int nug = (KC + 7) / 8;
  const int nch = (nug + DCH - 1) / DCH;
  const int64_t task = blockIdx.x;
  const int64_t mt = task / tcount;
  const int t = (int)(task % tcount);
  const int64_t m = mt * 32 + el;
  const bool valid = m < M;
  const float* inrow = in + (valid ? m : (M - 1)) * KC;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 a0[DCH], b0[DCH], a1[DCH], b1[DCH];
  if (wv < nch) dense_load_chunk<TRANS, SPK_ACT_NONE>(a0, b0, wv, nug, inrow, nullptr, w, KC, NW, t, el, hi);
  for (int c = wv; c < nch; c += 8) {
    if (c + 4 < nch) dense_load_chunk<TRANS, SPK_ACT_NONE>(a1, b1, c + 4, nug, inrow, nullptr, w, KC, NW, t, el, hi);
    acc = dense_mfma_chunk(a0, b0, c, nug, acc);
    if (c + 8 < nch) dense_load_chunk<TRANS, SPK_ACT_NONE>(a0, b0, c + 8, nug, inrow, nullptr, w, KC, NW, t, el, hi);
    if (c + 4 < nch) acc = dense_mfma_chunk(a1, b1, c + 4, nug, acc);
  }
  if (wv > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv - 1][(r & 3) + 8 * (r >> 2) + 4 * hi][el] = acc[r];
  }
  __syncthreads();
  if (wv == 0 && valid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 32 * t + 8 * q + 4 * hi;
      if (col >= NW) continue;
      f32x4 o;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 4 * q + v, rr = (r & 3) + 8 * (r >> 2) + 4 * hi;
        o[v] = ((acc[r] + red[0][rr][el]) + red[1][rr][el]) + red[2][rr][el] + (b ? b[col + v] : 0.f);
      }
      *(f32x4*)(out + m * NW + col) = o;
    }
  }
}

// Few output tiles (small batches: a training step has 168 - 336 rows per Dense layer): one tile per WORKGROUP, the four waves split the
// contraction chunk-wise (K = 128 -> one chunk of four k-blocks each: ONE memory round trip per wave instead of a chain of them), the partial
// tiles meet in LDS in wave order (deterministic), waves 0 .. 3 share the epilogue of dense_tiles (bias, pre-activation store, activation, residual).
// Same operand conventions and template switches as k_dense_mfma.
template <int ACT, bool TRANS, int PRO, int CH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_dense_mfma_sk(const float* __restrict__ in, const float* __restrict__ pre_in, const float* __restrict__ w,
                                                       const float* __restrict__ b, const float* res, float* out, float* __restrict__ pre_out, int64_t M, int KC,
                                                       int NW) {
  __shared__ float red[WAVES][32][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  const int tcount = (NW + 31) / 32;
  const int nug = (KC + 7) / 8;
  const int nch = (nug + CH - 1) / CH;
  const int64_t task = blockIdx.x;
  const int64_t mt = task / tcount;
  const int t = (int)(task % tcount);
  const int64_t m = mt * 32 + el;
  const bool valid = m < M;
  const int64_t mc = valid ? m : (M - 1);
  const float* inrow = in + mc * KC;
  const float* prow = PRO != SPK_ACT_NONE ? pre_in + mc * KC : nullptr;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 a0[CH], b0[CH], a1[CH], b1[CH];
  if (wv < nch) dense_load_chunk<TRANS, PRO>(a0, b0, wv, nug, inrow, prow, w, KC, NW, t, el, hi);
  // the epilogue is shared by waves 0 .. 3: wave q finishes the column group q (8 of the 32 output features); its operands are requested ahead
  const int col = 32 * t + 8 * wv + 4 * hi;
  const bool fin = valid && wv < 4 && col < NW;
  const int64_t off = m * NW + col;
  const f32x4 z4{0.f, 0.f, 0.f, 0.f};
  const f32x4 rv = (res && fin) ? *(const f32x4*)(res + off) : z4;
  f32x4 bq = z4;      // (scalar loads: a bias may be any 4-byte aligned view of a flat parameter buffer)
  if (b && fin) { bq.x = b[col]; bq.y = b[col + 1]; bq.z = b[col + 2]; bq.w = b[col + 3]; }
  for (int c = wv; c < nch; c += 2 * WAVES) {
    if (c + WAVES < nch) dense_load_chunk<TRANS, PRO>(a1, b1, c + WAVES, nug, inrow, prow, w, KC, NW, t, el, hi);
    acc = dense_mfma_chunk(a0, b0, c, nug, acc);
    if (c + 2 * WAVES < nch) dense_load_chunk<TRANS, PRO>(a0, b0, c + 2 * WAVES, nug, inrow, prow, w, KC, NW, t, el, hi);
    if (c + WAVES < nch) acc = dense_mfma_chunk(a1, b1, c + WAVES, nug, acc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wv][(r & 3) + 8 * (r >> 2) + 4 * hi][el] = acc[r];
  __syncthreads();
  if (!fin) return;
  f32x4 o;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int rr = 8 * wv + 4 * hi + v;
    float sum = red[0][rr][el];
#pragma unroll
    for (int q2 = 1; q2 < WAVES; ++q2) sum += red[q2][rr][el];      // wave order: deterministic
    o[v] = sum + bq[v];
  }
  if (pre_out) *(f32x4*)(pre_out + off) = o;
  o.x = spk_act<ACT>(o.x); o.y = spk_act<ACT>(o.y); o.z = spk_act<ACT>(o.z); o.w = spk_act<ACT>(o.w);
  o += rv;
  *(f32x4*)(out + off) = o;
}

template <int ACT, bool TRANS, int PRO>
static void dense_sk_launch(int KC, unsigned ntasks, hipStream_t stream, const float* in, const float* pre_in, const float* w, const float* b, const float* res, float* out,
                            float* pre_out, int64_t M, int NW) {
  // chunk = 2 / 4 k-blocks over four waves up to K = 128 (every wave gets one chunk: one memory round trip), eight waves with chunks of 4 beyond
  const int nug = (KC + 7) / 8;
#define SPK_SK(CH, WV) hipLaunchKernelGGL((k_dense_mfma_sk<ACT, TRANS, PRO, CH, WV>), dim3(ntasks), dim3(64 * WV), 0, stream, in, pre_in, w, b, res, out, pre_out, M, KC, NW)
  if (nug <= 8) SPK_SK(2, 4);
  else if (nug <= 16) SPK_SK(4, 4);
  else SPK_SK(4, 8);
#undef SPK_SK
}

// Straightforward kernel for any shape: one thread per output element.
__global__ void k_dense_simple(const float* __restrict__ in, const float* __restrict__ pre_in,
                               const float* __restrict__ w, const float* __restrict__ b,
                               const float* res, float* out,
                               float* __restrict__ pre_out, int64_t M, int KC, int NW, int act,
                               int trans, int pro) {
  const int64_t total = M * (int64_t)NW;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / NW;
    const int i = (int)(t % NW);
    float acc = b ? b[i] : 0.f;
    for (int kk = 0; kk < KC; ++kk) {
      float v = in[m * KC + kk];
      if (pro == SPK_ACT_SSP) v *= spk_act_grad<SPK_ACT_SSP>(pre_in[m * KC + kk]);
      else if (pro == SPK_ACT_SILU) v *= spk_act_grad<SPK_ACT_SILU>(pre_in[m * KC + kk]);
      float a = trans ? w[(int64_t)kk * NW + i] : w[(int64_t)i * KC + kk];
      acc = fmaf(a, v, acc);
    }
    if (pre_out) pre_out[t] = acc;
    if (act == SPK_ACT_SSP) acc = spk_ssp(acc);
    else if (act == SPK_ACT_SILU) acc = acc * spk_sigmoid(acc);
    if (res) acc += res[t];
    out[t] = acc;
  }
}

static bool aligned16(const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; }

// in [M,KC] -> out [M,NW]
static int dense_dispatch(const float* in, const float* pre_in, const float* w, const float* b,
                          const float* res, float* out, float* pre_out, int64_t M, int KC, int NW,
                          int act, bool trans, int pro, hipStream_t stream, const char* who) {
  SPK_CHECK_ARG(M >= 0 && KC > 0 && NW > 0, "%s: bad sizes M=%lld K=%d N=%d", who, (long long)M, KC, NW);
  if (M == 0) return SPK_OK;
  SPK_CHECK_ARG(in && w && out, "%s: null pointer", who);
  SPK_CHECK_ARG(act >= 0 && act <= 2 && pro >= 0 && pro <= 2, "%s: unknown activation", who);
  SPK_CHECK_ARG(pro == SPK_ACT_NONE || pre_in != nullptr, "%s: pre-activation required", who);
  const int variant = spk_get_variant();
  const bool shape_ok = (KC % 4 == 0) && (NW % 4 == 0) && aligned16(in) && aligned16(pre_in) &&
                        aligned16(w) && aligned16(res) && aligned16(out) && aligned16(pre_out);
  SPK_CHECK_ARG(variant != SPK_VARIANT_MFMA || shape_ok, "%s: shape K=%d N=%d not supported by the MFMA kernel", who, KC, NW);
  SpkProfScope prof(trans ? "dense_bwd" : "dense_fwd", stream);
  // MFMA instantiations: forward layers (activation epilogue, any weight layout) and input-gradient layers
  // (k-major weights with an act' prologue); a layer that wants both goes to the simple kernel
  const bool mfma_combo = (pro == SPK_ACT_NONE) || (act == SPK_ACT_NONE && trans);
  if (shape_ok && mfma_combo && variant != SPK_VARIANT_SIMPLE) {
    const int64_t ntasks = ((M + 31) / 32) * ((NW + 31) / 32);
    if (ntasks <= 4 * (int64_t)spk_num_cus() && KC >= 16 && !getenv("SPK_DENSE_NO_SK")) {
      // few tiles: one tile per workgroup, the contraction split over its waves
      const unsigned nt = (unsigned)ntasks;
#define SPK_SKL(A, T, P) dense_sk_launch<A, T, P>(KC, nt, stream, in, pre_in, w, b, res, out, pre_out, M, NW)
      if (pro == SPK_ACT_NONE) {
        if (!trans) {
          if (act == SPK_ACT_NONE) SPK_SKL(SPK_ACT_NONE, false, SPK_ACT_NONE);
          else if (act == SPK_ACT_SSP) SPK_SKL(SPK_ACT_SSP, false, SPK_ACT_NONE);
          else SPK_SKL(SPK_ACT_SILU, false, SPK_ACT_NONE);
        } else {
          if (act == SPK_ACT_NONE) SPK_SKL(SPK_ACT_NONE, true, SPK_ACT_NONE);
          else if (act == SPK_ACT_SSP) SPK_SKL(SPK_ACT_SSP, true, SPK_ACT_NONE);
          else SPK_SKL(SPK_ACT_SILU, true, SPK_ACT_NONE);
        }
      } else {
        if (pro == SPK_ACT_SSP) SPK_SKL(SPK_ACT_NONE, true, SPK_ACT_SSP);
        else SPK_SKL(SPK_ACT_NONE, true, SPK_ACT_SILU);
      }
#undef SPK_SKL
      SPK_LAUNCH_CHECK();
      return SPK_OK;
    }
    if (getenv("SPK_DENSE_NO_SK") && KC >= 256 && ntasks <= 2 * (int64_t)spk_num_cus() && act == SPK_ACT_NONE && pro == SPK_ACT_NONE && !res && !pre_out) {
      if (trans) hipLaunchKernelGGL((k_dense_mfma_splitk<true>), dim3((unsigned)ntasks), dim3(256), 0, stream, in, w, b, out, M, KC, NW);
      else hipLaunchKernelGGL((k_dense_mfma_splitk<false>), dim3((unsigned)ntasks), dim3(256), 0, stream, in, w, b, out, M, KC, NW);
      SPK_LAUNCH_CHECK();
      return SPK_OK;
    }
    const int grid = spk_grid_for(ntasks, 4, spk_num_cus() * 8);
#define SPK_DENSE_LAUNCH(A, T, P)                                                               \
  hipLaunchKernelGGL((k_dense_mfma<A, T, P>), dim3(grid), dim3(256), 0, stream, in, pre_in, w, b, \
                     res, out, pre_out, M, KC, NW, ntasks)
    if (pro == SPK_ACT_NONE) {
      if (!trans) {
        if (act == SPK_ACT_NONE) SPK_DENSE_LAUNCH(SPK_ACT_NONE, false, SPK_ACT_NONE);
        else if (act == SPK_ACT_SSP) SPK_DENSE_LAUNCH(SPK_ACT_SSP, false, SPK_ACT_NONE);
        else SPK_DENSE_LAUNCH(SPK_ACT_SILU, false, SPK_ACT_NONE);
      } else {   // forward layer reading a transposed (k-major) copy of its weight
        if (act == SPK_ACT_NONE) SPK_DENSE_LAUNCH(SPK_ACT_NONE, true, SPK_ACT_NONE);
        else if (act == SPK_ACT_SSP) SPK_DENSE_LAUNCH(SPK_ACT_SSP, true, SPK_ACT_NONE);
        else SPK_DENSE_LAUNCH(SPK_ACT_SILU, true, SPK_ACT_NONE);
      }
    } else {
      if (pro == SPK_ACT_SSP) SPK_DENSE_LAUNCH(SPK_ACT_NONE, true, SPK_ACT_SSP);
      else SPK_DENSE_LAUNCH(SPK_ACT_NONE, true, SPK_ACT_SILU);
    }
#undef SPK_DENSE_LAUNCH
  } else {
    const int grid = spk_grid_for(M * NW, 256, spk_num_cus() * 16);
    hipLaunchKernelGGL(k_dense_simple, dim3(grid), dim3(256), 0, stream, in, pre_in, w, b, res, out,
                       pre_out, M, KC, NW, act, trans ? 1 : 0, pro);
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_dense_f32(const float* x, const float* w, const float* b, const float* res,
                             float* y, float* pre, int64_t m, int32_t k, int32_t n_out,
                             int32_t act, void* stream) {
  return dense_dispatch(x, nullptr, w, b, res, y, pre, m, k, n_out, act, false, SPK_ACT_NONE,
                        (hipStream_t)stream, "spk_dense_f32");
}

extern "C" int spk_dense_bwd_input_f32(const float* dy, const float* pre, const float* w,
                                       const float* res, float* dx, int64_t m, int32_t k,
                                       int32_t n_out, int32_t act, void* stream) {
  // contraction over the n_out outputs; result width k
  return dense_dispatch(dy, pre, w, nullptr, res, dx, nullptr, m, n_out, k, SPK_ACT_NONE, true, act,
                        (hipStream_t)stream, "spk_dense_bwd_input_f32");
}

extern "C" int spk_gemm_tn_plan(int64_t n, int32_t O, int32_t K, int32_t* n_slices, int64_t* ws_floats, int32_t* n_tiles);
// out [m, n_out] = a w^T (trans = 0: a [m, k], w [n_out, k]) or out [m, k] = a w (trans = 1: a [m, n_out], w [n_out, k]),
// and G [O, K] = U^T X, gb [O] = column sums of U (U [n, O], X [n, K]) -- both in one launch.  Returns SPK_ERR_ARG when the
// first product does not fit the MFMA tiles (widths that are not multiples of 4): call the two entry points separately then.
extern "C" int spk_gemm_pair_f32(const float* a, const float* w, int32_t trans, int64_t m, int32_t k, int32_t n_out, float* out, const float* U,
                                 const float* X, int64_t n, int32_t O, int32_t K, float* G, float* gb, float* ws, uint32_t* tickets, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int KC = trans ? n_out : k, NW = trans ? k : n_out;
  SPK_CHECK_ARG(m >= 0 && KC > 0 && NW > 0 && KC % 4 == 0 && NW % 4 == 0, "spk_gemm_pair_f32: widths K=%d N=%d are not multiples of 4", KC, NW);
  SPK_CHECK_ARG(aligned16(a) && aligned16(w) && aligned16(out), "spk_gemm_pair_f32: 16-byte alignment required");
  int32_t S, tiles;
  int64_t wsf;
  int rc = spk_gemm_tn_plan(n, O, K, &S, &wsf, &tiles);
  if (rc) return rc;
  SPK_CHECK_ARG(G != nullptr && (n == 0 || (U && X)) && (m == 0 || (a && w && out)), "spk_gemm_pair_f32: null pointer");
  SPK_CHECK_ARG(S == 1 || (ws && tickets), "spk_gemm_pair_f32: workspace / ticket buffer required for %d slices", S);
  SPK_CHECK_ARG(tiles <= 4096, "spk_gemm_pair_f32: %d output tiles (max 4096)", tiles);
  SpkProfScope prof("gemm_pair", stream);
  const int64_t ntasks = ((m + 31) / 32) * ((NW + 31) / 32);
  int64_t nb = (ntasks + TN_WAVES - 1) / TN_WAVES;
  if (nb > 2 * (int64_t)spk_num_cus()) nb = 2 * (int64_t)spk_num_cus();
  const int nblk_dense = (int)nb;
  GemmTnArgs tn = spk_gemm_tn_args(U, X, n, O, K, S, tiles, G, gb, ws, tickets);
  const unsigned grid = (unsigned)(nblk_dense + tiles * S);
  if (trans) hipLaunchKernelGGL((k_gemm_pair<true>), dim3(grid), dim3(64 * TN_WAVES), 0, stream, a, w, out, m, KC, NW, ntasks, nblk_dense, tn);
  else hipLaunchKernelGGL((k_gemm_pair<false>), dim3(grid), dim3(64 * TN_WAVES), 0, stream, a, w, out, m, KC, NW, ntasks, nblk_dense, tn);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ---------------------------------------------------------------- Atomwise head (atomistic/atomwise.py:69-88)
// Default head = build_mlp(n_in, 1, n_layers=2): y_n = w2 . act(W1 x_n + b1) + b2, E[idx_m[n]] += y_n.
// One wave owns 32 atoms and walks all hidden tiles; the second (width-1) layer is a per-lane dot
// product over the accumulator registers + one cross-half shuffle; the molecule sum is one float
// atomic per atom (n_atoms atomics on n_mol addresses).
template <int ACT>
__global__ __launch_bounds__(256) void k_atomwise_fwd(
    const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const int64_t* __restrict__ idx_m,
    int64_t M, int KC, int H, int64_t n_mol, float* __restrict__ pre, float* __restrict__ y_atom,
    float* __restrict__ E) {
  // molecule sums: segmented scan inside the wave, then one LDS slot per molecule of the block's
  // 128 consecutive atoms, then ONE global atomic per (block, molecule) -- float atomics that meet on
  // one cache line serialise at the memory side, so a 32 k-atom box must not issue 32 k of them.
  __shared__ float s_sum[128];
  __shared__ int s_flag[128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  const int nug = KC / 8;
  const int nch = (nug + DCH - 1) / DCH;
  const int64_t ntiles = (M + 31) / 32;
  for (int64_t mt0 = (int64_t)blockIdx.x * 4; mt0 < ntiles; mt0 += (int64_t)gridDim.x * 4) {
    const int64_t mt = mt0 + wv;
    const int64_t m = mt * 32 + el;
    const bool valid = m < M;
    const int64_t mc = valid ? m : (M - 1);
    const float* inrow = x + mc * KC;
    const int64_t mol = (idx_m && valid) ? idx_m[m] : -1;
    const int64_t mol0 = idx_m ? idx_m[mt0 * 32] : 0;
    if (threadIdx.x < 128) { s_sum[threadIdx.x] = 0.f; s_flag[threadIdx.x] = 0; }
    __syncthreads();
    float part = 0.f;
    if (mt < ntiles) {
      for (int t = 0; t < H / 32; ++t) {
        f32x4 a0[DCH], b0[DCH], a1[DCH], b1v[DCH];
        dense_load_chunk<false, SPK_ACT_NONE>(a0, b0, 0, nug, inrow, nullptr, w1, KC, H, t, el, hi);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b1 ? b1[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi] : 0.f;
        for (int c = 0; c < nch; c += 2) {
          if (c + 1 < nch) dense_load_chunk<false, SPK_ACT_NONE>(a1, b1v, c + 1, nug, inrow, nullptr, w1, KC, H, t, el, hi);
          acc = dense_mfma_chunk(a0, b0, c, nug, acc);
          if (c + 2 < nch) dense_load_chunk<false, SPK_ACT_NONE>(a0, b0, c + 2, nug, inrow, nullptr, w1, KC, H, t, el, hi);
          if (c + 1 < nch) acc = dense_mfma_chunk(a1, b1v, c + 1, nug, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f0 = 32 * t + 8 * q + 4 * hi;
          const f32x4 wv2 = *(const f32x4*)(w2 + f0);
          f32x4 o;
          o.x = acc[4 * q]; o.y = acc[4 * q + 1]; o.z = acc[4 * q + 2]; o.w = acc[4 * q + 3];
          if (pre && valid) *(f32x4*)(pre + m * H + f0) = o;
          part += wv2.x * spk_act<ACT>(o.x) + wv2.y * spk_act<ACT>(o.y) + wv2.z * spk_act<ACT>(o.z) + wv2.w * spk_act<ACT>(o.w);
        }
      }
      part += __shfl_xor(part, 32, 64);
      const float y = part + (b2 ? b2[0] : 0.f);
      if (valid && hi == 0 && y_atom) y_atom[m] = y;
      if (E) {
        // segmented inclusive scan over the 32 atoms of the tile (segments = runs of equal molecule id)
        const int64_t mprev = __shfl_up(mol, 1, 64);
        int head = (el == 0 || mprev != mol) ? 1 : 0;
        float v = (mol >= 0 && mol < n_mol) ? y : 0.f;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
          const float vu = __shfl_up(v, d, 64);
          const int hu = __shfl_up(head, d, 64);
          if (el >= d && !head) { v += vu; head = hu; }
        }
        const int64_t mnext = __shfl_down(mol, 1, 64);
        const bool last = (el == 31) || (mnext != mol);
        if (hi == 0 && last && mol >= 0 && mol < n_mol) {
          const int64_t rel = mol - mol0;
          if (rel >= 0 && rel < 128) { atomicAdd(&s_sum[rel], v); s_flag[rel] = 1; }
          else unsafeAtomicAdd(&E[mol], v);
        }
      }
    }
    __syncthreads();
    if (E && threadIdx.x < 128 && s_flag[threadIdx.x]) unsafeAtomicAdd(&E[mol0 + threadIdx.x], s_sum[threadIdx.x]);
    __syncthreads();
  }
}

// Same head for the common widths (n_hidden 32 or 64, n_in <= 512): 16-atom tiles on v_mfma_f32_16x16x4_f32,
// the work of a tile split over all 4 waves (2 pairs of 16-feature tiles x 2 halves of the contraction)
// instead of one wave walking everything, the x tile staged once in LDS.  Twice as many, four times
// shorter workgroups: 17 -> ~8 us at 5 k atoms.  A workgroup owns a contiguous range of tiles and keeps
// the molecule sums of its atoms in LDS until the end (one global atomic per workgroup and molecule).
template <int ACT>
__global__ __launch_bounds__(256) void k_atomwise_fwd16(
    const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const int64_t* __restrict__ idx_m,
    int64_t M, int F, int H, int64_t n_mol, float* __restrict__ pre, float* __restrict__ y_atom,
    float* __restrict__ E, int64_t tiles_per_block) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float s_sum[128];
  __shared__ int s_flag[128];
  const int ld = F + 4;
  float* xs = smem;                       // [16][F + 4]
  float* red = xs + 16 * ld;              // [2 pairs][2 tiles][64 lanes][4]: partial sums of the upper k half
  float* sy = red + 1024;                 // [16] per-atom outputs of the tile
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 4, el = lane & 15;
  const int p = wv & 1, khalf = wv >> 1;
  const int64_t ntiles = (M + 15) / 16;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_block;
  int64_t t1 = t0 + tiles_per_block;
  if (t1 > ntiles) t1 = ntiles;
  if (threadIdx.x < 128) { s_sum[threadIdx.x] = 0.f; s_flag[threadIdx.x] = 0; }
  const int64_t mol0 = (idx_m && t0 < ntiles) ? idx_m[t0 * 16] : 0;
  const bool has_pair = 32 * p < H;
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t m0 = tile * 16;
    const int q4 = F / 4;
    for (int s = threadIdx.x; s < 16 * q4; s += 256) {
      const int row = s / q4, c4 = s - row * q4;
      int64_t mm = m0 + row;
      if (mm >= M) mm = M - 1;
      *(f32x4*)(xs + row * ld + 4 * c4) = *(const f32x4*)(x + mm * F + 4 * c4);
    }
    if (threadIdx.x < 16) sy[threadIdx.x] = 0.f;
    __syncthreads();
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (has_pair) {
      const int kb = khalf * (F / 2);
      const float* wr0 = w1 + (int64_t)(32 * p + el) * F + kb + 4 * h;
      const float* wr1 = wr0 + 16 * (int64_t)F;
      const float* br = xs + el * ld + kb + 4 * h;
      for (int u = 0; u < F / 32; ++u) {
        const f32x4 b = *(const f32x4*)(br + 16 * u);
        const f32x4 a0 = *(const f32x4*)(wr0 + 16 * u), a1 = *(const f32x4*)(wr1 + 16 * u);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc1, 0, 0, 0);
      }
      if (khalf == 1) {
        *(f32x4*)(red + ((p * 2 + 0) * 64 + lane) * 4) = acc0;
        *(f32x4*)(red + ((p * 2 + 1) * 64 + lane) * 4) = acc1;
      }
    }
    __syncthreads();
    if (has_pair && khalf == 0) {
      const int64_t m = m0 + el;
      const bool valid = m < M;
      float part = 0.f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int f0 = 32 * p + 16 * k + 4 * h;
        f32x4 o = (k ? acc1 : acc0) + *(const f32x4*)(red + ((p * 2 + k) * 64 + lane) * 4);
        if (b1) o += *(const f32x4*)(b1 + f0);
        if (pre && valid) *(f32x4*)(pre + m * H + f0) = o;
        const f32x4 wv2 = *(const f32x4*)(w2 + f0);
        part += wv2.x * spk_act<ACT>(o.x) + wv2.y * spk_act<ACT>(o.y) + wv2.z * spk_act<ACT>(o.z) + wv2.w * spk_act<ACT>(o.w);
      }
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      if (h == 0) atomicAdd(&sy[el], part);
    }
    __syncthreads();
    if (wv == 0 && lane < 16) {
      const int64_t m = m0 + lane;
      const bool valid = m < M;
      const float y = sy[lane] + (b2 ? b2[0] : 0.f);
      if (valid && y_atom) y_atom[m] = y;
      if (E) {
        const int64_t mol = valid ? idx_m[m] : -1;
        const int64_t mprev = __shfl_up(mol, 1, 16);
        int head = (lane == 0 || mprev != mol) ? 1 : 0;
        float v = (mol >= 0 && mol < n_mol) ? y : 0.f;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          const float vu = __shfl_up(v, d, 16);
          const int hu = __shfl_up(head, d, 16);
          if (lane >= d && !head) { v += vu; head = hu; }
        }
        const int64_t mnext = __shfl_down(mol, 1, 16);
        const bool last = (lane == 15) || (mnext != mol);
        if (last && mol >= 0 && mol < n_mol) {
          const int64_t rel = mol - mol0;
          if (rel >= 0 && rel < 128) { s_sum[rel] += v; s_flag[rel] = 1; }   // only this wave touches s_sum inside the loop
          else unsafeAtomicAdd(&E[mol], v);
        }
      }
    }
  }
  __syncthreads();
  if (E && threadIdx.x < 128 && s_flag[threadIdx.x]) unsafeAtomicAdd(&E[mol0 + threadIdx.x], s_sum[threadIdx.x]);
}

// gx[n][k] = sum_f s_n w2[f] act'(pre[n][f]) W1[f][k],  s_n = gE[idx_m[n]] (+ gy_atom[n]).
template <int ACT>
__global__ __launch_bounds__(256) void k_atomwise_bwd(
    const float* __restrict__ gE, const float* __restrict__ gy_atom, const float* __restrict__ pre,
    const float* __restrict__ w1, const float* __restrict__ w2, const int64_t* __restrict__ idx_m,
    int64_t M, int KC /*=H*/, int NW /*=n_in*/, int64_t n_mol, float* __restrict__ gx, int64_t ntasks) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  const int tcount = NW / 32;
  const int nug = KC / 8;
  for (int64_t task = blockIdx.x * 4 + wv; task < ntasks; task += (int64_t)gridDim.x * 4) {
    const int64_t mt = task / tcount;
    const int t = (int)(task % tcount);
    const int64_t m = mt * 32 + el;
    const bool valid = m < M;
    const int64_t mc = valid ? m : (M - 1);
    float s = gy_atom ? gy_atom[mc] : 0.f;
    if (gE && idx_m) {
      const int64_t mol = idx_m[mc];
      if (mol >= 0 && mol < n_mol) s += gE[mol];
    }
    const float* prow = pre + mc * KC;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int ug0 = 0; ug0 < nug; ug0 += DCH) {
      f32x4 av[DCH], bv[DCH];
#pragma unroll
      for (int u = 0; u < DCH; ++u) {
        const int ug = ug0 + u;
        if (ug < nug) {
          const int kk0 = 8 * ug + 4 * hi;
          const f32x4 pv = *(const f32x4*)(prow + kk0);
          const f32x4 v2 = *(const f32x4*)(w2 + kk0);
          f32x4 b;
          b.x = s * v2.x * spk_act_grad<ACT>(pv.x); b.y = s * v2.y * spk_act_grad<ACT>(pv.y);
          b.z = s * v2.z * spk_act_grad<ACT>(pv.z); b.w = s * v2.w * spk_act_grad<ACT>(pv.w);
          bv[u] = b;
          const float* wp = w1 + (int64_t)kk0 * NW + 32 * t + el;
          f32x4 a4;
          a4.x = wp[0]; a4.y = wp[NW]; a4.z = wp[2 * (int64_t)NW]; a4.w = wp[3 * (int64_t)NW];
          av[u] = a4;
        }
      }
#pragma unroll
      for (int u = 0; u < DCH; ++u) {
        if (ug0 + u < nug) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
      }
    }
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
        o.x = acc[4 * q]; o.y = acc[4 * q + 1]; o.z = acc[4 * q + 2]; o.w = acc[4 * q + 3];
        *(f32x4*)(gx + m * NW + 32 * t + 8 * q + 4 * hi) = o;
      }
    }
  }
}

static bool atomwise_shape_ok(int n_in, int n_hidden) { return n_in > 0 && n_hidden > 0 && n_in % 32 == 0 && n_hidden % 32 == 0; }

extern "C" int spk_atomwise_supported(int32_t n_in, int32_t n_hidden, int32_t act) {
  return atomwise_shape_ok(n_in, n_hidden) && (act == SPK_ACT_SSP || act == SPK_ACT_SILU) ? 1 : 0;
}

extern "C" int spk_atomwise_fwd_f32(const float* x, const float* w1, const float* b1, const float* w2,
                                    const float* b2, const int64_t* idx_m, int64_t n_atoms,
                                    int32_t n_in, int32_t n_hidden, int32_t act, int64_t n_mol,
                                    float* pre, float* y_atom, float* E, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n_atoms >= 0 && n_mol >= 0, "spk_atomwise_fwd_f32: bad sizes");
  SPK_CHECK_ARG(spk_atomwise_supported(n_in, n_hidden, act), "spk_atomwise_fwd_f32: head %d -> %d -> 1 (act %d) not supported by the fused kernel", n_in, n_hidden, act);
  SPK_CHECK_ARG((E == nullptr) == (idx_m == nullptr), "spk_atomwise_fwd_f32: E and idx_m go together");
  SPK_CHECK_ARG(E != nullptr || y_atom != nullptr, "spk_atomwise_fwd_f32: no output requested");
  if (E && n_mol > 0) { int _zr = spk_zero_async(E, (size_t)n_mol * sizeof(float), stream); if (_zr) return _zr; }
  if (n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(x && w1 && w2, "spk_atomwise_fwd_f32: null pointer");
  SPK_CHECK_ARG(aligned16(x) && aligned16(w1) && aligned16(w2) && aligned16(pre), "spk_atomwise_fwd_f32: 16-byte alignment required");
  SpkProfScope prof("atomwise_fwd", stream);
  // small systems (at most one 16-atom tile per workgroup slot): the 4-wave-per-tile kernel; large ones keep the
  // one-wave-per-32-atoms kernel, whose 128-atom workgroups issue 4x fewer same-address molecule atomics
  if ((n_hidden == 32 || n_hidden == 64) && n_in <= 512 && (!b1 || aligned16(b1)) && spk_get_variant() != SPK_VARIANT_SIMPLE &&
      (n_atoms + 15) / 16 <= 2 * (int64_t)spk_num_cus()) {
    const int64_t nt16 = (n_atoms + 15) / 16;
    const int64_t nblk = nt16 < 2 * (int64_t)spk_num_cus() ? nt16 : 2 * (int64_t)spk_num_cus();
    const int64_t tpb = (nt16 + nblk - 1) / nblk;
    const int grid16 = (int)((nt16 + tpb - 1) / tpb);
    const size_t lds = sizeof(float) * (16 * (size_t)(n_in + 4) + 1024 + 16);
    if (act == SPK_ACT_SILU)
      hipLaunchKernelGGL((k_atomwise_fwd16<SPK_ACT_SILU>), dim3(grid16), dim3(256), lds, stream, x, w1, b1, w2, b2, idx_m, n_atoms, n_in, n_hidden, n_mol, pre, y_atom, E, tpb);
    else
      hipLaunchKernelGGL((k_atomwise_fwd16<SPK_ACT_SSP>), dim3(grid16), dim3(256), lds, stream, x, w1, b1, w2, b2, idx_m, n_atoms, n_in, n_hidden, n_mol, pre, y_atom, E, tpb);
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  const int64_t ntiles = (n_atoms + 31) / 32;
  const int grid = spk_grid_for(ntiles, 4, spk_num_cus() * 8);
  if (act == SPK_ACT_SILU)
    hipLaunchKernelGGL((k_atomwise_fwd<SPK_ACT_SILU>), dim3(grid), dim3(256), 0, stream, x, w1, b1, w2, b2, idx_m, n_atoms, n_in, n_hidden, n_mol, pre, y_atom, E);
  else
    hipLaunchKernelGGL((k_atomwise_fwd<SPK_ACT_SSP>), dim3(grid), dim3(256), 0, stream, x, w1, b1, w2, b2, idx_m, n_atoms, n_in, n_hidden, n_mol, pre, y_atom, E);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_atomwise_bwd_f32(const float* gE, const float* gy_atom, const float* pre,
                                    const float* w1, const float* w2, const int64_t* idx_m,
                                    int64_t n_atoms, int32_t n_in, int32_t n_hidden, int32_t act,
                                    int64_t n_mol, float* gx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n_atoms >= 0, "spk_atomwise_bwd_f32: bad sizes");
  SPK_CHECK_ARG(spk_atomwise_supported(n_in, n_hidden, act), "spk_atomwise_bwd_f32: head %d -> %d -> 1 (act %d) not supported by the fused kernel", n_in, n_hidden, act);
  if (n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(pre && w1 && w2 && gx && (gy_atom || (gE && idx_m)), "spk_atomwise_bwd_f32: null pointer");
  SPK_CHECK_ARG(aligned16(pre) && aligned16(w1) && aligned16(w2) && aligned16(gx), "spk_atomwise_bwd_f32: 16-byte alignment required");
  SpkProfScope prof("atomwise_bwd", stream);
  const int64_t ntasks = ((n_atoms + 31) / 32) * (n_in / 32);
  const int grid = spk_grid_for(ntasks, 4, spk_num_cus() * 8);
  if (act == SPK_ACT_SILU)
    hipLaunchKernelGGL((k_atomwise_bwd<SPK_ACT_SILU>), dim3(grid), dim3(256), 0, stream, gE, gy_atom, pre, w1, w2, idx_m, n_atoms, n_hidden, n_in, n_mol, gx, ntasks);
  else
    hipLaunchKernelGGL((k_atomwise_bwd<SPK_ACT_SSP>), dim3(grid), dim3(256), 0, stream, gE, gy_atom, pre, w1, w2, idx_m, n_atoms, n_hidden, n_in, n_mol, gx, ntasks);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// internal C++ entry used by the whole-representation drivers
int spk_dense_internal(const float* in, const float* pre_in, const float* w, const float* b,
                       const float* res, float* out, float* pre_out, int64_t M, int KC, int NW,
                       int act, bool trans, int pro, hipStream_t stream) {
  return dense_dispatch(in, pre_in, w, b, res, out, pre_out, M, KC, NW, act, trans, pro, stream,
                        "spk_dense");
}

// ---------------------------------------------------------------- Dense layers on (value, tangent) pairs
// The force-matching engine (spk_fm_engine.h) carries every activation as a pair (value, tangent along t = -dL/dF).  A Dense layer acts
// on both with the same weights, and the activation couples them element by element (tangent: act'(a) a_t; reverse of the pair:
// g_a = g_z act'(a) + h_z act''(a) a_t, h_a = h_z act'(a)).  As separate launches that is two Dense launches and an element-wise one per
// layer, each a few microseconds of latency at training sizes (168 - 2 500 rows).  Here ONE workgroup owns the 32 x 32 tile of both
// members: the weight operand is fetched once, the two accumulators meet in the epilogue.
struct DenseDualArgs {
  const float *in_v, *in_t;       // [M, KC]; in_v == NULL: tangent member only
  const float *w, *b;
  const float *res_v, *res_t;     // [M, NW] or NULL
  const float *epre_v, *epre_t;   // [M, NW] saved pre-activations (modes TANGENT / DUAL_BWD)
  const float *fc, *fc1;          // [M] row scale of mode FWD: y_v = p_v fc, y_t = p_t fc + p_v fc'
  float *out_v, *out_t, *pre_v, *pre_t;
  int64_t M;
  int KC, NW, act, mode;
};

__device__ __forceinline__ float dd_act(int act, int order, float z) {
  if (act == SPK_ACT_NONE) return order == 0 ? z : (order == 1 ? 1.f : 0.f);
  const float s = spk_sigmoid(z);
  if (act == SPK_ACT_SSP) {
    if (order == 0) return spk_ssp(z);
    return order == 1 ? s : s * (1.f - s);
  }
  if (order == 0) return z * s;
  if (order == 1) return s * (1.f + z * (1.f - s));
  return s * (1.f - s) * (2.f + z * (1.f - 2.f * s));
}

template <bool TRANS, int CH>
__device__ __forceinline__ void dd_load_chunk(f32x4 (&av)[CH], f32x4 (&bt)[CH], f32x4 (&bvv)[CH], int c, int nug, const float* __restrict__ row_t,
                                              const float* __restrict__ row_v, bool dual, const float* __restrict__ w, int KC, int NW, int t, int el, int hi) {
  const bool rin = 32 * t + el < NW;
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int ug = c * CH + u;
    if (ug < nug) {
      const int kk0 = 8 * ug + 4 * hi;
      const bool kin = kk0 < KC;
      f32x4 x{0.f, 0.f, 0.f, 0.f}, y{0.f, 0.f, 0.f, 0.f}, a4{0.f, 0.f, 0.f, 0.f};
      if (kin) x = *(const f32x4*)(row_t + kk0);
      if (kin && dual) y = *(const f32x4*)(row_v + kk0);
      if (kin && rin) {
        if (!TRANS) {
          a4 = *(const f32x4*)(w + (int64_t)(32 * t + el) * KC + kk0);
        } else {
          const float* wp = w + (int64_t)kk0 * NW + 32 * t + el;
          a4.x = wp[0]; a4.y = wp[NW]; a4.z = wp[2 * (int64_t)NW]; a4.w = wp[3 * (int64_t)NW];
        }
      }
      bt[u] = x; bvv[u] = y; av[u] = a4;
    }
  }
}

// Four waves: wave w contracts the chunks w, w + 4, ... of the tile; all partial tiles go through LDS, and wave w finishes the column
// group q = w (8 of the 32 output features) -- the epilogue (up to three transcendental activations per element) runs on all four waves
// instead of one (first version, epilogue on wave 0 only: 10 - 13 us per launch against 5 - 6 us of the plain kernel).
// Long contractions (K > 128) run with eight waves: waves 4 .. 7 hand their partial tiles to waves 0 .. 3 through the same LDS buffer first.
template <bool TRANS, int CH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_dense_dual_sk(DenseDualArgs a) {
  __shared__ float red[2][4][32][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  const int KC = a.KC, NW = a.NW;
  const int tcount = (NW + 31) / 32;
  const int nug = (KC + 7) / 8;
  const int nch = (nug + CH - 1) / CH;
  const int64_t task = blockIdx.x;
  const int64_t mt = task / tcount;
  const int t = (int)(task % tcount);
  const int64_t m = mt * 32 + el;
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : (a.M - 1);
  const bool dual = a.in_v != nullptr;
  const float* row_t = a.in_t + mc * KC;
  const float* row_v = dual ? a.in_v + mc * KC : row_t;
  f32x16 acc_v, acc_t;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_v[r] = 0.f; acc_t[r] = 0.f; }
  f32x4 a0[CH], t0[CH], v0[CH], a1[CH], t1[CH], v1[CH];
  if (wv < nch) dd_load_chunk<TRANS, CH>(a0, t0, v0, wv, nug, row_t, row_v, dual, a.w, KC, NW, t, el, hi);
  // epilogue operands of the wave's column group, requested ahead of the contraction
  const int col = 32 * t + 8 * wv + 4 * hi;
  const bool fin = valid && wv < 4 && col < NW;
  const int64_t off = m * NW + col;
  const f32x4 z4{0.f, 0.f, 0.f, 0.f};
  const f32x4 ev = (a.epre_v && fin) ? *(const f32x4*)(a.epre_v + off) : z4;
  const f32x4 et = (a.epre_t && fin) ? *(const f32x4*)(a.epre_t + off) : z4;
  const f32x4 rv = (a.res_v && fin) ? *(const f32x4*)(a.res_v + off) : z4;
  const f32x4 rt = (a.res_t && fin) ? *(const f32x4*)(a.res_t + off) : z4;
  f32x4 bq = z4;
  if (a.b && fin) { bq.x = a.b[col]; bq.y = a.b[col + 1]; bq.z = a.b[col + 2]; bq.w = a.b[col + 3]; }
  float fcm = 0.f, fc1m = 0.f;
  if (a.fc && valid) { fcm = a.fc[m]; fc1m = a.fc1[m]; }
  for (int c = wv; c < nch; c += 2 * WAVES) {
    if (c + WAVES < nch) dd_load_chunk<TRANS, CH>(a1, t1, v1, c + WAVES, nug, row_t, row_v, dual, a.w, KC, NW, t, el, hi);
    acc_t = dense_mfma_chunk(a0, t0, c, nug, acc_t);
    if (dual) acc_v = dense_mfma_chunk(a0, v0, c, nug, acc_v);
    if (c + 2 * WAVES < nch) dd_load_chunk<TRANS, CH>(a0, t0, v0, c + 2 * WAVES, nug, row_t, row_v, dual, a.w, KC, NW, t, el, hi);
    if (c + WAVES < nch) {
      acc_t = dense_mfma_chunk(a1, t1, c + WAVES, nug, acc_t);
      if (dual) acc_v = dense_mfma_chunk(a1, v1, c + WAVES, nug, acc_v);
    }
  }
  if (WAVES == 8) {
    if (wv >= 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * hi;
        red[0][wv - 4][rr][el] = acc_t[r];
        if (dual) red[1][wv - 4][rr][el] = acc_v[r];
      }
    }
    __syncthreads();
    if (wv < 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * hi;
        acc_t[r] += red[0][wv][rr][el];
        if (dual) acc_v[r] += red[1][wv][rr][el];
      }
    }
    __syncthreads();
  }
  if (wv < 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      red[0][wv][rr][el] = acc_t[r];
      if (dual) red[1][wv][rr][el] = acc_v[r];
    }
  }
  __syncthreads();
  if (!fin) return;
  const int act = a.act, mode = a.mode;
  f32x4 ov, ot, pv, pt;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int rr = 8 * wv + 4 * hi + v;
    float st = 0.f, sv = 0.f;
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {      // wave order: deterministic
      st += red[0][q2][rr][el];
      if (dual) sv += red[1][q2][rr][el];
    }
    if (mode == SPK_DD_FWD) {
      sv += bq[v];
      pv[v] = sv; pt[v] = st;
      if (a.fc) { ov[v] = sv * fcm; ot[v] = st * fcm + sv * fc1m; }
      else { ov[v] = dd_act(act, 0, sv); ot[v] = dd_act(act, 1, sv) * st; }
    } else if (mode == SPK_DD_TANGENT) {
      pt[v] = st; pv[v] = 0.f; ov[v] = 0.f;
      ot[v] = dd_act(act, 1, ev[v]) * st;
    } else {      // SPK_DD_DUAL_BWD: (sv, st) = (g_z, h_z)
      const float p = ev[v], a1v = dd_act(act, 1, p);
      pv[v] = sv; pt[v] = st;
      ov[v] = sv * a1v + st * dd_act(act, 2, p) * et[v];
      ot[v] = st * a1v;
    }
  }
  if (a.pre_v && dual) *(f32x4*)(a.pre_v + off) = pv;
  if (a.pre_t) *(f32x4*)(a.pre_t + off) = pt;
  if (dual) { ov += rv; *(f32x4*)(a.out_v + off) = ov; }
  ot += rt;
  *(f32x4*)(a.out_t + off) = ot;
}

// Many tiles (pair rows of larger batches: the filter networks of a 128-frame training step have 40 k rows): the forward pair on the
// grid-stride form of k_dense_mfma -- one wave walks 32 x 32 tiles, whole contraction, both accumulators, epilogue in the wave.  At this
// size the separate route is two Dense launches plus an element-wise pass over 2 x [rows, n_out] that is pure memory traffic (PaiNN's
// cutoff product: 150 us of a 2.5 ms step).
__global__ __launch_bounds__(256) void k_dense_dual_tiles(DenseDualArgs a, int64_t ntasks) {
  constexpr int CH = 4;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 5, el = lane & 31;
  const int KC = a.KC, NW = a.NW;
  const int tcount = (NW + 31) / 32;
  const int nug = (KC + 7) / 8;
  const int nch = (nug + CH - 1) / CH;
  const int act = a.act;
  for (int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); task < ntasks; task += (int64_t)gridDim.x * 4) {
    const int64_t mt = task / tcount;
    const int t = (int)(task % tcount);
    const int64_t m = mt * 32 + el;
    const bool valid = m < a.M;
    const int64_t mc = valid ? m : (a.M - 1);
    const float* row_t = a.in_t + mc * KC;
    const float* row_v = a.in_v + mc * KC;
    f32x4 a0[CH], t0[CH], v0[CH], a1[CH], t1[CH], v1[CH];
    dd_load_chunk<false, CH>(a0, t0, v0, 0, nug, row_t, row_v, true, a.w, KC, NW, t, el, hi);
    float fcm = 0.f, fc1m = 0.f;
    if (a.fc && valid) { fcm = a.fc[m]; fc1m = a.fc1[m]; }
    f32x16 acc_v, acc_t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
      acc_v[r] = (a.b && col < NW) ? a.b[col] : 0.f;
      acc_t[r] = 0.f;
    }
    for (int c = 0; c < nch; c += 2) {
      if (c + 1 < nch) dd_load_chunk<false, CH>(a1, t1, v1, c + 1, nug, row_t, row_v, true, a.w, KC, NW, t, el, hi);
      acc_v = dense_mfma_chunk(a0, v0, c, nug, acc_v);
      acc_t = dense_mfma_chunk(a0, t0, c, nug, acc_t);
      if (c + 2 < nch) dd_load_chunk<false, CH>(a0, t0, v0, c + 2, nug, row_t, row_v, true, a.w, KC, NW, t, el, hi);
      if (c + 1 < nch) {
        acc_v = dense_mfma_chunk(a1, v1, c + 1, nug, acc_v);
        acc_t = dense_mfma_chunk(a1, t1, c + 1, nug, acc_t);
      }
    }
    if (!valid) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 32 * t + 8 * q + 4 * hi;
      if (col >= NW) continue;
      const int64_t off = m * NW + col;
      f32x4 pv, pt, ov, ot;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float sv = acc_v[4 * q + v], st = acc_t[4 * q + v];
        pv[v] = sv; pt[v] = st;
        if (a.fc) { ov[v] = sv * fcm; ot[v] = st * fcm + sv * fc1m; }
        else { ov[v] = dd_act(act, 0, sv); ot[v] = dd_act(act, 1, sv) * st; }
      }
      if (a.pre_v) *(f32x4*)(a.pre_v + off) = pv;
      if (a.pre_t) *(f32x4*)(a.pre_t + off) = pt;
      if (a.res_v) ov += *(const f32x4*)(a.res_v + off);
      if (a.res_t) ot += *(const f32x4*)(a.res_t + off);
      *(f32x4*)(a.out_v + off) = ov;
      *(f32x4*)(a.out_t + off) = ot;
    }
  }
}

extern "C" int spk_dense_dual_supported(int64_t m, int32_t k_in, int32_t n_out) {
  if (m <= 0 || k_in <= 0 || n_out <= 0 || k_in % 4 || n_out % 4) return 0;
  const int64_t ntasks = ((m + 31) / 32) * ((n_out + 31) / 32);
  return ntasks <= 4 * (int64_t)spk_num_cus() && spk_get_variant() != SPK_VARIANT_SIMPLE;      // (the bound of the one-tile-per-workgroup Dense kernel)
}
// the forward pair (mode FWD, trans = 0) has a second kernel for any number of tiles
extern "C" int spk_dense_dual_fwd_supported(int64_t m, int32_t k_in, int32_t n_out) {
  return m > 0 && k_in > 0 && n_out > 0 && k_in % 4 == 0 && n_out % 4 == 0 && spk_get_variant() != SPK_VARIANT_SIMPLE;
}

extern "C" int spk_dense_dual_f32(const spk_dense_dual_t* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(d != nullptr, "spk_dense_dual_f32: null description");
  SPK_CHECK_ARG(d->m >= 0 && d->k_in > 0 && d->n_out > 0, "spk_dense_dual_f32: bad sizes");
  if (d->m == 0) return SPK_OK;
  const bool small = spk_dense_dual_supported(d->m, d->k_in, d->n_out) != 0;
  const bool big_fwd = !small && d->mode == SPK_DD_FWD && !d->trans && spk_dense_dual_fwd_supported(d->m, d->k_in, d->n_out);
  SPK_CHECK_ARG(small || big_fwd, "spk_dense_dual_f32: shape m=%lld k=%d n=%d outside the pair kernels (see spk_dense_dual_supported / _fwd_supported)",
                (long long)d->m, d->k_in, d->n_out);
  SPK_CHECK_ARG(d->mode == SPK_DD_FWD || d->mode == SPK_DD_TANGENT || d->mode == SPK_DD_DUAL_BWD, "spk_dense_dual_f32: unknown mode %d", d->mode);
  SPK_CHECK_ARG(d->act == SPK_ACT_NONE || d->act == SPK_ACT_SSP || d->act == SPK_ACT_SILU, "spk_dense_dual_f32: unknown activation %d", d->act);
  SPK_CHECK_ARG(d->x_t && d->w && d->y_t, "spk_dense_dual_f32: null pointer");
  SPK_CHECK_ARG((d->mode == SPK_DD_TANGENT) == (d->x_v == nullptr), "spk_dense_dual_f32: x_v is given in every mode but TANGENT");
  SPK_CHECK_ARG(d->x_v == nullptr || d->y_v != nullptr, "spk_dense_dual_f32: y_v missing");
  SPK_CHECK_ARG(d->mode == SPK_DD_FWD || d->pre_v_in != nullptr, "spk_dense_dual_f32: the saved pre-activation is missing");
  SPK_CHECK_ARG(d->mode == SPK_DD_FWD || (!d->b && !d->fc), "spk_dense_dual_f32: bias / row scale belong to mode FWD");
  SPK_CHECK_ARG(!d->fc || (d->fc1 && d->act == SPK_ACT_NONE), "spk_dense_dual_f32: the row scale needs fc1 and a linear layer");
  SPK_CHECK_ARG(aligned16(d->x_v) && aligned16(d->x_t) && aligned16(d->w) && aligned16(d->res_v) && aligned16(d->res_t) && aligned16(d->pre_v_in) &&
                    aligned16(d->pre_t_in) && aligned16(d->y_v) && aligned16(d->y_t) && aligned16(d->pre_v) && aligned16(d->pre_t),
                "spk_dense_dual_f32: 16-byte alignment required");
  DenseDualArgs a;
  a.in_v = d->x_v; a.in_t = d->x_t; a.w = d->w; a.b = d->b; a.res_v = d->res_v; a.res_t = d->res_t; a.epre_v = d->pre_v_in; a.epre_t = d->pre_t_in;
  a.fc = d->fc; a.fc1 = d->fc1; a.out_v = d->y_v; a.out_t = d->y_t; a.pre_v = d->pre_v; a.pre_t = d->pre_t;
  a.M = d->m; a.KC = d->k_in; a.NW = d->n_out; a.act = d->act; a.mode = d->mode;
  SpkProfScope prof(d->mode == SPK_DD_FWD ? "dense_dual_fwd" : (d->mode == SPK_DD_TANGENT ? "dense_tangent" : "dense_dual_bwd"), stream);
  if (big_fwd) {
    const int64_t ntasks = ((d->m + 31) / 32) * ((d->n_out + 31) / 32);
    hipLaunchKernelGGL(k_dense_dual_tiles, dim3(spk_grid_for(ntasks, 4, spk_num_cus() * 8)), dim3(256), 0, stream, a, ntasks);
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  const unsigned nt = (unsigned)(((d->m + 31) / 32) * ((d->n_out + 31) / 32));
  const int nug = (d->k_in + 7) / 8;
#define SPK_DDL(T, CH, WV) hipLaunchKernelGGL((k_dense_dual_sk<T, CH, WV>), dim3(nt), dim3(64 * WV), 0, stream, a)
  if (d->trans) {
    if (nug <= 8) SPK_DDL(true, 2, 4);
    else if (nug <= 16) SPK_DDL(true, 4, 4);
    else SPK_DDL(true, 4, 8);
  } else {
    if (nug <= 8) SPK_DDL(false, 2, 4);
    else if (nug <= 16) SPK_DDL(false, 4, 4);
    else SPK_DDL(false, 4, 8);
  }
#undef SPK_DDL
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
