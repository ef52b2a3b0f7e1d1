// Fused chain of up to 3 Dense layers (nn/base.py:52-55 applied back to back, e.g. SchNet's
// f2out.0 -> f2out.1 (+residual) -> next in2f, or their input-gradient transposes) in ONE launch.
//
// At N ~ 5k atoms a single Dense is latency- and launch-bound (~0.2 GFLOP), so the win is in
// removing launches and HBM round trips: one workgroup (4 waves) owns a tile of 32 rows (atoms), the
// activations of the tile travel from layer to layer through LDS (two ping-pong buffers,
// [32][K+4] floats, conflict-free for the 16-byte accesses used), every wave computes the 32x32
// output tiles t = wave, wave+4, ... of a layer with the fp32 MFMA in the T-GEMM convention of
// spk_dense.hip (A = weights straight from L2, prefetched in chunks; B = activations from LDS).
// Epilogue per layer: bias, activation, optional pre-activation store, optional residual add,
// optional global store, optional "post" multiply by act'(pre) for the next layer's input
// (backward chains).  The kernel can also zero a buffer for the following edge kernel, which
// replaces a separate memset launch.
#include "spk_common.h"
#include "spk_split.h"

#define CH_MAXL 3
#define CH_MAXW 384

struct ChainLayerDev {
  const float* w;        // packed image of the [NW, KC] matrix (spk_pack_weight_f32)
  const float* b;        // [NW] or null
  const float* res;      // [M, NW] or null: added after the activation
  float* out;            // [M, NW] or null: result (after act and residual)
  float* pre_out;        // [M, NW] or null: pre-activation
  const float* post_pre; // [M, NW] or null: LDS copy handed to the next layer is multiplied by post_act'(post_pre)
  int KC, NW, act, trans, post_act;
};

struct ChainArgs {
  ChainLayerDev L[CH_MAXL];
  int n_layers;
  const float* in;       // [M, KC0]
  const float* in_pre;   // or null: input is multiplied by in_act'(in_pre) while it is staged
  int in_act;
  int64_t M;
  float* zero_ptr;       // optional buffer to clear (zero_count floats)
  int64_t zero_count;
  long long* dbg;        // tuning aid: cycle stamps of thread 0 of workgroup 0 (null in production)
};
#define CH_STAMP(n) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[n] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ float chain_act(int act, float x) {
  if (act == SPK_ACT_SSP) return spk_fast_ssp(x);
  if (act == SPK_ACT_SILU) return x * spk_sigmoid(x);
  return x;
}
__device__ __forceinline__ float chain_act_grad(int act, float x) {
  if (act == SPK_ACT_SSP) return spk_sigmoid(x);
  if (act == SPK_ACT_SILU) { const float s = spk_sigmoid(x); return s * (1.0f + x * (1.0f - s)); }
  return 1.0f;
}


#define CCH 8  // k-blocks per prefetch chunk (64 contraction indices, 32 MFMAs)

// A operands of one chunk (8 k-blocks).  The fused kernels read weights in the PACKED image made by
// spk_pack_weight_f32:  P[((t KB + ug) 64 + lane) 4 + v] = A[32 t + (lane & 31)][8 ug + 4 (lane >> 5) + v]
// (A[i][kk] = the [n_out, k] matrix of the layer), so every k-block of a tile is ONE fully coalesced
// 16-byte-per-lane load (1 KB per instruction, a quarter of the memory instructions of k-major rows).
__device__ __forceinline__ void chain_load_a(f32x4 (&av)[CCH], const float* __restrict__ w, int KB, int t, int c, int lane) {
  const f32x4* wp = (const f32x4*)w + ((int64_t)t * KB + c * CCH) * 64 + lane;
#pragma unroll
  for (int u = 0; u < CCH; ++u) av[u] = wp[u * 64];
}

__device__ __forceinline__ f32x16 chain_mfma(const f32x4 (&av)[CCH], int c, const float* __restrict__ brow, int hi,
                                             f32x16 acc) {
  f32x4 bv[CCH];
#pragma unroll
  for (int u = 0; u < CCH; ++u) bv[u] = *(const f32x4*)(brow + 8 * (c * CCH + u) + 4 * hi);
#pragma unroll
  for (int u = 0; u < CCH; ++u) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
  }
  return acc;
}

// Work of one wave = a stream of chunks (layer l, tile t = wave + 4 n, chunk c).  The weights do not
// depend on the data, so the A operands are always requested one chunk ahead -- across tile and layer
// boundaries too -- with two statically named register sets (KC % 128 == 0 => an even number of chunks
// per tile, so the parity never flips): a wave pays the L2 latency of the weights once per launch.
__device__ __forceinline__ bool chain_next_tile(const ChainArgs& a, int l, int t, int wv, int& l2, int& t2) {
  if (t + 4 < a.L[l].NW / 32) { l2 = l; t2 = t + 4; return true; }
  for (int q = l + 1; q < a.n_layers; ++q)
    if (wv < a.L[q].NW / 32) { l2 = q; t2 = wv; return true; }
  return false;
}

__global__ __launch_bounds__(256, 2) void k_dense_chain(ChainArgs a, int ld0, int ld1) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* buf0 = smem;               // [32][ld0]: the input tile, later the output of layer 1
  float* buf1 = smem + 32 * ld0;    // [32][ld1]: the output of layer 0
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;

  // optional clear of a buffer for the kernel that follows
  if (a.zero_ptr) {
    const int64_t n4 = a.zero_count / 4;
    f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int64_t s = blockIdx.x * 256 + threadIdx.x; s < n4; s += (int64_t)gridDim.x * 256) ((f32x4*)a.zero_ptr)[s] = z4;
    for (int64_t s = 4 * n4 + blockIdx.x * 256 + threadIdx.x; s < a.zero_count; s += (int64_t)gridDim.x * 256) a.zero_ptr[s] = 0.f;
  }

  const int64_t ntiles = (a.M + 31) / 32;
  for (int64_t mt = blockIdx.x; mt < ntiles; mt += gridDim.x) {
    const int64_t m0 = mt * 32;
    f32x4 a0[CCH], a1[CCH];
    {  // first chunk of this wave's stream, requested before the input tile is staged
      int l2 = 0, t2 = wv;
      bool have = wv < a.L[0].NW / 32;
      if (!have) have = chain_next_tile(a, 0, 1 << 20, wv, l2, t2);
      if (have) chain_load_a(a0, a.L[l2].w, a.L[l2].KC / 8, t2, 0, lane);
    }
    // ---- stage the input tile [32][KC0] into buf0 (coalesced 16-byte rows) in two batches of up to 6
    //      pieces per thread (KC0 <= 384), all loads of a batch in flight together
    {
      const int KC0 = a.L[0].KC;
      const int q4 = KC0 / 4;
      const int total = 32 * q4;
      for (int i0 = 0; i0 < 12; i0 += 6) {
        if (threadIdx.x + 256 * i0 >= total) break;
        f32x4 v[6], pz[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int s = threadIdx.x + 256 * (i0 + i);
          if (s < total) {
            const int row = s / q4, c4 = s - row * q4;
            int64_t mm = m0 + row;
            if (mm >= a.M) mm = a.M - 1;
            v[i] = *(const f32x4*)(a.in + mm * KC0 + 4 * c4);
            if (a.in_pre) pz[i] = *(const f32x4*)(a.in_pre + mm * KC0 + 4 * c4);
          }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int s = threadIdx.x + 256 * (i0 + i);
          if (s < total) {
            const int row = s / q4, c4 = s - row * q4;
            f32x4 x = v[i];
            if (a.in_pre) {
              x.x *= chain_act_grad(a.in_act, pz[i].x); x.y *= chain_act_grad(a.in_act, pz[i].y);
              x.z *= chain_act_grad(a.in_act, pz[i].z); x.w *= chain_act_grad(a.in_act, pz[i].w);
            }
            *(f32x4*)(buf0 + row * ld0 + 4 * c4) = x;
          }
        }
      }
    }
    __syncthreads();
    const int64_t m = m0 + el;
    const bool valid = m < a.M;
#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
      const ChainLayerDev& L = a.L[l];
      const bool last = (l == a.n_layers - 1);
      const float* brow = ((l & 1) ? buf1 + el * ld1 : buf0 + el * ld0);
      float* nxt = (l & 1) ? buf0 : buf1;
      const int ldn = (l & 1) ? ld0 : ld1;
      const int nch = L.KC / (8 * CCH);   // even
      const int tcount = L.NW / 32;
#pragma unroll 1
      for (int t = wv; t < tcount; t += 4) {
        // epilogue operands (residual, act' argument) are requested before the MFMAs of the tile
        f32x4 rv[4], pv[4];
        const bool has_res = L.res && valid, has_post = (!last) && L.post_pre && valid;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t off = m * L.NW + 32 * t + 8 * q + 4 * hi;
          rv[q] = has_res ? *(const f32x4*)(L.res + off) : f32x4{0.f, 0.f, 0.f, 0.f};
          pv[q] = has_post ? *(const f32x4*)(L.post_pre + off) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = L.b ? L.b[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi] : 0.f;
        for (int c = 0; c < nch; c += 2) {
          chain_load_a(a1, L.w, L.KC / 8, t, c + 1, lane);
          acc = chain_mfma(a0, c, brow, hi, acc);
          if (c + 2 < nch) {
            chain_load_a(a0, L.w, L.KC / 8, t, c + 2, lane);
          } else {
            int l2, t2;
            if (chain_next_tile(a, l, t, wv, l2, t2)) chain_load_a(a0, a.L[l2].w, a.L[l2].KC / 8, t2, 0, lane);
          }
          acc = chain_mfma(a1, c + 1, brow, hi, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * t + 8 * q + 4 * hi;
          const int64_t off = m * L.NW + col;
          f32x4 o;
          o.x = acc[4 * q]; o.y = acc[4 * q + 1]; o.z = acc[4 * q + 2]; o.w = acc[4 * q + 3];
          if (L.pre_out && valid) *(f32x4*)(L.pre_out + off) = o;
          if (L.act != SPK_ACT_NONE) {
            o.x = chain_act(L.act, o.x); o.y = chain_act(L.act, o.y); o.z = chain_act(L.act, o.z); o.w = chain_act(L.act, o.w);
          }
          if (has_res) o += rv[q];
          if (L.out && valid) *(f32x4*)(L.out + off) = o;
          if (!last) {
            if (has_post) {
              const f32x4 p = pv[q];
              o.x *= chain_act_grad(L.post_act, p.x); o.y *= chain_act_grad(L.post_act, p.y);
              o.z *= chain_act_grad(L.post_act, p.z); o.w *= chain_act_grad(L.post_act, p.w);
            }
            *(f32x4*)(nxt + el * ldn + col) = o;
          }
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------
// 16-row variant for small row counts (N ~ 5 k atoms = 168 tiles of 32 rows cannot fill 256 CUs and
// leave one wave per SIMD with every latency exposed): tiles of 16 rows on v_mfma_f32_16x16x4_f32
// (A: lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], acc[r] <-> row 4 (l >> 4) + r,
// col l & 15; k-step s = 4u + v uses kk = 16u + 4 (l >> 4) + v so a lane again fetches 4 consecutive kk
// with one 16-byte access).  Twice the workgroups, half the LDS and registers each => two workgroups per
// CU; a wave owns PAIRS of adjacent 16-feature tiles: two independent accumulators (the 16x16x4 form
// has a 40-cycle dependent latency against a 32-cycle issue) that share the activation operand.
#define C16 4   // u-steps per prefetch chunk: 64 contraction indices, 2 x 16 MFMAs

// same packed image: lane (h, el) of the 16x16x4 form needs A[32 p + 16 k + el][16 u + 4 h + v], which is the
// 16-byte unit of packed lane (h & 1) * 32 + 16 k + el in k-block 2 u + (h >> 1)
__device__ __forceinline__ void chain16_load_a(f32x4 (&av)[2][C16], const float* __restrict__ w, int KB, int p, int c,
                                               int el, int h) {
  const f32x4* wp = (const f32x4*)w + ((int64_t)p * KB + 2 * c * C16 + (h >> 1)) * 64 + (h & 1) * 32 + el;
#pragma unroll
  for (int u = 0; u < C16; ++u) {
    av[0][u] = wp[u * 128];
    av[1][u] = wp[u * 128 + 16];
  }
}

__device__ __forceinline__ void chain16_mfma(const f32x4 (&av)[2][C16], int c, const float* __restrict__ brow, int h,
                                             f32x4& acc0, f32x4& acc1) {
  f32x4 bv[C16];
#pragma unroll
  for (int u = 0; u < C16; ++u) bv[u] = *(const f32x4*)(brow + 16 * (c * C16 + u) + 4 * h);
#pragma unroll
  for (int u = 0; u < C16; ++u) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][u].x, bv[u].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][u].x, bv[u].x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][u].y, bv[u].y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][u].y, bv[u].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][u].z, bv[u].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][u].z, bv[u].z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][u].w, bv[u].w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][u].w, bv[u].w, acc1, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256, 2) void k_dense_chain16(ChainArgs a, int ld0, int ld1) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* buf0 = smem;               // [16][ld0]
  float* buf1 = smem + 16 * ld0;    // [16][ld1]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 4, el = lane & 15;

  if (a.zero_ptr) {
    const int64_t n4 = a.zero_count / 4;
    f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int64_t s = blockIdx.x * 256 + threadIdx.x; s < n4; s += (int64_t)gridDim.x * 256) ((f32x4*)a.zero_ptr)[s] = z4;
    for (int64_t s = 4 * n4 + blockIdx.x * 256 + threadIdx.x; s < a.zero_count; s += (int64_t)gridDim.x * 256) a.zero_ptr[s] = 0.f;
  }

  CH_STAMP(0);
  const int64_t ntiles = (a.M + 15) / 16;
  for (int64_t mt = blockIdx.x; mt < ntiles; mt += gridDim.x) {
    const int64_t m0 = mt * 16;
    f32x4 a0[2][C16], a1[2][C16];
    {
      int l2 = 0, t2 = wv;
      bool have = wv < a.L[0].NW / 32;
      if (!have) have = chain_next_tile(a, 0, 1 << 20, wv, l2, t2);
      if (have) chain16_load_a(a0, a.L[l2].w, a.L[l2].KC / 8, t2, 0, el, h);
    }
    {  // stage the input tile [16][KC0] (KC0 <= 384 => at most 6 pieces per thread)
      const int KC0 = a.L[0].KC;
      const int q4 = KC0 / 4;
      const int total = 16 * q4;
      f32x4 v[6], pz[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int s = threadIdx.x + 256 * i;
        if (s < total) {
          const int row = s / q4, c4 = s - row * q4;
          int64_t mm = m0 + row;
          if (mm >= a.M) mm = a.M - 1;
          v[i] = *(const f32x4*)(a.in + mm * KC0 + 4 * c4);
          if (a.in_pre) pz[i] = *(const f32x4*)(a.in_pre + mm * KC0 + 4 * c4);
        }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int s = threadIdx.x + 256 * i;
        if (s < total) {
          const int row = s / q4, c4 = s - row * q4;
          f32x4 x = v[i];
          if (a.in_pre) {
            x.x *= chain_act_grad(a.in_act, pz[i].x); x.y *= chain_act_grad(a.in_act, pz[i].y);
            x.z *= chain_act_grad(a.in_act, pz[i].z); x.w *= chain_act_grad(a.in_act, pz[i].w);
          }
          *(f32x4*)(buf0 + row * ld0 + 4 * c4) = x;
        }
      }
    }
    __syncthreads();
    CH_STAMP(1);
    const int64_t m = m0 + el;
    const bool valid = m < a.M;
    for (int l = 0; l < a.n_layers; ++l) {
      const ChainLayerDev& L = a.L[l];
      const bool last = (l == a.n_layers - 1);
      const float* brow = ((l & 1) ? buf1 + el * ld1 : buf0 + el * ld0);
      float* nxt = (l & 1) ? buf0 : buf1;
      const int ldn = (l & 1) ? ld0 : ld1;
      const int nch = L.KC / (16 * C16);   // even (KC % 128 == 0)
      const int pcount = L.NW / 32;        // pairs of 16-feature tiles
      for (int p = wv; p < pcount; p += 4) {
        f32x4 rv[2], pv[2];
        const bool has_res = L.res && valid, has_post = (!last) && L.post_pre && valid;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int64_t off = m * L.NW + 32 * p + 16 * k + 4 * h;
          rv[k] = has_res ? *(const f32x4*)(L.res + off) : f32x4{0.f, 0.f, 0.f, 0.f};
          pv[k] = has_post ? *(const f32x4*)(L.post_pre + off) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 acc0, acc1;
        if (L.b) { acc0 = *(const f32x4*)(L.b + 32 * p + 4 * h); acc1 = *(const f32x4*)(L.b + 32 * p + 16 + 4 * h); }
        else { acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0; }
        for (int c = 0; c < nch; c += 2) {
          chain16_load_a(a1, L.w, L.KC / 8, p, c + 1, el, h);
          chain16_mfma(a0, c, brow, h, acc0, acc1);
          if (c + 2 < nch) {
            chain16_load_a(a0, L.w, L.KC / 8, p, c + 2, el, h);
          } else {
            int l2, t2;
            if (chain_next_tile(a, l, p, wv, l2, t2)) chain16_load_a(a0, a.L[l2].w, a.L[l2].KC / 8, t2, 0, el, h);
          }
          chain16_mfma(a1, c + 1, brow, h, acc0, acc1);
        }
        if (p == wv) CH_STAMP(2 + 3 * l);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int col = 32 * p + 16 * k + 4 * h;
          const int64_t off = m * L.NW + col;
          f32x4 o = k ? acc1 : acc0;
          if (L.pre_out && valid) *(f32x4*)(L.pre_out + off) = o;
          if (L.act != SPK_ACT_NONE) {
            o.x = chain_act(L.act, o.x); o.y = chain_act(L.act, o.y); o.z = chain_act(L.act, o.z); o.w = chain_act(L.act, o.w);
          }
          if (has_res) o += rv[k];
          if (L.out && valid) *(f32x4*)(L.out + off) = o;
          if (!last) {
            if (has_post) {
              const f32x4 pp = pv[k];
              o.x *= chain_act_grad(L.post_act, pp.x); o.y *= chain_act_grad(L.post_act, pp.y);
              o.z *= chain_act_grad(L.post_act, pp.z); o.w *= chain_act_grad(L.post_act, pp.w);
            }
            *(f32x4*)(nxt + el * ldn + col) = o;
          }
        }
      }
      CH_STAMP(3 + 3 * l);
      __syncthreads();
      CH_STAMP(4 + 3 * l);
    }
  }
  CH_STAMP(12);
}

// ------------------------------------------------------------------------------------------
// Split-precision form of the 32-row kernel (round 6; spk_split.h): the same stream of weight chunks read from the SPLIT packed image (same chunk
// geometry: chunk 2 s / 2 s + 1 of a 64-k block = fp16 high / low parts of k-step s), the activations of the tile kept in LDS as a high and a low
// fp16 image [32][K + 8] each -- split ONCE by whoever writes them (the staging of the input tile, the epilogue of a layer), read as the B operands
// of v_mfma_f32_32x32x16_f16 as they lie.  12 matrix instructions of 32 cycles per 64-k chunk and tile instead of 32 of 64 cycles.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void chain_sp_store4(_Float16* __restrict__ bh, _Float16* __restrict__ bl, f32x4 x) {
  h16x4 h, l;
  sp_split4(x, h, l);
  *(h16x4*)bh = h; *(h16x4*)bl = l;
}
__device__ __forceinline__ void chain_mfma_sp(const f32x4 (&av)[CCH], int c, const _Float16* __restrict__ bh, const _Float16* __restrict__ bl, int hi,
                                              f32x16& acc, f32x16& cross) {
  h16x8 vh[4], vl[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    vh[s] = *(const h16x8*)(bh + 64 * c + 16 * s + 8 * hi);
    vl[s] = *(const h16x8*)(bl + 64 * c + 16 * s + 8 * hi);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const h16x8 ah = __builtin_bit_cast(h16x8, av[2 * s]), al = __builtin_bit_cast(h16x8, av[2 * s + 1]);
    SP_STEP(ah, al, vh[s], vl[s], acc, cross);
  }
}

__global__ __launch_bounds__(256, 2) void k_dense_chain_sp(ChainArgs a, int ld0, int ld1) {
  // ld0 / ld1: row strides of the two activation buffers in HALVES (K + 8); each buffer = high image [32][ld] then low image [32][ld]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* buf0 = (_Float16*)smem;
  _Float16* buf1 = buf0 + 2 * 32 * ld0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;

  if (a.zero_ptr) {
    const int64_t n4 = a.zero_count / 4;
    f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int64_t s = blockIdx.x * 256 + threadIdx.x; s < n4; s += (int64_t)gridDim.x * 256) ((f32x4*)a.zero_ptr)[s] = z4;
    for (int64_t s = 4 * n4 + blockIdx.x * 256 + threadIdx.x; s < a.zero_count; s += (int64_t)gridDim.x * 256) a.zero_ptr[s] = 0.f;
  }

  const int64_t ntiles = (a.M + 31) / 32;
  for (int64_t mt = blockIdx.x; mt < ntiles; mt += gridDim.x) {
    const int64_t m0 = mt * 32;
    f32x4 a0[CCH], a1[CCH];
    {
      int l2 = 0, t2 = wv;
      bool have = wv < a.L[0].NW / 32;
      if (!have) have = chain_next_tile(a, 0, 1 << 20, wv, l2, t2);
      if (have) chain_load_a(a0, a.L[l2].w, a.L[l2].KC / 8, t2, 0, lane);
    }
    {
      const int KC0 = a.L[0].KC;
      const int q4 = KC0 / 4;
      const int total = 32 * q4;
      for (int i0 = 0; i0 < 12; i0 += 6) {
        if (threadIdx.x + 256 * i0 >= total) break;
        f32x4 v[6], pz[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int s = threadIdx.x + 256 * (i0 + i);
          if (s < total) {
            const int row = s / q4, c4 = s - row * q4;
            int64_t mm = m0 + row;
            if (mm >= a.M) mm = a.M - 1;
            v[i] = *(const f32x4*)(a.in + mm * KC0 + 4 * c4);
            if (a.in_pre) pz[i] = *(const f32x4*)(a.in_pre + mm * KC0 + 4 * c4);
          }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int s = threadIdx.x + 256 * (i0 + i);
          if (s < total) {
            const int row = s / q4, c4 = s - row * q4;
            f32x4 x = v[i];
            if (a.in_pre) {
              x.x *= chain_act_grad(a.in_act, pz[i].x); x.y *= chain_act_grad(a.in_act, pz[i].y);
              x.z *= chain_act_grad(a.in_act, pz[i].z); x.w *= chain_act_grad(a.in_act, pz[i].w);
            }
            chain_sp_store4(buf0 + row * ld0 + 4 * c4, buf0 + (32 + row) * ld0 + 4 * c4, x);
          }
        }
      }
    }
    __syncthreads();
    const int64_t m = m0 + el;
    const bool valid = m < a.M;
#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
      const ChainLayerDev& L = a.L[l];
      const bool last = (l == a.n_layers - 1);
      const _Float16* bh = (l & 1) ? buf1 + el * ld1 : buf0 + el * ld0;
      const _Float16* bl = (l & 1) ? buf1 + (32 + el) * ld1 : buf0 + (32 + el) * ld0;
      _Float16* nxt = (l & 1) ? buf0 : buf1;
      const int ldn = (l & 1) ? ld0 : ld1;
      const int nch = L.KC / (8 * CCH);   // even
      const int tcount = L.NW / 32;
#pragma unroll 1
      for (int t = wv; t < tcount; t += 4) {
        f32x4 rv[4], pv[4];
        const bool has_res = L.res && valid, has_post = (!last) && L.post_pre && valid;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t off = m * L.NW + 32 * t + 8 * q + 4 * hi;
          rv[q] = has_res ? *(const f32x4*)(L.res + off) : f32x4{0.f, 0.f, 0.f, 0.f};
          pv[q] = has_post ? *(const f32x4*)(L.post_pre + off) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x16 acc, cross;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = L.b ? L.b[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi] : 0.f; cross[r] = 0.f; }
        for (int c = 0; c < nch; c += 2) {
          chain_load_a(a1, L.w, L.KC / 8, t, c + 1, lane);
          chain_mfma_sp(a0, c, bh, bl, hi, acc, cross);
          if (c + 2 < nch) {
            chain_load_a(a0, L.w, L.KC / 8, t, c + 2, lane);
          } else {
            int l2, t2;
            if (chain_next_tile(a, l, t, wv, l2, t2)) chain_load_a(a0, a.L[l2].w, a.L[l2].KC / 8, t2, 0, lane);
          }
          chain_mfma_sp(a1, c + 1, bh, bl, hi, acc, cross);
        }
        SP_FOLD(acc, cross);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * t + 8 * q + 4 * hi;
          const int64_t off = m * L.NW + col;
          f32x4 o;
          o.x = acc[4 * q]; o.y = acc[4 * q + 1]; o.z = acc[4 * q + 2]; o.w = acc[4 * q + 3];
          if (L.pre_out && valid) *(f32x4*)(L.pre_out + off) = o;
          if (L.act != SPK_ACT_NONE) {
            o.x = chain_act(L.act, o.x); o.y = chain_act(L.act, o.y); o.z = chain_act(L.act, o.z); o.w = chain_act(L.act, o.w);
          }
          if (has_res) o += rv[q];
          if (L.out && valid) *(f32x4*)(L.out + off) = o;
          if (!last) {
            if (has_post) {
              const f32x4 p = pv[q];
              o.x *= chain_act_grad(L.post_act, p.x); o.y *= chain_act_grad(L.post_act, p.y);
              o.z *= chain_act_grad(L.post_act, p.z); o.w *= chain_act_grad(L.post_act, p.w);
            }
            chain_sp_store4(nxt + el * ldn + col, nxt + (32 + el) * ldn + col, o);
          }
        }
      }
      __syncthreads();
    }
  }
}

// Split images of the chain that is about to be launched: spk_apply_pack (spk_pack.h) notes, per layer, the split image that belongs to the packed fp32
// image it put into the chain; spk_dense_chain_f32 -- called next on the same host thread -- takes them only if every layer's `w` is exactly the noted
// packed pointer, and forgets them on the way out.  (Not a global pointer map: a chain handed to the C ABI with its own packed buffer must never meet the
// split image of a freed model whose buffer address it happens to reuse.)
struct ChainSplitNote { const float* packed[CH_MAXL]; const float* split[CH_MAXL]; int n; };
static thread_local ChainSplitNote g_split_note = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, 0};
void spk_note_split_images(const float* const* packed, const float* const* split, int n) {
  g_split_note.n = n > CH_MAXL ? 0 : n;
  for (int l = 0; l < g_split_note.n; ++l) { g_split_note.packed[l] = packed[l]; g_split_note.split[l] = split[l]; }
}
struct ChainSplitNoteClear { ~ChainSplitNoteClear() { g_split_note.n = 0; } };

// ---- packed weight image ------------------------------------------------------------------
// w is a Linear weight [n_out, k_in].  transposed == 0: the layer y = x W^T (A[i][kk] = W[i][kk], contraction
// over k_in); transposed == 1: the input-gradient layer gx = gy W (A[i][kk] = W[kk][i], contraction over n_out).
__global__ void k_pack_weight(const float* __restrict__ w, int n_out, int k_in, int transposed, float* __restrict__ P) {
  const int KC = transposed ? n_out : k_in, NW = transposed ? k_in : n_out;
  const int KB = KC / 8;
  const int64_t total = (int64_t)KC * NW;
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < total; s += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(s & 3), lane = (int)((s >> 2) & 63);
    const int64_t blk = s >> 8;
    const int ug = (int)(blk % KB), t = (int)(blk / KB);
    const int i = 32 * t + (lane & 31), kk = 8 * ug + 4 * (lane >> 5) + v;
    P[s] = transposed ? w[(int64_t)kk * k_in + i] : w[(int64_t)i * k_in + kk];
  }
}

// Split-precision image of the same matrix (spk_split.h): the chunk geometry of the fp32 image -- tile t of 32 rows, KB = KC / 8 chunks of
// 1024 bytes, a lane's 16 bytes at ((t KB + ug) 64 + lane) 16 -- with chunk ug = 2 s + part holding, for k-step s (16 contraction
// indices), the eight fp16 HIGH parts (part 0) or the eight 2^11-scaled fp16 LOW parts (part 1) of A[32 t + (lane & 31)][16 s + 8 (lane >> 5) + e],
// e = 0..7: the A operand of v_mfma_f32_32x32x16_f16 as it lies.  A kernel that walks the fp32 image in blocks of 8 chunks (64 k) walks
// this one with the same offsets.
__global__ void k_pack_weight_split(const float* __restrict__ w, int n_out, int k_in, int transposed, float* __restrict__ P) {
  const int KC = transposed ? n_out : k_in, NW = transposed ? k_in : n_out;
  const int KB = KC / 8;
  const int64_t slots = (int64_t)(NW / 32) * (KC / 16) * 64;
  for (int64_t sl = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; sl < slots; sl += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(sl & 63);
    const int64_t blk = sl >> 6;
    const int s = (int)(blk % (KC / 16)), t = (int)(blk / (KC / 16));
    const int i = 32 * t + (lane & 31);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = 16 * s + 8 * (lane >> 5) + e;
      x[e] = transposed ? w[(int64_t)kk * k_in + i] : w[(int64_t)i * k_in + kk];
    }
    h16x8 h, l;
    sp_split8(x, h, l);
    h16x8* dst = (h16x8*)P;
    dst[((int64_t)t * KB + 2 * s) * 64 + lane] = h;
    dst[((int64_t)t * KB + 2 * s + 1) * 64 + lane] = l;
  }
}
int spk_pack_weight_split_internal(const float* w, int n_out, int k_in, int transposed, float* packed, hipStream_t stream) {
  const int KC = transposed ? n_out : k_in, NW = transposed ? k_in : n_out;
  SPK_CHECK_ARG(w && packed && n_out > 0 && k_in > 0, "spk_pack_weight_split: bad input");
  SPK_CHECK_ARG(KC % 16 == 0 && NW % 32 == 0, "spk_pack_weight_split: contraction length %d must be a multiple of 16, output width %d of 32", KC, NW);
  hipLaunchKernelGGL(k_pack_weight_split, dim3(spk_grid_for((int64_t)KC * NW / 8, 256, spk_num_cus() * 8)), dim3(256), 0, stream, w, n_out, k_in, transposed, packed);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

int spk_pack_weight_internal(const float* w, int n_out, int k_in, int transposed, float* packed, hipStream_t stream);
extern "C" int spk_pack_weight_f32(const float* w, int32_t n_out, int32_t k_in, int32_t transposed, float* packed, void* stream_) {
  return spk_pack_weight_internal(w, n_out, k_in, transposed, packed, (hipStream_t)stream_);
}
int spk_pack_weight_internal(const float* w, int n_out, int k_in, int transposed, float* packed, hipStream_t stream) {
  const int KC = transposed ? n_out : k_in, NW = transposed ? k_in : n_out;
  SPK_CHECK_ARG(w && packed && n_out > 0 && k_in > 0, "spk_pack_weight_f32: bad input");
  SPK_CHECK_ARG(KC % 8 == 0 && NW % 32 == 0, "spk_pack_weight_f32: contraction length %d must be a multiple of 8, output width %d of 32", KC, NW);
  hipLaunchKernelGGL(k_pack_weight, dim3(spk_grid_for((int64_t)KC * NW, 256, spk_num_cus() * 8)), dim3(256), 0, stream, w, n_out, k_in, transposed, packed);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

static long long* g_chain_dbg = nullptr;
extern "C" void spk_chain_set_debug_buffer(void* p) { g_chain_dbg = (long long*)p; }
static int g_chain_rows = 0;   // 0: by size, 16 / 32: forced (tests, tuning)
extern "C" void spk_chain_set_rows(int rows) { g_chain_rows = (rows == 16 || rows == 32) ? rows : 0; }

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int spk_dense_internal(const float* in, const float* pre_in, const float* w, const float* b,
                       const float* res, float* out, float* pre_out, int64_t M, int KC, int NW,
                       int act, bool trans, int pro, hipStream_t stream);

static bool al16(const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; }

// true if the fused kernel can run this chain
static bool chain_supported(const spk_chain_t* c) {
  if (c->n_layers < 1 || c->n_layers > CH_MAXL) return false;
  if (!al16(c->in) || !al16(c->in_pre) || !al16(c->zero_ptr)) return false;
  int kc = c->layers[0].k;
  for (int l = 0; l < c->n_layers; ++l) {
    const spk_chain_layer_t& L = c->layers[l];
    if (L.k != kc) return false;
    if (L.trans != 2) return false;  // the fused kernels read packed weights only (see chain_load_a)
    if (L.k % 128 != 0 || L.n_out % 32 != 0 || L.k > CH_MAXW || L.n_out > CH_MAXW) return false;
    if (!al16(L.w) || !al16(L.b) || !al16(L.res) || !al16(L.out) || !al16(L.pre_out) || !al16(L.post_pre)) return false;
    kc = L.n_out;
  }
  return true;
}

extern "C" int spk_dense_chain_f32(const spk_chain_t* c, void* stream_) {
  ChainSplitNoteClear forget_note_on_exit;
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_dense_chain_f32";
  SPK_CHECK_ARG(c != nullptr && c->n_layers >= 1 && c->n_layers <= CH_MAXL, "%s: 1..%d layers", who, CH_MAXL);
  SPK_CHECK_ARG(c->m >= 0, "%s: negative row count", who);
  if (c->m == 0 && !(c->zero_ptr && c->zero_count > 0)) return SPK_OK;
  const int variant = spk_get_variant();
  if (c->m > 0 && chain_supported(c) && variant != SPK_VARIANT_SIMPLE) {
    ChainArgs a = {};
    int maxw = c->layers[0].k;
    for (int l = 0; l < c->n_layers; ++l) {
      const spk_chain_layer_t& S = c->layers[l];
      SPK_CHECK_ARG(S.w != nullptr, "%s: null weight in layer %d", who, l);
      ChainLayerDev& D = a.L[l];
      D.w = S.w; D.b = S.b; D.res = S.res; D.out = S.out; D.pre_out = S.pre_out; D.post_pre = S.post_pre;
      D.KC = S.k; D.NW = S.n_out; D.act = S.act; D.trans = 2; D.post_act = S.post_act;
      if (S.n_out > maxw) maxw = S.n_out;
    }
    SPK_CHECK_ARG(c->in != nullptr, "%s: null input", who);
    a.n_layers = c->n_layers; a.in = c->in; a.in_pre = c->in_pre; a.in_act = c->in_act; a.M = c->m;
    a.zero_ptr = c->zero_ptr; a.zero_count = c->zero_ptr ? c->zero_count : 0;
    a.dbg = g_chain_dbg;
    (void)maxw;
    // buf0 holds the input tile and (3 layers) the output of layer 1; buf1 the output of layer 0
    int w0 = c->layers[0].k, w1 = c->n_layers > 1 ? c->layers[0].n_out : 0;
    if (c->n_layers > 2 && c->layers[1].n_out > w0) w0 = c->layers[1].n_out;
    const int ld0 = w0 + 4, ld1 = w1 + 4;
    static SpkPerDevice attr_done;
    int attr_done_dev;
    if (attr_done.pending(&attr_done_dev)) {
      const int max_lds = (int)(2 * 32 * (CH_MAXW + 4) * sizeof(float));
      SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_dense_chain, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
      SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_dense_chain16, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
      attr_done.mark(attr_done_dev);
    }
    // 16-row tiles while 32-row tiles would leave CUs idle or alone with one workgroup
    const int64_t ntiles32 = (c->m + 31) / 32;
    const bool rows16 = g_chain_rows ? (g_chain_rows == 16) : (ntiles32 < 2 * (int64_t)spk_num_cus());
    SpkProfScope prof(c->n_layers == 1 ? "chain1" : (c->n_layers == 2 ? "chain2" : "chain3"), stream);
    bool split = !rows16 && spk_get_split() != 0 && g_split_note.n == c->n_layers;
    const float* wsp[CH_MAXL] = {nullptr, nullptr, nullptr};
    for (int l = 0; l < c->n_layers && split; ++l) {
      wsp[l] = (c->layers[l].k % 64 == 0 && g_split_note.packed[l] == c->layers[l].w) ? g_split_note.split[l] : nullptr;
      if (!wsp[l]) split = false;
    }
    if (split) {
      // split-precision form (round 6): same launch geometry, activations as fp16 (high, low) images in LDS
      for (int l = 0; l < c->n_layers; ++l) a.L[l].w = wsp[l];
      const int ldh0 = w0 + 8, ldh1 = w1 + 8;
      static SpkPerDevice attr_sp;
      int attr_sp_dev;
      if (attr_sp.pending(&attr_sp_dev)) {
        SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_dense_chain_sp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 2 * 32 * (CH_MAXW + 8) * sizeof(_Float16))));
        attr_sp.mark(attr_sp_dev);
      }
      const int grid = (int)(ntiles32 < 4096 ? ntiles32 : 4096);
      const size_t lds = (size_t)2 * 32 * (ldh0 + ldh1) * sizeof(_Float16);
      hipLaunchKernelGGL(k_dense_chain_sp, dim3(grid), dim3(256), lds, stream, a, ldh0, ldh1);
    } else if (rows16) {
      const int64_t ntiles = (c->m + 15) / 16;
      const int grid = (int)(ntiles < 8192 ? ntiles : 8192);
      const size_t lds = (size_t)16 * (ld0 + ld1) * sizeof(float);
      hipLaunchKernelGGL(k_dense_chain16, dim3(grid), dim3(256), lds, stream, a, ld0, ld1);
    } else {
      const int grid = (int)(ntiles32 < 4096 ? ntiles32 : 4096);
      const size_t lds = (size_t)32 * (ld0 + ld1) * sizeof(float);
      hipLaunchKernelGGL(k_dense_chain, dim3(grid), dim3(256), lds, stream, a, ld0, ld1);
    }
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  // general path: layer by layer on the single-layer kernels (any shape)
  for (int l = 0; l < c->n_layers; ++l)
    SPK_CHECK_ARG(c->layers[l].trans != 2, "%s: packed weights (trans == 2) need k %% 128 == 0, n_out %% 32 == 0, widths <= %d, 16-byte aligned buffers and a variant other than 'simple'", who, CH_MAXW);
  if (c->zero_ptr && c->zero_count > 0) { int _zr = spk_zero_async(c->zero_ptr, (size_t)c->zero_count * sizeof(float), stream); if (_zr) return _zr; }
  if (c->m == 0) return SPK_OK;
  SPK_CHECK_ARG(c->in != nullptr, "%s: null input", who);
  const float* cur = c->in;
  const float* cur_pre = c->in_pre;
  int cur_act = c->in_pre ? c->in_act : SPK_ACT_NONE;
  for (int l = 0; l < c->n_layers; ++l) {
    const spk_chain_layer_t& S = c->layers[l];
    const bool last = (l == c->n_layers - 1);
    float* dst = S.out;
    if (!dst) {
      SPK_CHECK_ARG(!last && c->tmp[l & 1] != nullptr, "%s: layer %d needs an output or a temporary buffer", who, l);
      dst = c->tmp[l & 1];
    }
    int rc = spk_dense_internal(cur, cur_pre, S.w, S.b, S.res, dst, S.pre_out, c->m, S.k, S.n_out, S.act, S.trans != 0, cur_act, stream);
    if (rc) return rc;
    cur = dst;
    cur_pre = S.post_pre;
    cur_act = S.post_pre ? S.post_act : SPK_ACT_NONE;
  }
  return SPK_OK;
}
