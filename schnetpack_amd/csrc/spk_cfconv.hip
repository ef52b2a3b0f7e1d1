// Fused SchNet continuous-filter convolution (representation/schnet.py:60-67) for gfx950.
//
//   d_e = |r_e|;  phi_k(d_e) (nn/radial.py);  fc = cosine_cutoff(d_e) (nn/cutoff.py)
//   a_e = W1 phi_e + b1;  z_e = ssp(a_e);  g_e = W2 z_e + b2;  W_e = g_e * fc
//   y_i = sum_{e -> i} h_{j(e)} * W_e
//
// MFMA kernel: one wavefront owns a tile of 32 consecutive edges (lanes 0..31 and 32..63 both map
// to edge l & 31).  The filter MLP runs as two chained T-GEMMs (see spk_dense.hip) with the
// packed filter weights staged once per workgroup in LDS; the [32 edge x nf] filter tile lives
// only in registers.  Modulation gathers h[j] as 16-byte pieces, the per-centre-atom reduction goes
// through a small per-wave LDS transposition buffer, and each (segment, feature) is flushed with
// one float atomic (idx_i sorted => few segments per tile).
//
// Backward (first order, what Forces needs): forward-mode derivative through the filter MLP
// (a' = W1 phi', z' = sigmoid(a) a', g' = W2 z'), so dW_e/dd is available per edge without
// storing anything of size E x nf:
//   gr_e += (sum_f gy_i h_j (g' fc + g fc')) r_e / d_e
//   gh_j  = sum_{e: j(e)=j} gy_{i(e)} W_e
// On a symmetric neighbour list W_e == W_{e'} for the reversed edge, so gh is again a row-local
// segmented reduction (gather gy of the neighbours); otherwise float atomics on gh[j].
#include "spk_common.h"
#include "spk_split.h"
#include "spk_filter_split.h"


struct CfArgs {
  const float* h;      // [N, NF]
  const float* gy;     // [N, NF]  (backward only)
  const float* rij;    // [E, 3]
  const int64_t* idx_i;
  const int64_t* idx_j;
  const float* w1;     // [NF, n_rbf]
  const float* b1;     // [NF]
  const float* w2;     // [NF, NF]
  const float* b2;     // [NF]
  float* y;            // fwd: [N, NF] (pre-zeroed);  bwd: gh [N, NF] (pre-zeroed)
  float* gr;           // bwd: [E, 3] accumulated (pair kernels: assigned when gr_assign != 0)
  int gr_assign;       // pair kernels write every edge exactly once: the first interaction of a backward can assign
  int skip_gh;         // saved-filter pair backward: dL/dh is not needed (first interaction of an eval-mode backward) => no transposed sum
  int64_t E;
  int64_t N;
  long long* dbg;       // optional: cycle stamps of wave 0 / workgroup 0 (kernel tuning aid)
  float* gsave;         // fwd: optional [n_tiles*32, NF] raw filter-MLP outputs g_e (before the cutoff)
  const float* gload;   // bwd: the same buffer written by the forward of this interaction (or null)
  const int32_t* half;  // pair kernel: canonical edge of every undirected pair
  const int32_t* rev;   // pair kernel: reversed edge
  int64_t n_half;
  const int32_t* n_half_dev;  // optional device count (<= n_half) of a per-call compacted pair list
  const int32_t* grp_atom0;  // mol kernel: [G+1] first atom of every group
  const int32_t* grp_pair0;  // [G+1] first entry of the group in `half`
  const int32_t* grp_tile0;  // [G+1] first (group-aligned) tile of the group
  int n_groups;
  int max_group_atoms;
  int xcd_walk;        // persistent tile loops: XCD-contiguous walk (spk_xcd_tile; set by the launchers when gridDim.x % 8 == 0 on large lists)
  RadialDev rb;
  const int32_t* rowptr;     // row-tile forward (round 6): CSR of the sorted list
  const int32_t* edge_pair;  //   position in `half` of the undirected pair of every edge
};

// ------------------------------------------------------------------------------------------
// simple kernels: one workgroup per edge, one thread per filter channel; any NF / n_rbf.
// ------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void k_cfconv_simple(CfArgs a, int NF, int sym) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sphi = sm;                    // n_rbf
  float* sdphi = sphi + a.rb.n_rbf;    // n_rbf
  float* sz = sdphi + a.rb.n_rbf;      // NF
  float* szp = sz + NF;                // NF
  float* sred = szp + NF;              // blockDim/64
  const int f = threadIdx.x;
  for (int64_t e = blockIdx.x; e < a.E; e += gridDim.x) {
    const int64_t i = a.idx_i[e], j = a.idx_j[e];
    const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    float fc, dfc;
    spk_cutoff_eval(a.rb.cutoff, d, fc, dfc);
    __syncthreads();
    if (f < a.rb.n_rbf) { float p, dp; spk_rbf_eval(a.rb, f, d, p, dp); sphi[f] = p; sdphi[f] = dp; }
    __syncthreads();
    float z = 0.f, zp = 0.f;
    if (f < NF) {
      float av = a.b1[f], ap = 0.f;
      for (int k = 0; k < a.rb.n_rbf; ++k) {
        float wv = a.w1[f * a.rb.n_rbf + k];
        av = fmaf(wv, sphi[k], av);
        ap = fmaf(wv, sdphi[k], ap);
      }
      float sg;
      spk_softplus_sigmoid(av, z, sg);
      z -= SPK_LN2_F;
      zp = sg * ap;
      sz[f] = z; szp[f] = zp;
    }
    __syncthreads();
    float contrib = 0.f;
    if (f < NF) {
      float g = a.b2[f], gp = 0.f;
      for (int k = 0; k < NF; ++k) {
        float wv = a.w2[(int64_t)f * NF + k];
        g = fmaf(wv, sz[k], g);
        gp = fmaf(wv, szp[k], gp);
      }
      const float W = g * fc;
      if (!BWD) {
        unsafeAtomicAdd(&a.y[i * NF + f], a.h[j * NF + f] * W);
      } else {
        const float Wp = gp * fc + g * dfc;
        const float gyi = a.gy[i * NF + f];
        contrib = gyi * a.h[j * NF + f] * Wp;
        // gh[j] += gy[i] * W_e   (general form; identical to the symmetric row-local form)
        unsafeAtomicAdd(&a.y[j * NF + f], gyi * W);
      }
    }
    if (BWD) {
      contrib = spk_wave_sum(contrib);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = contrib;
      __syncthreads();
      if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) tot += sred[wv];
        if (d > 0.f) {
          const float s = tot / d;
          a.gr[3 * e] += s * rx; a.gr[3 * e + 1] += s * ry; a.gr[3 * e + 2] += s * rz;
        }
      }
    }
  }
  (void)sym;
}

// ------------------------------------------------------------------------------------------
// MFMA kernels
// ------------------------------------------------------------------------------------------
// Packed LDS image of a weight matrix W[NOUT][K] (row-major, K padded with zeros to 8*KB):
//   P[((t * KB + ug) * 64 + lane) * 4 + v] = W[32 t + (lane & 31)][8 ug + 4 (lane >> 5) + v]
// so one ds_read_b128 per lane delivers the A operands of 4 consecutive k-steps, conflict-free.
// All global loads of a thread are issued before the first LDS store (latency paid once).
template <int NTHREADS, int SLOTS>
__device__ __forceinline__ void stage_packed(float* dst, const float* __restrict__ w, int K, int KB) {
  constexpr int PER = (SLOTS + NTHREADS - 1) / NTHREADS;
  f32x4 v[PER];
  const bool vec = (K & 3) == 0;
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int s = threadIdx.x + p * NTHREADS;
    v[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s < SLOTS) {
      const int lane = s & 63;
      const int ug = (s >> 6) % KB;
      const int t = (s >> 6) / KB;
      const int row = 32 * t + (lane & 31);
      const int k0 = 8 * ug + 4 * (lane >> 5);
      const float* src = w + (int64_t)row * K + k0;
      if (vec) {
        if (k0 < K) v[p] = *(const f32x4*)src;
      } else {
        if (k0 + 0 < K) v[p].x = src[0];
        if (k0 + 1 < K) v[p].y = src[1];
        if (k0 + 2 < K) v[p].z = src[2];
        if (k0 + 3 < K) v[p].w = src[3];
      }
    }
  }
#pragma unroll
  for (int p = 0; p < PER; ++p) {
    const int s = threadIdx.x + p * NTHREADS;
    if (s < SLOTS) *(f32x4*)(dst + (int64_t)s * 4) = v[p];
  }
}

#define SPK_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)

// Row stride (floats) of the per-wave transposition buffer: two 32-channel tiles + 4 pad floats
// => conflict-free ds_write_b128 (8-lane groups hit 8 distinct 4-bank groups) and ds_read_b32.
#define TP2 68

template <int NF, int KPB, int NWAVES, bool BWD, bool SYM>
__global__ __launch_bounds__(NWAVES * 64) void k_cfconv_mfma(CfArgs a) {
  constexpr int NT = NF / 32;   // feature tiles
  constexpr int KB2 = NF / 8;   // k-blocks (of 8) of the second GEMM
  constexpr bool USE_LDS = !BWD || SYM;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW2 = smem;                          // NF*NF
  float* sW1 = sW2 + NF * NF;                 // NF*KPB*8
  float* sb1 = sW1 + NF * KPB * 8;            // NF
  float* sb2 = sb1 + NF;                      // NF
  float* sT = sb2 + NF;                       // NWAVES * 32 * TP2
  int* sCnt = (int*)(sT + NWAVES * 32 * TP2); // 4 ints

  stage_packed<NWAVES * 64, NF * NF / 4>(sW2, a.w2, NF, KB2);
  stage_packed<NWAVES * 64, NF * KPB * 2>(sW1, a.w1, a.rb.n_rbf, KPB);
  for (int s = threadIdx.x; s < NF; s += NWAVES * 64) { sb1[s] = a.b1[s]; sb2[s] = a.b2[s]; }
  if (threadIdx.x == 0) sCnt[0] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  float* myT = sT + wv * (32 * TP2);
  const int64_t ntiles = (a.E + 31) / 32;

  // tiles of this workgroup: blockIdx.x + n * gridDim.x; waves take them from a shared counter so
  // that every CU gets the same number of tiles (+-1) and its SIMDs stay evenly loaded
  while (true) {
    int nidx = 0;
    if (lane == 0) nidx = atomicAdd(&sCnt[0], 1);
    nidx = __builtin_amdgcn_readfirstlane(nidx);
    const int64_t tile = a.xcd_walk ? spk_xcd_tile(nidx, ntiles) : (int64_t)blockIdx.x + (int64_t)nidx * gridDim.x;
    if (tile >= ntiles) break;

    const int64_t e = tile * 32 + el;
    const bool valid = e < a.E;
    const int64_t ec = valid ? e : (a.E - 1);
    const float rx = a.rij[3 * ec], ry = a.rij[3 * ec + 1], rz = a.rij[3 * ec + 2];
    const int64_t j = a.idx_j[ec];
    const int64_t i = a.idx_i[ec];
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    float fc, dfc;
    spk_cutoff_eval(a.rb.cutoff, d, fc, dfc);
    if (!valid) { fc = 0.f; dfc = 0.f; }
    // segment structure of the tile (wave-uniform): bit k set <=> the run of centre atom idx_i
    // ends at edge k.  Lanes 32..63 mirror lanes 0..31.
    const int ieff = valid ? (int)i : -1;
    const int inext = __shfl(ieff, (lane + 1) & 63, 64);
    const unsigned long long bal = __ballot(ieff != inext);
    const unsigned flushmask = __builtin_amdgcn_readfirstlane((unsigned)(bal & 0xffffffffull)) | 0x80000000u;

    // forward only: all neighbour rows of the tile are requested up front (64 VGPRs)
    f32x4 hjv[BWD ? 1 : NT][4];
    if (!BWD) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) hjv[BWD ? 0 : t][q] = *(const f32x4*)(a.h + j * NF + 32 * t + 8 * q + 4 * hi);
    }

    // ---- radial basis for this lane's k slots: kk = 8u + 4hi + v
    float phi[KPB][4], dphi[BWD ? KPB : 1][4];
#pragma unroll
    for (int u = 0; u < KPB; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float p, dp;
        spk_rbf_eval(a.rb, 8 * u + 4 * hi + v, d, p, dp);
        phi[u][v] = p;
        if (BWD) dphi[BWD ? u : 0][v] = dp;
      }

    // ---- GEMM 1 (transposed): a^T[f][e] = sum_k W1[f][k] phi[e][k] + b1[f];  z = ssp(a)
    f32x16 z[NT];
    f32x16 zp[BWD ? NT : 1];
    {
      f32x4 wq = *(const f32x4*)(sW1 + lane * 4);
#pragma unroll
      for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) z[c][r] = sb1[32 * c + (r & 3) + 8 * (r >> 2) + 4 * hi];
        if (BWD) {
#pragma unroll
          for (int r = 0; r < 16; ++r) zp[BWD ? c : 0][r] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < KPB; ++u) {
          const int nxt = c * KPB + u + 1;
          f32x4 wn = wq;
          if (nxt < NT * KPB) wn = *(const f32x4*)(sW1 + (nxt * 64 + lane) * 4);
          z[c] = SPK_MFMA(wq.x, phi[u][0], z[c]);
          z[c] = SPK_MFMA(wq.y, phi[u][1], z[c]);
          z[c] = SPK_MFMA(wq.z, phi[u][2], z[c]);
          z[c] = SPK_MFMA(wq.w, phi[u][3], z[c]);
          if (BWD) {
            f32x16& q = zp[BWD ? c : 0];
            q = SPK_MFMA(wq.x, dphi[BWD ? u : 0][0], q);
            q = SPK_MFMA(wq.y, dphi[BWD ? u : 0][1], q);
            q = SPK_MFMA(wq.z, dphi[BWD ? u : 0][2], q);
            q = SPK_MFMA(wq.w, dphi[BWD ? u : 0][3], q);
          }
          wq = wn;
        }
      }
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float sp, sg;
          spk_fast_softplus_sigmoid(z[c][r], sp, sg);
          z[c][r] = sp - SPK_LN2_F;
          if (BWD) zp[BWD ? c : 0][r] *= sg;
        }
    }

    float dsum = 0.f;  // BWD: sum_f gy_i h_j dW/dd over this lane's channels

    // ---- GEMM 2 per output tile t, modulation, segmented reduction per pair of tiles
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // backward: the three gathered rows of this tile are requested before the 128 MFMAs
      f32x4 hq[BWD ? 4 : 1], gyiq[BWD ? 4 : 1], gyjq[(BWD && SYM) ? 4 : 1];
      if (BWD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * t + 8 * q + 4 * hi;
          hq[BWD ? q : 0] = *(const f32x4*)(a.h + j * NF + col);
          gyiq[BWD ? q : 0] = *(const f32x4*)(a.gy + i * NF + col);
          if (SYM) gyjq[(BWD && SYM) ? q : 0] = *(const f32x4*)(a.gy + j * NF + col);
        }
      }
      f32x16 g, gp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { g[r] = sb2[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi]; gp[r] = 0.f; }
      {
        const float* wbase = sW2 + ((int64_t)t * KB2 * 64 + lane) * 4;
        f32x4 wq = *(const f32x4*)wbase;
#pragma unroll
        for (int ug = 0; ug < KB2; ++ug) {
          const int c = ug >> 2, q = ug & 3;
          f32x4 wn = wq;
          if (ug + 1 < KB2) wn = *(const f32x4*)(wbase + (ug + 1) * 256);
          g = SPK_MFMA(wq.x, z[c][4 * q + 0], g);
          g = SPK_MFMA(wq.y, z[c][4 * q + 1], g);
          g = SPK_MFMA(wq.z, z[c][4 * q + 2], g);
          g = SPK_MFMA(wq.w, z[c][4 * q + 3], g);
          if (BWD) {
            const f32x16& zq = zp[BWD ? c : 0];
            gp = SPK_MFMA(wq.x, zq[4 * q + 0], gp);
            gp = SPK_MFMA(wq.y, zq[4 * q + 1], gp);
            gp = SPK_MFMA(wq.z, zq[4 * q + 2], gp);
            gp = SPK_MFMA(wq.w, zq[4 * q + 3], gp);
          }
          wq = wn;
        }
      }
      // modulation; lane holds channels 32t + 8q + 4hi + v of edge el
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 p;
        if (!BWD) {
          const f32x4 hj = hjv[BWD ? 0 : t][q];
          p.x = g[4 * q + 0] * fc * hj.x; p.y = g[4 * q + 1] * fc * hj.y;
          p.z = g[4 * q + 2] * fc * hj.z; p.w = g[4 * q + 3] * fc * hj.w;
        } else {
          const f32x4 hj = hq[BWD ? q : 0];
          const f32x4 gyi = gyiq[BWD ? q : 0];
          const float W0 = g[4 * q + 0] * fc, W1v = g[4 * q + 1] * fc, W2v = g[4 * q + 2] * fc, W3 = g[4 * q + 3] * fc;
          dsum += gyi.x * hj.x * (gp[4 * q + 0] * fc + g[4 * q + 0] * dfc);
          dsum += gyi.y * hj.y * (gp[4 * q + 1] * fc + g[4 * q + 1] * dfc);
          dsum += gyi.z * hj.z * (gp[4 * q + 2] * fc + g[4 * q + 2] * dfc);
          dsum += gyi.w * hj.w * (gp[4 * q + 3] * fc + g[4 * q + 3] * dfc);
          if (SYM) {
            const f32x4 gyj = gyjq[(BWD && SYM) ? q : 0];
            p.x = W0 * gyj.x; p.y = W1v * gyj.y; p.z = W2v * gyj.z; p.w = W3 * gyj.w;
          } else if (valid && !a.skip_gh) {          // (skip_gh: the transposed sum comes from a row pass over the by-neighbour list)
            float* dst = a.y + j * NF + 32 * t + 8 * q + 4 * hi;
            unsafeAtomicAdd(dst + 0, W0 * gyi.x);
            unsafeAtomicAdd(dst + 1, W1v * gyi.y);
            unsafeAtomicAdd(dst + 2, W2v * gyi.z);
            unsafeAtomicAdd(dst + 3, W3 * gyi.w);
          }
        }
        if (USE_LDS) *(f32x4*)(myT + el * TP2 + 32 * (t & 1) + 8 * q + 4 * hi) = p;
      }
      // after an odd tile (or the last one): lanes 0..31 own the channels of tile t-1 (or t if it is
      // a lone tile), lanes 32..63 those of tile t; every lane scans the 32 edges of its column
      if (USE_LDS && ((t & 1) == 1 || t == NT - 1)) {
        const bool pair = (t & 1) == 1;
        const int t0 = pair ? t - 1 : t;
        spk_wave_lds_sync();
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = myT[k * TP2 + lane];
        const bool active = pair || hi == 0;
        float* ybase = a.y + 32 * t0 + lane;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          acc += v[k];
          if ((flushmask >> k) & 1u) {   // wave-uniform
            const int ci = __builtin_amdgcn_readlane(ieff, k);
            if (ci >= 0 && active) unsafeAtomicAdd(ybase + (int64_t)ci * NF, acc);
            acc = 0.f;
          }
        }
        spk_wave_lds_sync();
      }
    }
    if (BWD) {
      dsum += __shfl_xor(dsum, 32, 64);
      if (hi == 0 && valid && d > 0.f) {
        const float s = dsum / d;
        a.gr[3 * e] += s * rx; a.gr[3 * e + 1] += s * ry; a.gr[3 * e + 2] += s * rz;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Pair kernels: on a symmetric neighbour list the filter of an edge and of its reverse are identical
// (they depend on d only), so one tile holds 32 UNDIRECTED pairs (the canonical edge e < rev[e] of
// each) and the filter MLP -- the MFMA work -- is evaluated once per pair, i.e. half as often:
//   forward :  y[i] += h[j] * W_e            and   y[j] += h[i] * W_e
//   backward:  gh[i] += gy[j] * W_e          and   gh[j] += gy[i] * W_e
//              gr[e]  += (sum gy[i] h[j] W'_e) r_e / d   and   gr[rev e] -= (sum gy[j] h[i] W'_e) r_e / d
//
// GEMM 2 runs with SWAPPED operands (A = hidden activations, lane = pair; B = packed W2): the output
// tile then has rows = pairs and columns = channels, i.e. every lane owns one channel and its 16
// accumulator registers are 16 pairs of the tile.  Consequences: the neighbour rows are gathered as
// coalesced 128-byte row segments (lanes = consecutive channels), the per-centre-atom sum is a
// per-lane running sum over registers (idx_i is sorted inside the half list: flush at run ends), no
// LDS transposition is needed, and every flush is a coalesced float atomic.
//
// GS (backward): the raw filter outputs g_e were saved by the forward kernel, so only the derivative
// GEMM (g') runs: 96 + 256 MFMAs per tile instead of 96 + 512.
// MOL: the list is block diagonal with small blocks (a batch of molecules): a workgroup owns whole
// groups of atoms, accumulates their rows in LDS (conflict-free ds_add_f32: lanes = channels) and writes
// them once -- no global atomics, no memset.  Tiles are aligned to the groups.
// ------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) EdgeRec { int i; int j; float fc; float dfc; };

// register r of the half hi holds pair (r & 3) + 8 (r >> 2) + 4 hi of the tile
__device__ __forceinline__ int pair_of(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int NF, int KPB, int NWAVES, bool BWD, bool GS, bool MOL>
__global__ __launch_bounds__(NWAVES * 64) void k_cfconv_pair(CfArgs a) {
  constexpr int NT = NF / 32;
  constexpr int KB2 = NF / 8;
#define SPK_STAMP(n) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[n] = (long long)__builtin_readcyclecounter(); } while (0)
  SPK_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW2 = smem;                                   // NF*NF
  float* sW1 = sW2 + NF * NF;                          // NF*KPB*8
  float* sb1 = sW1 + NF * KPB * 8;                     // NF
  float* sb2 = sb1 + NF;                               // NF
  EdgeRec* sE = (EdgeRec*)(sb2 + NF);                  // NWAVES * 32 records
  constexpr int DS = BWD ? 2 * 32 * 33 : 0;            // backward: per-wave [2][32 pairs][32 lanes + 1 pad] partial sums
  float* sD = (float*)(sE + NWAVES * 32);              // NWAVES * DS floats
  float* sY = sD + NWAVES * DS;                        // MOL: max_group_atoms * NF
  int* sCnt = (int*)(sY + (MOL ? (int64_t)a.max_group_atoms * NF : 0));

  stage_packed<NWAVES * 64, NF * NF / 4>(sW2, a.w2, NF, KB2);
  stage_packed<NWAVES * 64, NF * KPB * 2>(sW1, a.w1, a.rb.n_rbf, KPB);
  for (int s = threadIdx.x; s < NF; s += NWAVES * 64) { sb1[s] = a.b1[s]; sb2[s] = a.b2[s]; }
  if (!MOL && threadIdx.x == 0) sCnt[0] = 0;
  __syncthreads();
  SPK_STAMP(1);

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  EdgeRec* myE = sE + wv * 32;
  float* myD = sD + wv * DS;

  const int ngroups = MOL ? a.n_groups : 1;
  for (int grp = MOL ? (int)blockIdx.x : 0; grp < ngroups; grp += MOL ? (int)gridDim.x : 1) {
    int ga0 = 0, ga1 = 0, gp0 = 0, gp1 = (int)(a.n_half_dev ? (int64_t)a.n_half_dev[0] : a.n_half), gt0 = 0;
    if (MOL) {
      ga0 = a.grp_atom0[grp]; ga1 = a.grp_atom0[grp + 1];
      gp0 = a.grp_pair0[grp]; gp1 = a.grp_pair0[grp + 1];
      gt0 = a.grp_tile0[grp];
      for (int s2 = threadIdx.x; s2 < (ga1 - ga0) * NF; s2 += NWAVES * 64) sY[s2] = 0.f;
      if (threadIdx.x == 0) sCnt[0] = 0;
      __syncthreads();
    }
    const int gtiles = (gp1 - gp0 + 31) / 32;

    while (true) {
      int nidx = 0;
      if (lane == 0) nidx = atomicAdd(&sCnt[0], 1);
      nidx = __builtin_amdgcn_readfirstlane(nidx);
      // MOL: the group's tiles; otherwise tiles blockIdx.x + n * gridDim.x of the whole list
      const int ltile = MOL ? nidx : (a.xcd_walk ? (int)spk_xcd_tile(nidx, gtiles) : (int)blockIdx.x + nidx * (int)gridDim.x);
      if (ltile >= gtiles) break;
      const int64_t gtile = (int64_t)gt0 + ltile;        // addresses the saved filters

      // ---- per-pair geometry (lanes 32..63 mirror lanes 0..31)
      const int pfirst = gp0 + 32 * ltile;
      const int nvalid = (gp1 - pfirst) < 32 ? (gp1 - pfirst) : 32;
      const bool valid = el < nvalid;
      const int64_t e = a.half[pfirst + (valid ? el : (nvalid - 1))];
      const int64_t e2 = a.rev[e];
      const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
      const int i = (int)a.idx_i[e], j = (int)a.idx_j[e];
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      float fc, dfc;
      spk_cutoff_eval_fast(a.rb.cutoff, d, fc, dfc);
      if (!valid) { fc = 0.f; dfc = 0.f; }
      if (hi == 0) { EdgeRec er; er.i = i - ga0; er.j = j - ga0; er.fc = fc; er.dfc = dfc; myE[el] = er; }

      float phi[KPB][4], dphi[BWD ? KPB : 1][4];
#pragma unroll
      for (int u = 0; u < KPB; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float p, dp;
          spk_rbf_eval_fast(a.rb, 8 * u + 4 * hi + v, d, p, dp);
          phi[u][v] = p;
          if (BWD) dphi[BWD ? u : 0][v] = dp;
        }
      SPK_STAMP(2);

      // ---- GEMM 1 (rows = hidden channels, columns = pairs): a = W1 phi + b1; z = ssp(a); z' = sigma(a) W1 phi'
      f32x16 z[(BWD && GS) ? 1 : NT];
      f32x16 zp[BWD ? NT : 1];
      {
        f32x4 wq = *(const f32x4*)(sW1 + lane * 4);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
          f32x16 zc, zq;
#pragma unroll
          for (int r = 0; r < 16; ++r) { zc[r] = sb1[32 * c + pair_of(r, hi)]; zq[r] = 0.f; }
#pragma unroll
          for (int u = 0; u < KPB; ++u) {
            const int nxt = c * KPB + u + 1;
            f32x4 wn = wq;
            if (nxt < NT * KPB) wn = *(const f32x4*)(sW1 + (nxt * 64 + lane) * 4);
            zc = SPK_MFMA(wq.x, phi[u][0], zc);
            zc = SPK_MFMA(wq.y, phi[u][1], zc);
            zc = SPK_MFMA(wq.z, phi[u][2], zc);
            zc = SPK_MFMA(wq.w, phi[u][3], zc);
            if (BWD) {
              zq = SPK_MFMA(wq.x, dphi[BWD ? u : 0][0], zq);
              zq = SPK_MFMA(wq.y, dphi[BWD ? u : 0][1], zq);
              zq = SPK_MFMA(wq.z, dphi[BWD ? u : 0][2], zq);
              zq = SPK_MFMA(wq.w, dphi[BWD ? u : 0][3], zq);
            }
            wq = wn;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float sp, sg;
            spk_fast_softplus_sigmoid(zc[r], sp, sg);
            zc[r] = sp - SPK_LN2_F;
            if (BWD) zq[r] *= sg;
          }
          if (!(BWD && GS)) z[(BWD && GS) ? 0 : c] = zc;
          if (BWD) zp[BWD ? c : 0] = zq;
        }
      }
      spk_wave_lds_sync();   // myE visible to the whole wave
      // run structure of this lane's 16 pairs: bit r set <=> the centre atom changes after register r
      unsigned runmask = 0x8000u;
      {
        int prev = myE[pair_of(0, hi)].i;
#pragma unroll
        for (int r = 1; r < 16; ++r) {
          const int cur = myE[pair_of(r, hi)].i;
          if (cur != prev) runmask |= 1u << (r - 1);
          prev = cur;
        }
      }
      // backward: the per-pair sums over channels run over LANES in this layout; every lane keeps its
      // partial (pair, lane) in LDS (own slot, no atomics), the pair owners add the 32 partials at the end
      SPK_STAMP(3);

      // ---- per output tile t: gathers, GEMM 2 (rows = pairs, columns = channels), modulation, sums
#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        const int c0 = 32 * t + el;   // this lane's channel
        // forward / recomputing backward: all neighbour rows of the tile are requested before the MFMAs.
        // Saved-filter backward (2 waves/SIMD => 256 VGPRs): rows are requested after the MFMAs, 8 pairs
        // at a time, to bound the live registers.
        constexpr bool LATE = BWD && GS;
        constexpr int RG = LATE ? 8 : 16;   // pairs per gather group
        float hj[RG], hc[RG], gyi[BWD ? RG : 1], gyj[BWD ? RG : 1];
        if (!LATE) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const EdgeRec er = myE[pair_of(r, hi)];
            // 32-bit element offsets (n_atoms * nf < 2^31, checked by the launcher): one VGPR per address
            const unsigned oi = (unsigned)(er.i + ga0) * NF + c0, oj = (unsigned)(er.j + ga0) * NF + c0;
            hj[LATE ? 0 : r] = a.h[oj];
            hc[LATE ? 0 : r] = a.h[oi];
            if (BWD) { gyi[(BWD && !LATE) ? r : 0] = a.gy[oi]; gyj[(BWD && !LATE) ? r : 0] = a.gy[oj]; }
          }
        }
        f32x16 g, gp;
        const float bias2 = sb2[c0];
#pragma unroll
        for (int r = 0; r < 16; ++r) { g[r] = bias2; gp[r] = 0.f; }
        {
          const float* wbase = sW2 + ((int64_t)t * KB2 * 64 + lane) * 4;
          f32x4 wq = *(const f32x4*)wbase;
#pragma unroll
          for (int ug = 0; ug < KB2; ++ug) {
            const int c = ug >> 2, q = ug & 3;
            f32x4 wn = wq;
            if (ug + 1 < KB2) wn = *(const f32x4*)(wbase + (ug + 1) * 256);
            if (!GS) {
              const f32x16& zc = z[(BWD && GS) ? 0 : c];
              g = SPK_MFMA(zc[4 * q + 0], wq.x, g);
              g = SPK_MFMA(zc[4 * q + 1], wq.y, g);
              g = SPK_MFMA(zc[4 * q + 2], wq.z, g);
              g = SPK_MFMA(zc[4 * q + 3], wq.w, g);
            }
            if (BWD) {
              const f32x16& zq = zp[BWD ? c : 0];
              gp = SPK_MFMA(zq[4 * q + 0], wq.x, gp);
              gp = SPK_MFMA(zq[4 * q + 1], wq.y, gp);
              gp = SPK_MFMA(zq[4 * q + 2], wq.z, gp);
              gp = SPK_MFMA(zq[4 * q + 3], wq.w, gp);
            }
            wq = wn;
          }
        }
        SPK_STAMP(4 + 3 * t);
        if (!BWD && a.gsave) {
          float* gts = a.gsave + gtile * 32 * NF;   // this tile's filters, saved for the backward
#pragma unroll
          for (int r = 0; r < 16; ++r) gts[(unsigned)pair_of(r, hi) * NF + c0] = g[r];
        }
        // lane = channel c0: running sum over this lane's pairs for the centre atoms, one add per pair
        // for the neighbours
        float acc = 0.f;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += RG) {
          float gl[LATE ? RG : 1];
          if (LATE) {
            __builtin_amdgcn_sched_barrier(0);   // keep the second gather group behind the first group's math
            const float* gtp = a.gload + gtile * 32 * NF;
#pragma unroll
            for (int rr = 0; rr < RG; ++rr) {
              const EdgeRec er = myE[pair_of(r0 + rr, hi)];
              const unsigned oi = (unsigned)(er.i + ga0) * NF + c0, oj = (unsigned)(er.j + ga0) * NF + c0;
              hj[rr] = a.h[oj]; hc[rr] = a.h[oi];
              gyi[BWD ? rr : 0] = a.gy[oi]; gyj[BWD ? rr : 0] = a.gy[oj];
              gl[LATE ? rr : 0] = gtp[(unsigned)pair_of(r0 + rr, hi) * NF + c0];
            }
          }
#pragma unroll
          for (int rr = 0; rr < RG; ++rr) {
            const int r = r0 + rr;
            const EdgeRec er = myE[pair_of(r, hi)];
            const float gv = LATE ? gl[LATE ? rr : 0] : g[r];
            const float W = gv * er.fc;
            float toI, toJ;
            if (!BWD) {
              toI = W * hj[rr];
              toJ = W * hc[rr];
            } else {
              const float D = gp[r] * er.fc + gv * er.dfc;
              float* slot = myD + pair_of(r, hi) * 33 + el;
              const float p1 = gyi[BWD ? rr : 0] * hj[rr] * D, p2 = gyj[BWD ? rr : 0] * hc[rr] * D;
              if (t == 0) { slot[0] = p1; slot[32 * 33] = p2; }
              else { slot[0] += p1; slot[32 * 33] += p2; }
              toI = W * gyj[BWD ? rr : 0];
              toJ = W * gyi[BWD ? rr : 0];
            }
            acc += toI;
            if (MOL) {
              atomicAdd(sY + er.j * NF + c0, toJ);
              if ((runmask >> r) & 1u) { atomicAdd(sY + er.i * NF + c0, acc); acc = 0.f; }
            } else {
              unsafeAtomicAdd(a.y + ((unsigned)er.j * NF + c0), toJ);
              if ((runmask >> r) & 1u) { unsafeAtomicAdd(a.y + ((unsigned)er.i * NF + c0), acc); acc = 0.f; }
            }
          }
        }
        SPK_STAMP(6 + 3 * t);
      }
      if (BWD) {
        spk_wave_lds_sync();
        if (hi == 0 && valid) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int k2 = 0; k2 < 32; ++k2) { s1 += myD[el * 33 + k2]; s2 += myD[32 * 33 + el * 33 + k2]; }
          s1 = d > 0.f ? s1 / d : 0.f; s2 = d > 0.f ? s2 / d : 0.f;
          if (a.gr_assign) {
            a.gr[3 * e] = s1 * rx; a.gr[3 * e + 1] = s1 * ry; a.gr[3 * e + 2] = s1 * rz;
            a.gr[3 * e2] = -s2 * rx; a.gr[3 * e2 + 1] = -s2 * ry; a.gr[3 * e2 + 2] = -s2 * rz;
          } else {
            a.gr[3 * e] += s1 * rx; a.gr[3 * e + 1] += s1 * ry; a.gr[3 * e + 2] += s1 * rz;
            a.gr[3 * e2] -= s2 * rx; a.gr[3 * e2 + 1] -= s2 * ry; a.gr[3 * e2 + 2] -= s2 * rz;
          }
        }
      }
      spk_wave_lds_sync();   // myE / myD may be rewritten by the next tile
      SPK_STAMP(20);
    }
    if (MOL) {
      // the group's rows are complete: one coalesced store per row
      __syncthreads();
      for (int s2 = threadIdx.x; s2 < (ga1 - ga0) * NF; s2 += NWAVES * 64) a.y[(int64_t)ga0 * NF + s2] = sY[s2];
      __syncthreads();
    }
  }
  SPK_STAMP(21);
#undef SPK_STAMP
}

// Transposition variant of the pair kernel (GEMM 2 with rows = channels, columns = pairs; the products
// go through a per-wave LDS transposition buffer before the segmented flush).  Its per-pair sums over
// channels are per-LANE sums over registers, which saves ~30 VGPRs against the lane = channel layout:
// the saved-filter backward fits 2 waves/SIMD only in this form, so it is the one dispatched for it.
// SKIPGH (saved-filter backward only): dL/dh is not wanted -- no transposition, no transposed sum, no atomics
template <int NF, int KPB, int NWAVES, bool BWD, bool GS, bool SKIPGH = false>
__global__ __launch_bounds__(NWAVES * 64) void k_cfconv_pair_t(CfArgs a) {
  constexpr int NT = NF / 32;
  constexpr int KB2 = NF / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW2 = smem;
  float* sW1 = sW2 + NF * NF;
  float* sb1 = sW1 + NF * KPB * 8;
  float* sb2 = sb1 + NF;
  float* sT = sb2 + NF;
  int* sCnt = (int*)(sT + NWAVES * 32 * TP2);

  stage_packed<NWAVES * 64, NF * NF / 4>(sW2, a.w2, NF, KB2);
  stage_packed<NWAVES * 64, NF * KPB * 2>(sW1, a.w1, a.rb.n_rbf, KPB);
  for (int s = threadIdx.x; s < NF; s += NWAVES * 64) { sb1[s] = a.b1[s]; sb2[s] = a.b2[s]; }
  if (threadIdx.x == 0) sCnt[0] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  float* myT = sT + wv * (32 * TP2);
  const int64_t nhalf = a.n_half_dev ? (int64_t)a.n_half_dev[0] : a.n_half;
  const int64_t ntiles = (nhalf + 31) / 32;

  while (true) {
    int nidx = 0;
    if (lane == 0) nidx = atomicAdd(&sCnt[0], 1);
    nidx = __builtin_amdgcn_readfirstlane(nidx);
    const int64_t tile = a.xcd_walk ? spk_xcd_tile(nidx, ntiles) : (int64_t)blockIdx.x + (int64_t)nidx * gridDim.x;
    if (tile >= ntiles) break;

    const int64_t hidx = tile * 32 + el;
    const bool valid = hidx < nhalf;
    const int64_t e = a.half[valid ? hidx : (nhalf - 1)];
    const int64_t e2 = a.rev[e];
    const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
    const int64_t j = a.idx_j[e];
    const int64_t i = a.idx_i[e];
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    float fc, dfc;
    spk_cutoff_eval(a.rb.cutoff, d, fc, dfc);
    if (!valid) { fc = 0.f; dfc = 0.f; }
    const int ieff = valid ? (int)i : -1;
    const int jeff = valid ? (int)j : -1;
    const int inext = __shfl(ieff, (lane + 1) & 63, 64);
    const unsigned long long bal = __ballot(ieff != inext);
    const unsigned flushmask = __builtin_amdgcn_readfirstlane((unsigned)(bal & 0xffffffffull)) | 0x80000000u;

    float phi[KPB][4], dphi[BWD ? KPB : 1][4];
#pragma unroll
    for (int u = 0; u < KPB; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float p, dp;
        spk_rbf_eval(a.rb, 8 * u + 4 * hi + v, d, p, dp);
        phi[u][v] = p;
        if (BWD) dphi[BWD ? u : 0][v] = dp;
      }

    f32x16 z[NT];
    f32x16 zp[BWD ? NT : 1];
    {
      f32x4 wq = *(const f32x4*)(sW1 + lane * 4);
#pragma unroll
      for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) z[c][r] = sb1[32 * c + (r & 3) + 8 * (r >> 2) + 4 * hi];
        if (BWD) {
#pragma unroll
          for (int r = 0; r < 16; ++r) zp[BWD ? c : 0][r] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < KPB; ++u) {
          const int nxt = c * KPB + u + 1;
          f32x4 wn = wq;
          if (nxt < NT * KPB) wn = *(const f32x4*)(sW1 + (nxt * 64 + lane) * 4);
          z[c] = SPK_MFMA(wq.x, phi[u][0], z[c]);
          z[c] = SPK_MFMA(wq.y, phi[u][1], z[c]);
          z[c] = SPK_MFMA(wq.z, phi[u][2], z[c]);
          z[c] = SPK_MFMA(wq.w, phi[u][3], z[c]);
          if (BWD) {
            f32x16& q = zp[BWD ? c : 0];
            q = SPK_MFMA(wq.x, dphi[BWD ? u : 0][0], q);
            q = SPK_MFMA(wq.y, dphi[BWD ? u : 0][1], q);
            q = SPK_MFMA(wq.z, dphi[BWD ? u : 0][2], q);
            q = SPK_MFMA(wq.w, dphi[BWD ? u : 0][3], q);
          }
          wq = wn;
        }
      }
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float sp, sg;
          spk_fast_softplus_sigmoid(z[c][r], sp, sg);
          z[c][r] = sp - SPK_LN2_F;
          if (BWD) zp[BWD ? c : 0][r] *= sg;
        }
    }

    float dsum1 = 0.f, dsum2 = 0.f;

#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      // gathered rows of both end points, requested before the MFMAs of this tile
      f32x4 hjq[4], hiq[4], gyiq[BWD ? 4 : 1], gyjq[BWD ? 4 : 1], gsv[GS ? 4 : 1];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * t + 8 * q + 4 * hi;
        hjq[q] = *(const f32x4*)(a.h + j * NF + col);
        hiq[q] = *(const f32x4*)(a.h + i * NF + col);
        if (BWD) {
          gyiq[BWD ? q : 0] = *(const f32x4*)(a.gy + i * NF + col);
          gyjq[BWD ? q : 0] = *(const f32x4*)(a.gy + j * NF + col);
        }
        if (GS) gsv[GS ? q : 0] = *(const f32x4*)(a.gload + (tile * 32 + el) * NF + col);
      }
      f32x16 g, gp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { g[r] = sb2[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi]; gp[r] = 0.f; }
      {
        const float* wbase = sW2 + ((int64_t)t * KB2 * 64 + lane) * 4;
        f32x4 wq = *(const f32x4*)wbase;
#pragma unroll
        for (int ug = 0; ug < KB2; ++ug) {
          const int c = ug >> 2, q = ug & 3;
          f32x4 wn = wq;
          if (ug + 1 < KB2) wn = *(const f32x4*)(wbase + (ug + 1) * 256);
          if (!GS) {
            g = SPK_MFMA(wq.x, z[c][4 * q + 0], g);
            g = SPK_MFMA(wq.y, z[c][4 * q + 1], g);
            g = SPK_MFMA(wq.z, z[c][4 * q + 2], g);
            g = SPK_MFMA(wq.w, z[c][4 * q + 3], g);
          }
          if (BWD) {
            const f32x16& zq = zp[BWD ? c : 0];
            gp = SPK_MFMA(wq.x, zq[4 * q + 0], gp);
            gp = SPK_MFMA(wq.y, zq[4 * q + 1], gp);
            gp = SPK_MFMA(wq.z, zq[4 * q + 2], gp);
            gp = SPK_MFMA(wq.w, zq[4 * q + 3], gp);
          }
          wq = wn;
        }
      }
      if (GS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 gq = gsv[GS ? q : 0];
          g[4 * q + 0] = gq.x; g[4 * q + 1] = gq.y; g[4 * q + 2] = gq.z; g[4 * q + 3] = gq.w;
        }
      } else if (!BWD && a.gsave) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 gq;
          gq.x = g[4 * q + 0]; gq.y = g[4 * q + 1]; gq.z = g[4 * q + 2]; gq.w = g[4 * q + 3];
          *(f32x4*)(a.gsave + (tile * 32 + el) * NF + 32 * t + 8 * q + 4 * hi) = gq;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 pA, pB;
        const f32x4 hj = hjq[q], hc = hiq[q];
        const float W0 = g[4 * q + 0] * fc, W1v = g[4 * q + 1] * fc, W2v = g[4 * q + 2] * fc, W3 = g[4 * q + 3] * fc;
        if (!BWD) {
          pA.x = W0 * hj.x; pA.y = W1v * hj.y; pA.z = W2v * hj.z; pA.w = W3 * hj.w;
          pB.x = W0 * hc.x; pB.y = W1v * hc.y; pB.z = W2v * hc.z; pB.w = W3 * hc.w;
        } else {
          const f32x4 gyi = gyiq[BWD ? q : 0], gyj = gyjq[BWD ? q : 0];
          const float D0 = gp[4 * q + 0] * fc + g[4 * q + 0] * dfc, D1 = gp[4 * q + 1] * fc + g[4 * q + 1] * dfc;
          const float D2 = gp[4 * q + 2] * fc + g[4 * q + 2] * dfc, D3 = gp[4 * q + 3] * fc + g[4 * q + 3] * dfc;
          dsum1 += gyi.x * hj.x * D0 + gyi.y * hj.y * D1 + gyi.z * hj.z * D2 + gyi.w * hj.w * D3;
          dsum2 += gyj.x * hc.x * D0 + gyj.y * hc.y * D1 + gyj.z * hc.z * D2 + gyj.w * hc.w * D3;
          pA.x = W0 * gyj.x; pA.y = W1v * gyj.y; pA.z = W2v * gyj.z; pA.w = W3 * gyj.w;
          pB.x = W0 * gyi.x; pB.y = W1v * gyi.y; pB.z = W2v * gyi.z; pB.w = W3 * gyi.w;
        }
        if (!SKIPGH) {
          *(f32x4*)(myT + el * TP2 + 8 * q + 4 * hi) = pA;
          *(f32x4*)(myT + el * TP2 + 32 + 8 * q + 4 * hi) = pB;
        }
      }
      if (SKIPGH) continue;   // geometry gradient only: nothing to scatter
      spk_wave_lds_sync();
      {
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = myT[k * TP2 + lane];
        float* ybase = a.y + 32 * t + el;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          acc += v[k];
          const int ik = __builtin_amdgcn_readlane(ieff, k);
          const int jk = __builtin_amdgcn_readlane(jeff, k);
          const bool fl = hi ? true : (((flushmask >> k) & 1u) != 0);
          const int row = hi ? jk : ik;
          if (fl) {
            if (row >= 0) unsafeAtomicAdd(ybase + (int64_t)row * NF, acc);
            acc = 0.f;
          }
        }
      }
      spk_wave_lds_sync();
    }
    if (BWD) {
      dsum1 += __shfl_xor(dsum1, 32, 64);
      dsum2 += __shfl_xor(dsum2, 32, 64);
      if (hi == 0 && valid) {
        const float s1 = d > 0.f ? dsum1 / d : 0.f, s2 = d > 0.f ? dsum2 / d : 0.f;
        if (a.gr_assign) {
          a.gr[3 * e] = s1 * rx; a.gr[3 * e + 1] = s1 * ry; a.gr[3 * e + 2] = s1 * rz;
          a.gr[3 * e2] = -s2 * rx; a.gr[3 * e2 + 1] = -s2 * ry; a.gr[3 * e2 + 2] = -s2 * rz;
        } else {
          a.gr[3 * e] += s1 * rx; a.gr[3 * e + 1] += s1 * ry; a.gr[3 * e + 2] += s1 * rz;
          a.gr[3 * e2] -= s2 * rx; a.gr[3 * e2 + 1] -= s2 * ry; a.gr[3 * e2 + 2] -= s2 * rz;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Split-precision forms of the two pair kernels that carry the box regime (round 6; spk_split.h, spk_filter_split.h; n_filters = 128):
// the filter-network GEMMs as three v_mfma_f32_32x32x16_f16 products of (high, low) fp16 operand pairs with fp32 accumulation --
// per 32-pair tile 24 + 96 f16 instructions (3.8 k cycles of matrix-pipe time) instead of 12 KPB + 256 f32 ones (19 k), same layouts,
// same gathers, same accumulation of the results.  W1 / W2 are staged into LDS as split images (contraction index of W2 in accumulator
// order: the hidden activations go from the accumulator registers of GEMM 1 into GEMM 2 as they lie).
// ------------------------------------------------------------------------------------------
// forward: one wavefront per tile of 32 undirected pairs; GEMM 2 with rows = pairs, columns = channels (a lane owns a channel)
template <int KPB, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_cfconv_pair_sp(CfArgs a) {
  constexpr int NF = 128, NT = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  h16x8* sW2h = (h16x8*)smem;                               // 32 KB high | 32 KB low
  h16x8* sW2l = sW2h + 2048;
  char* sW1h = (char*)(smem + NF * NF);
  char* sW1l = sW1h + MlW1Image<KPB>::BYTES;
  float* sb1 = (float*)(sW1l + MlW1Image<KPB>::BYTES);
  float* sb2 = sb1 + NF;
  EdgeRec* sE = (EdgeRec*)(sb2 + NF);                       // NWAVES * 32 records
  float* sRb = (float*)(sE + NWAVES * 32);                  // [2][32] radial-basis parameters (read per tile from LDS: as loop-invariant global
  int* sCnt = (int*)(sRb + 64);                             //  loads the compiler hoists all 24 of a lane out of the tile loop -- and spills them)

  if (threadIdx.x < 64) {
    const int k = threadIdx.x & 31;
    const float* src = (threadIdx.x < 32) ? a.rb.p0 : a.rb.p1;
    sRb[threadIdx.x] = (src && k < a.rb.n_rbf) ? src[k] : 1.0f;
  }
  ml_stage_w2_split<NWAVES * 64>(sW2h, sW2l, a.w2, threadIdx.x);
  ml_stage_w1_split<KPB>(sW1h, sW1l, a.w1, a.rb.n_rbf, threadIdx.x);
  for (int s = threadIdx.x; s < NF; s += NWAVES * 64) { sb1[s] = a.b1[s]; sb2[s] = a.b2[s]; }
  if (threadIdx.x == 0) sCnt[0] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  EdgeRec* myE = sE + wv * 32;
  const int gp1 = (int)(a.n_half_dev ? (int64_t)a.n_half_dev[0] : a.n_half);
  const int gtiles = (gp1 + 31) / 32;

  while (true) {
    int nidx = 0;
    if (lane == 0) nidx = atomicAdd(&sCnt[0], 1);
    nidx = __builtin_amdgcn_readfirstlane(nidx);
    const int ltile = a.xcd_walk ? (int)spk_xcd_tile(nidx, gtiles) : (int)blockIdx.x + nidx * (int)gridDim.x;
    if (ltile >= gtiles) break;
    const int64_t gtile = ltile;
    // (lane re-derived through an opaque asm per tile: the per-lane LDS offsets of the loop body -- basis parameters, weight slots,
    //  records -- are otherwise hoisted out of the persistent loop and parked in scratch: 440 B/lane)
    int lane_o_ = lane;
    asm volatile("" : "+v"(lane_o_));
    const int lane = lane_o_, hi = lane >> 5, el = lane & 31;

    // ---- per-pair geometry (lanes 32..63 mirror lanes 0..31)
    const int pfirst = 32 * ltile;
    const int nvalid = (gp1 - pfirst) < 32 ? (gp1 - pfirst) : 32;
    const bool valid = el < nvalid;
    const int64_t e = a.half[pfirst + (valid ? el : (nvalid - 1))];
    const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
    const int i = (int)a.idx_i[e], j = (int)a.idx_j[e];
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    float fc, dfc;
    spk_cutoff_eval_fast(a.rb.cutoff, d, fc, dfc);
    if (!valid) { fc = 0.f; dfc = 0.f; }
    if (hi == 0) { EdgeRec er; er.i = i; er.j = j; er.fc = fc; er.dfc = dfc; myE[el] = er; }

    // ---- GEMM 1 (rows = hidden channels, columns = pairs): z = ssp(W1 phi + b1), kept as the split A operand of GEMM 2
    h16x8 zh[NT][2], zl[NT][2];
    {
      h16x8 ph[2], pl[2], dh_[2], dl_[2];
      ml_basis_split<KPB, false>(a.rb.kind, a.rb.n_rbf, sRb, sRb + 32, hi, d, ph, pl, dh_, dl_);
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        f32x16 zc, zx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { zc[r] = sb1[32 * c + pair_of(r, hi)]; zx[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < (KPB > 2 ? 2 : 1); ++s) {
          h16x8 wh, wl;
          ml_w1_operand<KPB>(sW1h, sW1l, s, c * 64 + lane, wh, wl);
          SP_STEP(wh, wl, ph[s], pl[s], zc, zx);
        }
        SP_FOLD(zc, zx);
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = spk_fast_ssp(zc[8 * sp + k]);
          sp_split8(v, zh[c][sp], zl[c][sp]);
        }
        __builtin_amdgcn_sched_barrier(0);      // one hidden tile at a time: interleaved, the four accumulator pairs push the kernel into scratch
      }
    }
    spk_wave_lds_sync();   // myE visible to the whole wave
    unsigned runmask = 0x8000u;       // bit r set <=> the centre atom changes after register r
    {
      int prev = myE[pair_of(0, hi)].i;
#pragma unroll
      for (int r = 1; r < 16; ++r) {
        const int cur = myE[pair_of(r, hi)].i;
        if (cur != prev) runmask |= 1u << (r - 1);
        prev = cur;
      }
    }

#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      const int c0 = 32 * t + el;   // this lane's channel
      float hj[16], hc[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const EdgeRec er = myE[pair_of(r, hi)];
        const unsigned oi = (unsigned)er.i * NF + c0, oj = (unsigned)er.j * NF + c0;      // (n_atoms * nf < 2^31, checked by the launcher)
        hj[r] = a.h[oj];
        hc[r] = a.h[oi];
      }
      f32x16 g, gx;
      const float bias2 = sb2[c0];
#pragma unroll
      for (int r = 0; r < 16; ++r) { g[r] = bias2; gx[r] = 0.f; }
      {
        const h16x8* wbh = sW2h + (t * 8) * 64 + lane;
        const h16x8* wbl = sW2l + (t * 8) * 64 + lane;
        h16x8 wh = wbh[0], wl = wbl[0];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          h16x8 nh = wh, nl = wl;
          if (s + 1 < 8) { nh = wbh[(s + 1) * 64]; nl = wbl[(s + 1) * 64]; }
          SP_STEP(zh[s >> 1][s & 1], zl[s >> 1][s & 1], wh, wl, g, gx);
          wh = nh; wl = nl;
        }
      }
      SP_FOLD(g, gx);
      if (a.gsave) {
        float* gts = a.gsave + gtile * 32 * NF;   // this tile's filters, saved for the backward
#pragma unroll
        for (int r = 0; r < 16; ++r) gts[(unsigned)pair_of(r, hi) * NF + c0] = g[r];
      }
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const EdgeRec er = myE[pair_of(r, hi)];
        const float W = g[r] * er.fc;
        acc += W * hj[r];
        unsafeAtomicAdd(a.y + ((unsigned)er.j * NF + c0), W * hc[r]);
        if ((runmask >> r) & 1u) { unsafeAtomicAdd(a.y + ((unsigned)er.i * NF + c0), acc); acc = 0.f; }
      }
    }
    spk_wave_lds_sync();   // myE may be rewritten by the next tile
  }
}

// saved-filter backward (rows = channels, columns = pairs: a lane owns a pair), SKIPGH as in k_cfconv_pair_t
template <int KPB, int NWAVES, bool SKIPGH>
__global__ __launch_bounds__(NWAVES * 64) void k_cfconv_pair_t_sp(CfArgs a) {
  constexpr int NF = 128, NT = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  h16x8* sW2h = (h16x8*)smem;
  h16x8* sW2l = sW2h + 2048;
  char* sW1h = (char*)(smem + NF * NF);
  char* sW1l = sW1h + MlW1Image<KPB>::BYTES;
  float* sb1 = (float*)(sW1l + MlW1Image<KPB>::BYTES);
  float* sT = sb1 + 2 * NF;
  float* sRb = sT + NWAVES * 32 * TP2;                      // [2][32] radial-basis parameters
  int* sCnt = (int*)(sRb + 64);

  if (threadIdx.x < 64) {
    const int k = threadIdx.x & 31;
    const float* src = (threadIdx.x < 32) ? a.rb.p0 : a.rb.p1;
    sRb[threadIdx.x] = (src && k < a.rb.n_rbf) ? src[k] : 1.0f;
  }
  ml_stage_w2_split<NWAVES * 64>(sW2h, sW2l, a.w2, threadIdx.x);
  ml_stage_w1_split<KPB>(sW1h, sW1l, a.w1, a.rb.n_rbf, threadIdx.x);
  for (int s = threadIdx.x; s < NF; s += NWAVES * 64) sb1[s] = a.b1[s];
  if (threadIdx.x == 0) sCnt[0] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  float* myT = sT + wv * (32 * TP2);
  const int64_t nhalf = a.n_half_dev ? (int64_t)a.n_half_dev[0] : a.n_half;
  const int64_t ntiles = (nhalf + 31) / 32;

  while (true) {
    int nidx = 0;
    if (lane == 0) nidx = atomicAdd(&sCnt[0], 1);
    nidx = __builtin_amdgcn_readfirstlane(nidx);
    const int64_t tile = a.xcd_walk ? spk_xcd_tile(nidx, ntiles) : (int64_t)blockIdx.x + (int64_t)nidx * gridDim.x;
    if (tile >= ntiles) break;
    int lane_o_ = lane;      // (see k_cfconv_pair_sp)
    asm volatile("" : "+v"(lane_o_));
    const int lane = lane_o_, hi = lane >> 5, el = lane & 31;

    const int64_t hidx = tile * 32 + el;
    const bool valid = hidx < nhalf;
    const int64_t e = a.half[valid ? hidx : (nhalf - 1)];
    const int64_t e2 = a.rev[e];
    const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
    const int64_t j = a.idx_j[e];
    const int64_t i = a.idx_i[e];
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    float fc, dfc;
    spk_cutoff_eval_fast(a.rb.cutoff, d, fc, dfc);
    if (!valid) { fc = 0.f; dfc = 0.f; }
    const int ieff = valid ? (int)i : -1;
    const int jeff = valid ? (int)j : -1;
    const int inext = __shfl(ieff, (lane + 1) & 63, 64);
    const unsigned long long bal = __ballot(ieff != inext);
    const unsigned flushmask = __builtin_amdgcn_readfirstlane((unsigned)(bal & 0xffffffffull)) | 0x80000000u;

    // ---- GEMM 1, value and derivative: z' = sigmoid(W1 phi + b1) (W1 phi'), kept as the split B operand of GEMM 2'
    h16x8 zph[NT][2], zpl[NT][2];
    {
      h16x8 ph[2], pl[2], dh[2], dl[2];
      ml_basis_split<KPB, true>(a.rb.kind, a.rb.n_rbf, sRb, sRb + 32, hi, d, ph, pl, dh, dl);
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        f32x16 zc, zcx, zq, zqx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { zc[r] = sb1[32 * c + pair_of(r, hi)]; zcx[r] = 0.f; zq[r] = 0.f; zqx[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < (KPB > 2 ? 2 : 1); ++s) {
          h16x8 wh, wl;
          ml_w1_operand<KPB>(sW1h, sW1l, s, c * 64 + lane, wh, wl);
          SP_STEP(wh, wl, ph[s], pl[s], zc, zcx);
          SP_STEP(wh, wl, dh[s], dl[s], zq, zqx);
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int r = 8 * sp + k;
            float spv, sg;
            spk_fast_softplus_sigmoid(fmaf(zcx[r], SP_DOWN, zc[r]), spv, sg);
            v[k] = fmaf(zqx[r], SP_DOWN, zq[r]) * sg;
          }
          sp_split8(v, zph[c][sp], zpl[c][sp]);
        }
        __builtin_amdgcn_sched_barrier(0);      // one hidden tile at a time (register pressure)
      }
    }

    float dsum1 = 0.f, dsum2 = 0.f;
#ifdef SPK_DBG_DSUM
    float dbg_p0 = 0.f, dbg_p1 = 0.f, dbg_p2 = 0.f;
#endif
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      f32x4 hjq[4], hiq[4], gyiq[4], gyjq[4], gsv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * t + 8 * q + 4 * hi;
        hjq[q] = *(const f32x4*)(a.h + j * NF + col);
        gyiq[q] = *(const f32x4*)(a.gy + i * NF + col);
        gsv[q] = *(const f32x4*)(a.gload + (tile * 32 + el) * NF + col);
        hiq[q] = *(const f32x4*)(a.h + i * NF + col);
        gyjq[q] = *(const f32x4*)(a.gy + j * NF + col);
      }
      f32x16 gp, gpx;
#pragma unroll
      for (int r = 0; r < 16; ++r) { gp[r] = 0.f; gpx[r] = 0.f; }
      {
        const h16x8* wbh = sW2h + (t * 8) * 64 + lane;
        const h16x8* wbl = sW2l + (t * 8) * 64 + lane;
        h16x8 wh = wbh[0], wl = wbl[0];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          h16x8 nh = wh, nl = wl;
          if (s + 1 < 8) { nh = wbh[(s + 1) * 64]; nl = wbl[(s + 1) * 64]; }
          SP_STEP(wh, wl, zph[s >> 1][s & 1], zpl[s >> 1][s & 1], gp, gpx);
          wh = nh; wl = nl;
        }
      }
      SP_FOLD(gp, gpx);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 pA, pB;
        const f32x4 hj = hjq[q], hc = hiq[q], gq = gsv[q];
        const float W0 = gq.x * fc, W1v = gq.y * fc, W2v = gq.z * fc, W3 = gq.w * fc;
        const f32x4 gyi = gyiq[q], gyj = gyjq[q];
        const float D0 = gp[4 * q + 0] * fc + gq.x * dfc, D1 = gp[4 * q + 1] * fc + gq.y * dfc;
        const float D2 = gp[4 * q + 2] * fc + gq.z * dfc, D3 = gp[4 * q + 3] * fc + gq.w * dfc;
        dsum1 += gyi.x * hj.x * D0 + gyi.y * hj.y * D1 + gyi.z * hj.z * D2 + gyi.w * hj.w * D3;
        dsum2 += gyj.x * hc.x * D0 + gyj.y * hc.y * D1 + gyj.z * hc.z * D2 + gyj.w * hc.w * D3;
        pA.x = W0 * gyj.x; pA.y = W1v * gyj.y; pA.z = W2v * gyj.z; pA.w = W3 * gyj.w;
        pB.x = W0 * gyi.x; pB.y = W1v * gyi.y; pB.z = W2v * gyi.z; pB.w = W3 * gyi.w;
        if (!SKIPGH) {
          *(f32x4*)(myT + el * TP2 + 8 * q + 4 * hi) = pA;
          *(f32x4*)(myT + el * TP2 + 32 + 8 * q + 4 * hi) = pB;
        }
      }
#ifdef SPK_DBG_DSUM
      if (t == 0) dbg_p0 = dsum2; else if (t == 1) dbg_p1 = dsum2; else if (t == 2) dbg_p2 = dsum2;
#endif
      if (SKIPGH) continue;   // geometry gradient only: nothing to scatter
      spk_wave_lds_sync();
      {
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = myT[k * TP2 + lane];
        float* ybase = a.y + 32 * t + el;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          acc += v[k];
          const int ik = __builtin_amdgcn_readlane(ieff, k);
          const int jk = __builtin_amdgcn_readlane(jeff, k);
          const bool fl = hi ? true : (((flushmask >> k) & 1u) != 0);
          const int row = hi ? jk : ik;
          if (fl) {
            if (row >= 0) unsafeAtomicAdd(ybase + (int64_t)row * NF, acc);
            acc = 0.f;
          }
        }
      }
      spk_wave_lds_sync();
    }
#ifdef SPK_DBG_DSUM
    if (a.dbg) { float* o = (float*)a.dbg + 4 * (ntiles * 32) + 4 * (tile * 64 + lane); o[0] = dbg_p0; o[1] = dbg_p1; o[2] = dbg_p2; o[3] = dsum2; }
#endif
    dsum1 += __shfl_xor(dsum1, 32, 64);
    dsum2 += __shfl_xor(dsum2, 32, 64);
    if (hi == 0 && valid) {
      const float s1 = d > 0.f ? dsum1 / d : 0.f, s2 = d > 0.f ? dsum2 / d : 0.f;
#ifdef SPK_DBG_DSUM
      if (a.dbg) { float* o = (float*)a.dbg + 4 * (tile * 32 + el); o[0] = s1; o[1] = s2; o[2] = __int_as_float((int)e); o[3] = __int_as_float((int)e2); }
#endif
      if (a.gr_assign) {
        a.gr[3 * e] = s1 * rx; a.gr[3 * e + 1] = s1 * ry; a.gr[3 * e + 2] = s1 * rz;
        a.gr[3 * e2] = -s2 * rx; a.gr[3 * e2 + 1] = -s2 * ry; a.gr[3 * e2 + 2] = -s2 * rz;
      } else {
        a.gr[3 * e] += s1 * rx; a.gr[3 * e + 1] += s1 * ry; a.gr[3 * e + 2] += s1 * rz;
        a.gr[3 * e2] -= s2 * rx; a.gr[3 * e2 + 1] -= s2 * ry; a.gr[3 * e2 + 2] -= s2 * rz;
      }
    }
  }
}

template <int KPB>
static int launch_pair_sp(const CfArgs& a, hipStream_t stream) {
  constexpr int NWAVES = 8, NF = 128;
  const size_t lds = (size_t)(NF * NF + 2 * NF) * sizeof(float) + 2 * MlW1Image<KPB>::BYTES + (size_t)NWAVES * 32 * sizeof(EdgeRec) + (64 + 4) * sizeof(int);
  auto kern = k_cfconv_pair_sp<KPB, NWAVES>;
  static SpkPerDevice attr_set;
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.mark(attr_dev);
  }
  int grid = (int)(((a.n_half + 31) / 32 + NWAVES - 1) / NWAVES);
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  if (grid < 1) grid = 1;
  SpkProfScope prof("cfconv_fwd_pair", stream);
  CfArgs ax = a;
  ax.xcd_walk = (spk_xcd_walk_default() && grid % 8 == 0 && (a.n_half + 31) / 32 >= 16 * (int64_t)grid) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, stream, ax);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
template <int KPB>
static int launch_pair_t_bwd_gs_sp(const CfArgs& a, hipStream_t stream) {
  constexpr int NWAVES = 8, NF = 128;
  const size_t lds = (size_t)(NF * NF + 2 * NF + NWAVES * 32 * TP2 + 64) * sizeof(float) + 2 * MlW1Image<KPB>::BYTES + 4 * sizeof(int);
  auto kern = a.skip_gh ? k_cfconv_pair_t_sp<KPB, NWAVES, true> : k_cfconv_pair_t_sp<KPB, NWAVES, false>;
  static SpkPerDevice attr_set;
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_cfconv_pair_t_sp<KPB, NWAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_cfconv_pair_t_sp<KPB, NWAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_dev);
  }
  const int64_t ntiles = (a.n_half + 31) / 32;
  int grid = (int)((ntiles + NWAVES - 1) / NWAVES);
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  if (grid < 1) grid = 1;
  SpkProfScope prof(a.skip_gh ? "cfconv_bwd_pair_gs_geom" : "cfconv_bwd_pair_gs", stream);
  CfArgs ax = a;
  ax.xcd_walk = (spk_xcd_walk_default() && grid % 8 == 0 && ntiles >= 16 * (int64_t)grid) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, stream, ax);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------
// Row-tile forward (round 6): y[i] = sum_j W_ij o h[j] with a wavefront per ROW of the sorted list -- the filter of every DIRECTED edge from the split
// GEMMs of 32-edge chunks of the row (twice the matrix work of the pair kernel, affordable at the f16 rate: 120 instructions of 32 cycles per chunk),
// lanes own a channel, the sum over the row is a register sum parked in LDS between chunks and added over the two half-waves at the end: no float
// atomics (the pair kernel needs one per pair and channel for the neighbour's direction: 110 M per launch on the water box), y written once, a fixed
// summation order.  The raw filters of the CANONICAL edges are saved for the pair backward in its layout (row = position of the pair in `half`).
// Lists without a per-call compaction only (a.n_half_dev == null): the saved rows are addressed through the plan's edge_pair.
// ------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) RtRec { int j; int pos; float fc; float pad; };

template <int KPB, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_cfconv_rowtile_fwd(CfArgs a) {
  constexpr int NF = 128, NT = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  h16x8* sW2h = (h16x8*)smem;                               // 32 KB high | 32 KB low
  h16x8* sW2l = sW2h + 2048;
  char* sW1h = (char*)(smem + NF * NF);
  char* sW1l = sW1h + MlW1Image<KPB>::BYTES;
  float* sb1 = (float*)(sW1l + MlW1Image<KPB>::BYTES);
  float* sb2 = sb1 + NF;
  RtRec* sE = (RtRec*)(sb2 + NF);                           // NWAVES * 32 records
  float* sRb = (float*)(sE + NWAVES * 32);                  // [2][32] radial-basis parameters
  float* sAcc = sRb + 64;                                   // NWAVES x NT x 64 running sums

  if (threadIdx.x < 64) {
    const int k = threadIdx.x & 31;
    const float* src = (threadIdx.x < 32) ? a.rb.p0 : a.rb.p1;
    sRb[threadIdx.x] = (src && k < a.rb.n_rbf) ? src[k] : 1.0f;
  }
  ml_stage_w2_split<NWAVES * 64>(sW2h, sW2l, a.w2, threadIdx.x);
  ml_stage_w1_split<KPB>(sW1h, sW1l, a.w1, a.rb.n_rbf, threadIdx.x);
  for (int s = threadIdx.x; s < NF; s += NWAVES * 64) { sb1[s] = a.b1[s]; sb2[s] = a.b2[s]; }
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  RtRec* myE = sE + wv * 32;
  float* myA = sAcc + wv * (NT * 64);

  // the workgroups of one XCD walk a contiguous eighth of the atoms (their neighbours' rows are then shared inside that XCD's L2)
  const int64_t per_xcd = a.xcd_walk ? (a.N + 7) / 8 : a.N;
  const int64_t a_lo = a.xcd_walk ? (int64_t)(blockIdx.x & 7) * per_xcd : 0;
  const int64_t a_hi = a.xcd_walk ? (a_lo + per_xcd < a.N ? a_lo + per_xcd : a.N) : a.N;
  const int64_t a_first = a.xcd_walk ? a_lo + (int64_t)(blockIdx.x >> 3) * NWAVES + wv : (int64_t)blockIdx.x * NWAVES + wv;
  const int64_t a_step = a.xcd_walk ? (int64_t)((gridDim.x + 7) >> 3) * NWAVES : (int64_t)gridDim.x * NWAVES;
  for (int64_t atom = a_first; atom < a_hi; atom += a_step) {
    const int32_t e0 = a.rowptr[atom], e1 = a.rowptr[atom + 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) myA[t * 64 + lane] = 0.f;
    for (int32_t cs = e0; cs < e1; cs += 32) {
      // (lane re-derived through an opaque asm per chunk: see k_cfconv_pair_sp)
      int lane_o_ = lane;
      asm volatile("" : "+v"(lane_o_));
      const int lane = lane_o_, hi = lane >> 5, el = lane & 31;
      // ---- geometry of this lane's edge (lanes 32..63 mirror lanes 0..31)
      const bool valid = cs + el < e1;
      const int64_t e = valid ? cs + el : e1 - 1;
      const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
      const int j = (int)a.idx_j[e];
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      float fc, dfc;
      spk_cutoff_eval_fast(a.rb.cutoff, d, fc, dfc);
      if (!valid) fc = 0.f;
      if (hi == 0) {
        RtRec rec; rec.j = j; rec.fc = fc; rec.pad = 0.f; rec.pos = -1;
        if (a.gsave && a.edge_pair && valid) { const int ep = a.edge_pair[e]; rec.pos = (a.half[ep] == (int32_t)e) ? ep : -1; }
        myE[el] = rec;
      }
      // ---- GEMM 1 (rows = hidden channels, columns = edges): z = ssp(W1 phi + b1), kept as the split A operand of GEMM 2
      h16x8 zh[NT][2], zl[NT][2];
      {
        h16x8 ph[2], pl[2], dh_[2], dl_[2];
        ml_basis_split<KPB, false>(a.rb.kind, a.rb.n_rbf, sRb, sRb + 32, hi, d, ph, pl, dh_, dl_);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
          f32x16 zc, zx;
#pragma unroll
          for (int r = 0; r < 16; ++r) { zc[r] = sb1[32 * c + pair_of(r, hi)]; zx[r] = 0.f; }
#pragma unroll
          for (int s = 0; s < (KPB > 2 ? 2 : 1); ++s) {
            h16x8 wh, wl;
            ml_w1_operand<KPB>(sW1h, sW1l, s, c * 64 + lane, wh, wl);
            SP_STEP(wh, wl, ph[s], pl[s], zc, zx);
          }
          SP_FOLD(zc, zx);
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = spk_fast_ssp(zc[8 * sp + k]);
            sp_split8(v, zh[c][sp], zl[c][sp]);
          }
          __builtin_amdgcn_sched_barrier(0);      // one hidden tile at a time (register pressure)
        }
      }
      spk_wave_lds_sync();   // records visible to the whole wave

#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        const int c0 = 32 * t + el;   // this lane's channel
        float hj[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) hj[r] = *(const float*)((const char*)a.h + ((unsigned)myE[pair_of(r, hi)].j * NF + c0) * 4u);
        __builtin_amdgcn_sched_barrier(0);        // the 16 gathers of the block are requested before its GEMM
        f32x16 g, gx;
        const float bias2 = sb2[c0];
#pragma unroll
        for (int r = 0; r < 16; ++r) { g[r] = bias2; gx[r] = 0.f; }
        {
          const h16x8* wbh = sW2h + (t * 8) * 64 + lane;
          const h16x8* wbl = sW2l + (t * 8) * 64 + lane;
          h16x8 wh = wbh[0], wl = wbl[0];
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            h16x8 nh = wh, nl = wl;
            if (s + 1 < 8) { nh = wbh[(s + 1) * 64]; nl = wbl[(s + 1) * 64]; }
            SP_STEP(zh[s >> 1][s & 1], zl[s >> 1][s & 1], wh, wl, g, gx);
            wh = nh; wl = nl;
          }
        }
        SP_FOLD(g, gx);
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const RtRec er = myE[pair_of(r, hi)];
          if (a.gsave && er.pos >= 0) a.gsave[(size_t)er.pos * NF + c0] = g[r];     // raw filter of the canonical edge, for the pair backward
          acc = fmaf(g[r] * er.fc, hj[r], acc);
        }
        myA[t * 64 + lane] += acc;
      }
      spk_wave_lds_sync();   // records may be rewritten by the next chunk
    }
    spk_wave_lds_sync();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float tot = myA[t * 64 + lane] + myA[t * 64 + (lane ^ 32)];
      if (hi == 0) a.y[(size_t)atom * NF + 32 * t + el] = tot;
    }
    spk_wave_lds_sync();
  }
}

static bool cfconv_rowtile_fwd_ok(const CfArgs& a) {
  static const int env = [] { const char* e = getenv("SPK_CF_ROWTILE"); return e ? (e[0] == '1' ? 1 : -1) : 0; }();
  // (saving filters for the pair backward needs the plan's pair positions and a list without per-call compaction; a forward that saves nothing -- asymmetric
  //  lists, the by-neighbour pass of their backward -- only needs the rows)
  if (env < 0 || !a.rowptr || a.N * (int64_t)128 >= (1LL << 30)) return false;
  if (a.gsave && (!a.edge_pair || !a.half || a.n_half_dev)) return false;
  return env > 0 || a.E >= (1 << 19);
}

template <int KPB>
static int launch_rowtile_fwd(const CfArgs& a, hipStream_t stream) {
  // sixteen wavefronts per workgroup, one workgroup per CU (the weight images take 77 KB of LDS): measured on the water box 8 / 12 / 16 waves = 444 / 391 / 382 us
  // (the pair kernel with its atomics: 437 us) -- the chunk is a long dependent chain (GEMM 1, softplus, split, GEMM 2 per block) that needs other waves to fill it
#ifdef SPK_CF_RT_WAVES
  constexpr int NWAVES = SPK_CF_RT_WAVES, NF = 128, NT = 4;
#else
  constexpr int NWAVES = 16, NF = 128, NT = 4;
#endif
  const size_t lds = (size_t)(NF * NF + 2 * NF) * sizeof(float) + 2 * MlW1Image<KPB>::BYTES + (size_t)NWAVES * 32 * sizeof(RtRec) + 64 * sizeof(float) +
                     (size_t)NWAVES * NT * 64 * sizeof(float);
  auto kern = k_cfconv_rowtile_fwd<KPB, NWAVES>;
  static SpkPerDevice attr_set;
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.mark(attr_dev);
  }
  int grid = (int)((a.N + NWAVES - 1) / NWAVES);
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  if (grid < 1) grid = 1;
  SpkProfScope prof("cfconv_fwd_rowtile", stream);
  CfArgs ax = a;
  ax.xcd_walk = (spk_xcd_walk_default() && grid % 8 == 0 && a.N >= (1 << 14)) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, stream, ax);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static int check_graph(const spk_graph_t* g, const char* who) {
  SPK_CHECK_ARG(g != nullptr, "%s: null graph", who);
  SPK_CHECK_ARG(g->n_atoms >= 0 && g->n_edges >= 0 && g->n_atoms < (1LL << 31) && g->n_edges < (1LL << 31), "%s: bad graph sizes", who);
  SPK_CHECK_ARG(g->n_edges == 0 || (g->idx_i && g->idx_j), "%s: null index arrays", who);
  return SPK_OK;
}

template <int NF, int KPB, bool BWD, bool SYM>
static int launch_mfma(const CfArgs& a, hipStream_t stream) {
  // forward: 2 waves/SIMD (195 VGPRs); backward keeps value + derivative tiles of the hidden layer
  // (128 VGPRs) plus prefetched gathers live => 1 wave/SIMD with the full 512-VGPR budget
  constexpr int NWAVES = BWD ? 4 : 8;
  const size_t lds = (size_t)(NF * NF + NF * KPB * 8 + 2 * NF + NWAVES * 32 * TP2) * sizeof(float) + 4 * sizeof(int);
  auto kern = k_cfconv_mfma<NF, KPB, NWAVES, BWD, SYM>;
  static SpkPerDevice attr_set;  // once per instantiation (not a stream operation; keep it out of graph capture)
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_dev);
  }
  const int64_t ntiles = (a.E + 31) / 32;
  // one persistent workgroup per CU; small problems use fewer, fuller workgroups
  int grid = (int)((ntiles + NWAVES - 1) / NWAVES);
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  if (grid < 1) grid = 1;
  SpkProfScope prof(BWD ? (SYM ? "cfconv_bwd_mfma_sym" : "cfconv_bwd_mfma_atomic") : "cfconv_fwd_mfma", stream);
  CfArgs ax = a;
  ax.xcd_walk = (spk_xcd_walk_default() && grid % 8 == 0 && ntiles >= 16 * (int64_t)grid) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, stream, ax);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

template <int NF, int KPB, bool BWD, bool GS, bool MOL>
static int launch_pair(const CfArgs& a, hipStream_t stream) {
  // forward and the saved-filter backward: 2 waves/SIMD (<= 256 VGPRs); recomputing backward: 1 wave/SIMD
  constexpr int NWAVES = (BWD && !GS) ? 4 : 8;
  const size_t lds = (size_t)(NF * NF + NF * KPB * 8 + 2 * NF + (BWD ? NWAVES * 2 * 32 * 33 : 0) + (MOL ? (size_t)a.max_group_atoms * NF : 0)) * sizeof(float) +
                     (size_t)NWAVES * 32 * sizeof(EdgeRec) + 4 * sizeof(int);
  auto kern = k_cfconv_pair<NF, KPB, NWAVES, BWD, GS, MOL>;
  static SpkPerDevice attr_set;
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.mark(attr_dev);
  }
  int grid;
  if (MOL) grid = a.n_groups;
  else grid = (int)(((a.n_half + 31) / 32 + NWAVES - 1) / NWAVES);
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  if (grid < 1) grid = 1;
  SpkProfScope prof(BWD ? (GS ? (MOL ? "cfconv_bwd_mol_gs" : "cfconv_bwd_pair_gs") : (MOL ? "cfconv_bwd_mol" : "cfconv_bwd_pair"))
                        : (MOL ? "cfconv_fwd_mol" : "cfconv_fwd_pair"), stream);
  CfArgs ax = a;
  ax.xcd_walk = (!MOL && spk_xcd_walk_default() && grid % 8 == 0 && (a.n_half + 31) / 32 >= 16 * (int64_t)grid) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, stream, ax);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

template <int NF, int KPB>
static int launch_pair_t_bwd_gs(const CfArgs& a, hipStream_t stream) {
  constexpr int NWAVES = 8;
  const size_t lds = (size_t)(NF * NF + NF * KPB * 8 + 2 * NF + NWAVES * 32 * TP2) * sizeof(float) + 4 * sizeof(int);
  auto kern = a.skip_gh ? k_cfconv_pair_t<NF, KPB, NWAVES, true, true, true> : k_cfconv_pair_t<NF, KPB, NWAVES, true, true, false>;
  static SpkPerDevice attr_set;
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_cfconv_pair_t<NF, KPB, NWAVES, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_cfconv_pair_t<NF, KPB, NWAVES, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_dev);
  }
  const int64_t ntiles = (a.n_half + 31) / 32;
  int grid = (int)((ntiles + NWAVES - 1) / NWAVES);
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  if (grid < 1) grid = 1;
  SpkProfScope prof(a.skip_gh ? "cfconv_bwd_pair_gs_geom" : "cfconv_bwd_pair_gs", stream);
  CfArgs ax = a;
  ax.xcd_walk = (spk_xcd_walk_default() && grid % 8 == 0 && ntiles >= 16 * (int64_t)grid) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, stream, ax);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// largest group (atoms) whose LDS accumulator fits next to the staged weights
static int mol_max_atoms(int NF, int kpb) {
  const size_t fixed = (size_t)(NF * NF + NF * kpb * 8 + 2 * NF + 8 * 2 * 32 * 33) * sizeof(float) + 8 * 32 * sizeof(EdgeRec) + 64;
  const size_t avail = 160 * 1024 - fixed;
  return (int)(avail / (NF * sizeof(float)));
}

template <bool BWD>
static int launch_simple(const CfArgs& a, int NF, int sym, hipStream_t stream) {
  int threads = ((NF > a.rb.n_rbf ? NF : a.rb.n_rbf) + 63) / 64 * 64;
  SPK_CHECK_ARG(threads <= 1024, "cfconv: n_filters=%d too large", NF);
  const size_t lds = (size_t)(2 * a.rb.n_rbf + 2 * NF + 16) * sizeof(float);
  int grid = (int)(a.E < 65535 * 16 ? a.E : 65535 * 16);
  if (grid < 1) grid = 1;
  SpkProfScope prof(BWD ? "cfconv_bwd_simple" : "cfconv_fwd_simple", stream);
  hipLaunchKernelGGL(k_cfconv_simple<BWD>, dim3(grid), dim3(threads), lds, stream, a, NF, sym);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

template <bool BWD>
static int cfconv_dispatch(const CfArgs& a, int NF, bool sym, hipStream_t stream, const char* who) {
  const int variant = spk_get_variant();
  const int kpb = (a.rb.n_rbf + 7) / 8;
  const bool al = (((uintptr_t)a.h | (uintptr_t)a.gy | (uintptr_t)a.y | (uintptr_t)a.w1 | (uintptr_t)a.w2) % 16) == 0;
  const bool mfma_ok = (NF == 128 || NF == 64) && kpb >= 1 && kpb <= 4 && al;
  SPK_CHECK_ARG(variant != SPK_VARIANT_MFMA || mfma_ok, "%s: shape nf=%d n_rbf=%d not supported by the MFMA kernel", who, NF, a.rb.n_rbf);
  if (!mfma_ok || variant == SPK_VARIANT_SIMPLE) return launch_simple<BWD>(a, NF, sym ? 1 : 0, stream);
  const bool pair = sym && a.half && a.rev && a.n_half > 0 && variant != SPK_VARIANT_MFMA_DIRECTED &&
                    a.N * (int64_t)NF < (1LL << 31);
  // group-local LDS accumulation: measured SLOWER than global float atomics on gfx950 (ds_add_f32 rate),
  // kept only as an explicitly selectable experiment
  const bool mol = pair && a.n_groups > 0 && a.grp_atom0 && a.max_group_atoms <= mol_max_atoms(NF, kpb) &&
                   variant == SPK_VARIANT_MFMA_MOL;
#define SPK_CF_CASE(NFv, KPBv)                                                      \
  if (NF == NFv && kpb == KPBv) {                                                   \
    if (mol && BWD && a.gload) return launch_pair<NFv, KPBv, BWD, BWD, true>(a, stream);   \
    if (mol) return launch_pair<NFv, KPBv, BWD, false, true>(a, stream);                   \
    if (pair && BWD && a.gload) return (NFv == 128 && spk_get_split() && !getenv("SPK_CF_NOSP_BWD")) ? launch_pair_t_bwd_gs_sp<KPBv>(a, stream) : launch_pair_t_bwd_gs<NFv, KPBv>(a, stream);   \
    if (!mol && !BWD && NFv == 128 && spk_get_split() && cfconv_rowtile_fwd_ok(a)) return launch_rowtile_fwd<KPBv>(a, stream);   \
    if (pair && !BWD && NFv == 128 && spk_get_split() && !getenv("SPK_CF_NOSP_FWD")) return launch_pair_sp<KPBv>(a, stream);   \
    if (pair) return launch_pair<NFv, KPBv, BWD, false, false>(a, stream);                 \
    if (!BWD) return launch_mfma<NFv, KPBv, false, false>(a, stream);               \
    return sym ? launch_mfma<NFv, KPBv, BWD, true>(a, stream)                       \
               : launch_mfma<NFv, KPBv, BWD, false>(a, stream);                     \
  }
  SPK_CF_CASE(128, 1) SPK_CF_CASE(128, 2) SPK_CF_CASE(128, 3) SPK_CF_CASE(128, 4)
  SPK_CF_CASE(64, 1) SPK_CF_CASE(64, 2) SPK_CF_CASE(64, 3) SPK_CF_CASE(64, 4)
#undef SPK_CF_CASE
  spk_set_error("%s: internal dispatch error", who);
  return SPK_ERR_ARG;
}

__global__ void k_gather_rows3(const float* __restrict__ r, const int32_t* __restrict__ perm, int64_t E, float* __restrict__ out) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < E; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = perm[k];
    out[3 * k] = r[3 * e]; out[3 * k + 1] = r[3 * e + 1]; out[3 * k + 2] = r[3 * e + 2];
  }
}

static long long* g_cf_dbg = nullptr;
static long long* spk_cf_debug_buffer() { return g_cf_dbg; }
// tuning aid: device buffer of >= 32 int64 that receives cycle-counter stamps of wave 0 / workgroup 0
extern "C" void spk_cfconv_set_debug_buffer(void* p) { g_cf_dbg = (long long*)p; }

// ---- per-call compaction of the pair list: canonical pairs with d < cutoff, order preserved -------------
#include <cstring>
#include <rocprim/rocprim.hpp>
__global__ void k_pair_live(const int32_t* __restrict__ half, const float* __restrict__ rij, int64_t n_half, float cutoff,
                            unsigned char* __restrict__ flags) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_half; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = half[k];
    const float x = rij[3 * e], y = rij[3 * e + 1], z = rij[3 * e + 2];
    flags[k] = sqrtf(x * x + y * y + z * z) < cutoff ? 1 : 0;
  }
}
static size_t active_tmp_bytes(int64_t n_half) {
  // The size query walks rocPRIM's device / config detection (tens of ms): ask once per power-of-two bucket
  // and use the bucket's (larger) requirement for every size inside it.
  static size_t cached[64] = {0};
  int b = 0;
  while (((int64_t)1 << b) < n_half && b < 62) ++b;
  if (cached[b] == 0) {
    size_t bytes = 0;
    int32_t* p = nullptr; unsigned char* f = nullptr;
    (void)rocprim::select(nullptr, bytes, p, f, p, p, (size_t)1 << b, (hipStream_t)0);
    cached[b] = bytes > 0 ? bytes : 1;
  }
  return cached[b];
}
// workspace floats: active list [n_half] + count [4] + flags [n_half bytes] + rocprim temporary
int64_t spk_active_pairs_floats(int64_t n_half) {
  if (n_half <= 0) return 0;
  return n_half + 4 + (n_half + 3) / 4 + 4 + (int64_t)((active_tmp_bytes(n_half) + 3) / 4) + 64;
}
int spk_active_pairs_internal(const spk_graph_t* g, const float* r_ij, float cutoff, float* ws, hipStream_t stream,
                              const int32_t** half_out, const int32_t** count_out) {
  const int64_t n = g->n_half;
  int32_t* act = (int32_t*)ws;
  int32_t* cnt = act + n;                                   // 4 words
  unsigned char* flags = (unsigned char*)(cnt + 4);
  void* tmp = (void*)(((uintptr_t)(flags + n) + 255) & ~(uintptr_t)255);
  size_t tmp_bytes = active_tmp_bytes(n);
  hipLaunchKernelGGL(k_pair_live, dim3(spk_grid_for(n, 256, spk_num_cus() * 8)), dim3(256), 0, stream, g->half, r_ij, n, cutoff, flags);
  SPK_LAUNCH_CHECK();
  SPK_HIP_TRY(rocprim::select(tmp, tmp_bytes, g->half, flags, act, cnt, (size_t)n, stream));
  *half_out = act; *count_out = cnt;
  return SPK_OK;
}

// floats of filter save space per interaction if the pair kernel will run for this graph/shape, else 0
int64_t spk_cfconv_gsave_floats(const spk_graph_t* g, const spk_radial_t* rb, int nf) {
  const int variant = spk_get_variant();
  const int kpb = (rb->n_rbf + 7) / 8;
  const bool mfma_ok = (nf == 128 || nf == 64) && kpb >= 1 && kpb <= 4;
  const bool pair = g->symmetric && g->sorted && g->half && g->rev && g->n_half > 0;
  if (!mfma_ok || !pair || variant == SPK_VARIANT_SIMPLE || variant == SPK_VARIANT_MFMA_DIRECTED) return 0;
  int64_t tiles = (g->n_half + 31) / 32;
  if (variant == SPK_VARIANT_MFMA_MOL && g->n_groups > 0 && g->n_tiles_grouped > tiles) tiles = g->n_tiles_grouped;  // group-aligned tiling
  return tiles * 32 * (int64_t)nf;
}

// EXPERIMENT (spk_tabfilter.hip): layers with a registered filter table run the table-driven row kernels
bool spk_filter_table_lookup(const float* key, const float** table, int* n_knots, float* d_max);
extern "C" int spk_cfconv_tab_f32(const spk_graph_t* g, const float* r_ij, const float* h, const float* table, int32_t n_knots, float d_max,
                                  float cutoff, int32_t nf, float* y, void* stream);
int spk_cfconv_tab_bwd_internal(const spk_graph_t* g, const float* r_ij, const float* h, const float* gy, const float* table, int n_knots, float d_max,
                                float cutoff, float* gh, float* gr, bool gr_assign, hipStream_t stream);

int spk_cfconv_fwd_internal(const spk_graph_t* g, const spk_radial_t* rb, const float* h,
                            const float* r_ij, const float* w1, const float* b1, const float* w2,
                            const float* b2, int nf, float* y, hipStream_t stream, bool pre_zeroed,
                            float* gsave) {
  const char* who = "spk_schnet_cfconv_fwd_f32";
  {
    const float* tab; int nk; float dmax;
    if (g && rb && nf == 128 && g->n_atoms > 0 && g->sorted && g->rowptr && h && r_ij && y && spk_filter_table_lookup(w2, &tab, &nk, &dmax))
      return spk_cfconv_tab_f32(g, r_ij, h, tab, nk, dmax, rb->cutoff, nf, y, stream);
  }
  int rc = check_graph(g, who);
  if (rc) return rc;
  SPK_CHECK_ARG(rb && rb->n_rbf >= 1 && rb->n_rbf <= 256, "%s: bad radial basis", who);
  SPK_CHECK_ARG(nf >= 1 && nf % 4 == 0, "%s: n_filters=%d must be a multiple of 4", who, nf);
  if (g->n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(y != nullptr, "%s: null output", who);
  if (!pre_zeroed) { int _zr = spk_zero_async(y, (size_t)g->n_atoms * nf * sizeof(float), stream); if (_zr) return _zr; }
  if (g->n_edges == 0) return SPK_OK;
  SPK_CHECK_ARG(h && r_ij && w1 && b1 && w2 && b2, "%s: null pointer", who);
  CfArgs a;
  a.h = h; a.gy = nullptr; a.rij = r_ij; a.idx_i = g->idx_i; a.idx_j = g->idx_j;
  a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.y = y; a.gr = nullptr; a.gr_assign = 0; a.skip_gh = 0; a.gsave = gsave; a.gload = nullptr; a.dbg = spk_cf_debug_buffer();
  a.E = g->n_edges; a.N = g->n_atoms; a.rb = spk_radial_dev(rb);
  a.half = g->half; a.rev = g->rev; a.n_half = g->n_half; a.n_half_dev = g->n_half_dev;
  a.grp_atom0 = g->grp_atom0; a.grp_pair0 = g->grp_pair0; a.grp_tile0 = g->grp_tile0; a.n_groups = g->n_groups; a.max_group_atoms = g->max_group_atoms;
  a.rowptr = g->sorted ? g->rowptr : nullptr; a.edge_pair = (g->sorted && g->symmetric) ? g->edge_pair : nullptr;
  return cfconv_dispatch<false>(a, nf, g->symmetric && g->sorted, stream, who);
}

int spk_cfconv_bwd_internal(const spk_graph_t* g, const spk_radial_t* rb, const float* h,
                            const float* gy, const float* r_ij, const float* w1, const float* b1,
                            const float* w2, const float* b2, int nf, float* gh, float* gr,
                            hipStream_t stream, bool pre_zeroed, const float* gload, bool gr_assign, bool want_gh) {
  const char* who = "spk_schnet_cfconv_bwd_f32";
  {
    const float* tab; int nk; float dmax;
    if (g && rb && nf == 128 && g->n_atoms > 0 && g->n_edges > 0 && g->sorted && g->symmetric && g->rowptr && h && gy && r_ij && gr &&
        spk_filter_table_lookup(w2, &tab, &nk, &dmax))
      return spk_cfconv_tab_bwd_internal(g, r_ij, h, gy, tab, nk, dmax, rb->cutoff, want_gh ? gh : nullptr, gr, gr_assign, stream);
  }
  int rc = check_graph(g, who);
  if (rc) return rc;
  SPK_CHECK_ARG(rb && rb->n_rbf >= 1 && rb->n_rbf <= 256, "%s: bad radial basis", who);
  SPK_CHECK_ARG(nf >= 1 && nf % 4 == 0, "%s: n_filters=%d must be a multiple of 4", who, nf);
  if (g->n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(gh != nullptr, "%s: null output", who);
  if (!pre_zeroed) { int _zr = spk_zero_async(gh, (size_t)g->n_atoms * nf * sizeof(float), stream); if (_zr) return _zr; }
  if (g->n_edges == 0) return SPK_OK;
  SPK_CHECK_ARG(h && gy && r_ij && w1 && b1 && w2 && b2 && gr, "%s: null pointer", who);
  CfArgs a;
  a.h = h; a.gy = gy; a.rij = r_ij; a.idx_i = g->idx_i; a.idx_j = g->idx_j;
  a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.y = gh; a.gr = gr; a.skip_gh = want_gh ? 0 : 1; a.gr_assign = (gr_assign && !g->n_half_dev && spk_cfconv_gsave_floats(g, rb, nf) > 0) ? 1 : 0; a.gsave = nullptr; a.gload = gload; a.dbg = spk_cf_debug_buffer();
  a.E = g->n_edges; a.N = g->n_atoms; a.rb = spk_radial_dev(rb);
  a.half = g->half; a.rev = g->rev; a.n_half = g->n_half; a.n_half_dev = g->n_half_dev;
  a.grp_atom0 = g->grp_atom0; a.grp_pair0 = g->grp_pair0; a.grp_tile0 = g->grp_tile0; a.n_groups = g->n_groups; a.max_group_atoms = g->max_group_atoms;
  a.rowptr = nullptr; a.edge_pair = nullptr;
  // the row-local transposed reduction needs idx_i sorted AND a symmetric list
  const bool sym = g->symmetric && g->sorted;
  // Asymmetric list with its by-neighbour copy (spk_transposed_t): gh[j] = sum_{e: idx_j[e] = j} gy[i(e)] W_e is the FORWARD pass over
  // the transposed list (the filter depends on |r_e| only) -- a row pass with one float atomic per run and channel instead of one per
  // pair and channel (measured at N = 16 384, k = 32: 1 773 us per launch with the atomics against 226 us of a forward) -- and the
  // geometry gradient is the directed backward over the original list with the scatter switched off.
  if (!sym && g->transposed && g->transposed->r_perm && spk_get_variant() == SPK_VARIANT_AUTO && (nf == 128 || nf == 64) &&
      (rb->n_rbf + 7) / 8 <= 4 && !getenv("SPK_NO_TRANSPOSED")) {
    const spk_transposed_t* T = g->transposed;
    if (want_gh) {
      hipLaunchKernelGGL(k_gather_rows3, dim3(spk_grid_for(a.E, 256, spk_num_cus() * 8)), dim3(256), 0, stream, r_ij, T->perm, a.E, T->r_perm);
      SPK_LAUNCH_CHECK();
      CfArgs f = a;
      f.h = gy; f.gy = nullptr; f.rij = T->r_perm; f.idx_i = T->idx_i; f.idx_j = T->idx_j; f.y = gh; f.gr = nullptr; f.gr_assign = 0; f.skip_gh = 0;
      f.gsave = nullptr; f.gload = nullptr; f.half = nullptr; f.rev = nullptr; f.n_half = 0; f.n_half_dev = nullptr; f.n_groups = 0;
      f.rowptr = T->rowptr; f.edge_pair = nullptr;      // (large lists: the row-tile forward over the by-neighbour rows)
      int rc2 = cfconv_dispatch<false>(f, nf, false, stream, who);
      if (rc2) return rc2;
    }
    a.skip_gh = 1;
    return cfconv_dispatch<true>(a, nf, false, stream, who);
  }
  return cfconv_dispatch<true>(a, nf, sym, stream, who);
}

extern "C" int spk_schnet_cfconv_fwd_f32(const spk_graph_t* g, const spk_radial_t* rb,
                                         const float* h, const float* r_ij, const float* w1,
                                         const float* b1, const float* w2, const float* b2,
                                         int32_t nf, float* y, void* stream) {
  return spk_cfconv_fwd_internal(g, rb, h, r_ij, w1, b1, w2, b2, nf, y, (hipStream_t)stream, false, nullptr);
}

extern "C" int spk_schnet_cfconv_bwd_f32(const spk_graph_t* g, const spk_radial_t* rb,
                                         const float* h, const float* gy, const float* r_ij,
                                         const float* w1, const float* b1, const float* w2,
                                         const float* b2, int32_t nf, float* gh, float* gr,
                                         void* stream) {
  return spk_cfconv_bwd_internal(g, rb, h, gy, r_ij, w1, b1, w2, b2, nf, gh, gr, (hipStream_t)stream, false, nullptr, false, true);
}
