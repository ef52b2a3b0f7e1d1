// PaiNN (representation/painn.py) for gfx950: fused equivariant message (+ first-order backward),
// the elementwise parts of the mixing block, and whole-representation drivers.
//
// Message kernels are HBM/L2-bound gathers, so they are organised around the centre atom:
// one wavefront owns one CSR row, lanes own features (VPL = F/64 consecutive channels per lane =>
// every neighbour row is read as one coalesced 4F-byte burst), the per-atom sums stay in registers
// (no atomics, no cross-lane reduction) and are written once.  The filter slice
// Phi_e = (phi(d_e) Wf^T + bf) fcut(d_e) is recomputed per edge from register-resident weights
// (n_rbf <= NRBF FMAs per channel); the reference's [E, 3F n_int] filter tensor never exists.
#include "spk_common.h"

int spk_dense_internal(const float* in, const float* pre_in, const float* w, const float* b,
                       const float* res, float* out, float* pre_out, int64_t M, int KC, int NW,
                       int act, bool trans, int pro, hipStream_t stream);

#include "spk_painn_msg.h"
#include "spk_painn_blk.h"

// ------------------------------------------------------------------------------------------
// row kernels (sorted idx_i; backward additionally needs a symmetric list)
// ------------------------------------------------------------------------------------------
// The VPL (1 or 2) consecutive channels of a lane are one value of type VT: float, or a 2-vector so that the
// filter recomputation and the message algebra compile to packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32:
// two channels per VALU lane and cycle -- the fp32 vector peak of the CU assumes them; the scalar form of the
// backward issued 445 v_fma + 98 v_pk_fma per edge).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Sum over the 64 lanes without the LDS crossbar: four DPP adds give every lane the sum of its row of 16 (VALU latency
// instead of four dependent ds_bpermute round trips), the four row sums are then read from lanes 15 / 31 / 47 / 63.  The
// result is wave-uniform.  (The backward row kernel needs four such sums per edge; with __shfl_xor they were its longest
// dependency chain.)
__device__ __forceinline__ float spk_row16_sum_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}
__device__ __forceinline__ float spk_wave_sum_dpp(float v) {
  v = spk_row16_sum_dpp(v);
  return (spk_readlane_f(v, 15) + spk_readlane_f(v, 31)) + (spk_readlane_f(v, 47) + spk_readlane_f(v, 63));
}

template <int VPL> struct MsgVec;
template <> struct MsgVec<1> {
  typedef float T;
  static __device__ __forceinline__ T load(const float* p) { return p[0]; }
  static __device__ __forceinline__ void store(float* p, T v) { p[0] = v; }
  static __device__ __forceinline__ float sum(T v) { return v; }
  static __device__ __forceinline__ T zero() { return 0.f; }
};
template <> struct MsgVec<2> {
  typedef f32x2 T;
  static __device__ __forceinline__ T load(const float* p) { return *(const f32x2*)p; }
  static __device__ __forceinline__ void store(float* p, T v) { *(f32x2*)p = v; }
  static __device__ __forceinline__ float sum(T v) { return v.x + v.y; }
  static __device__ __forceinline__ T zero() { return f32x2{0.f, 0.f}; }
};

// GEOM (backward only): form the geometry gradient gr alone -- no neighbour gradients are gathered, no gc / gmu
// MU0: mu == 0 everywhere (first interaction): the mu rows of the neighbours are not gathered
// TAB (experiment, opt-in): the raw filter and its slope come from a cubic-Hermite table (one 16-byte read per part and channel: knot
// value, slope, difference to the next knot, next slope;
// knot for the lane's two channels) instead of NRBF FMAs per channel from register-resident weights (which are then not loaded)
// LT (round 4; F = 128, n_rbf <= 20): the radial values phi_k(d), phi_k'(d) of the 64 edges of a chunk are evaluated with lanes = EDGES
// (n_rbf full-precision evaluations per lane and chunk instead of one per lane and EDGE) into a wave-private LDS table and read back
// per edge as 5 + 5 broadcast ds_read_b128 -- instead of one evaluation by lane k and 2 n_rbf v_readlane broadcasts per edge.  The
// SQ counters said this kernel is VALU-issue bound (330 instructions per edge and wavefront in the backward, 66 % busy): the
// broadcasts and the per-edge evaluation were 70 of them, and LDS reads issue beside the VALU.
// TS (backward only; round 5): the TRANSPOSED SUMS alone -- gc and the new gmu of the row atom, no geometry gradient: the slope of the filter is
// not formed and the neighbours' c / mu rows are not gathered.  Run over the by-neighbour copy of an asymmetric list (spk_transposed_t; rows =
// neighbour atoms j, entries = the pairs that point at j, pair vectors negated) it is the scatter over idx_j of painn.py:54-66's backward as a
// row pass: no atomics, fixed order.  The geometry gradient of such a list is the GEOM form over the list itself.
template <int VPL, int NRBF, bool BWD, bool GEOM = false, bool MU0 = false, bool TAB = false, bool LT = false, bool TS = false>
__global__ __launch_bounds__(256) void k_painn_msg_row(MsgArgs a) {
  typedef MsgVec<VPL> MV;
  typedef typename MV::T VT;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int F = a.F;
  const int K = a.rb.n_rbf;
  // register-resident filter weights of this lane's channels: part p in (q, R, mu).  The slice
  // [3F, K] is read once per workgroup with coalesced loads into a transposed, padded LDS image
  // (conflict-free writes), from which every lane picks its rows.
  extern __shared__ __attribute__((aligned(16))) float swf[];   // [K][3F + 1]
  if (!TAB) {
    const int ld = 3 * F + 1;
    for (int s = threadIdx.x; s < 3 * F * K; s += 256) {
      const int row = s / K, k = s - row * K;
      swf[k * ld + row] = a.wf[s];
    }
    __syncthreads();
  }
  VT w[3][TAB ? 1 : NRBF];
  VT bias[3];
#pragma unroll
  for (int p = 0; p < 3 && !TAB; ++p) {
    float tb[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) tb[v] = a.bf[p * F + VPL * lane + v];
    bias[p] = MV::load(tb);
#pragma unroll
    for (int k = 0; k < NRBF; ++k) {
      float tw[VPL];
#pragma unroll
      for (int v = 0; v < VPL; ++v) tw[v] = (k < K) ? swf[k * (3 * F + 1) + p * F + VPL * lane + v] : 0.f;
      w[p][k] = MV::load(tw);
    }
  }
  float* tabw = swf + wv * (64 * 2 * NRBF);      // LT: this wave's [64 edges][phi_0.. | phi'_0..] table (aliases the weight image, which is dead now)
  if (LT) __syncthreads();

  // Workgroups go round-robin over the 8 XCDs: with xcd_map the workgroups of one XCD walk a CONTIGUOUS eighth of the atoms (their
  // neighbours' rows are then shared inside that XCD's L2) instead of every 8th group of four (spk_painn_tile.hip measured +8 % for
  // the tile forward on the water box with the same walk)
  const int64_t per_xcd = a.xcd_map ? (a.N + 7) / 8 : a.N;
  const int64_t a_lo = a.xcd_map ? (int64_t)(blockIdx.x & 7) * per_xcd : 0;
  const int64_t a_hi = a.xcd_map ? (a_lo + per_xcd < a.N ? a_lo + per_xcd : a.N) : a.N;
  const int64_t a_first = a.xcd_map ? a_lo + (int64_t)(blockIdx.x >> 3) * 4 + wv : (int64_t)blockIdx.x * 4 + wv;
  const int64_t a_step = a.xcd_map ? (int64_t)((gridDim.x + 7) >> 3) * 4 : (int64_t)gridDim.x * 4;
  for (int64_t atom = a_first; atom < a_hi; atom += a_step) {
    const int32_t e0 = a.rowptr[atom], e1 = a.rowptr[atom + 1];
    const int64_t fo = (int64_t)VPL * lane;  // first channel of this lane
    VT accq = MV::zero(), accR = MV::zero();                                   // fwd: dq ; bwd: gc_q, gc_R
    VT accv[3] = {MV::zero(), MV::zero(), MV::zero()};                        // fwd: dmu ; bwd: S
    // values at the centre atom needed by the backward
    VT gqa = MV::zero(), gma[3] = {MV::zero(), MV::zero(), MV::zero()};
    if (BWD) {
      gqa = MV::load(a.gq_out + atom * F + fo);
#pragma unroll
      for (int x = 0; x < 3; ++x) gma[x] = MV::load(a.gmu_out + (atom * 3 + x) * F + fo);
    }
    for (int32_t cs = e0; cs < e1; cs += 64) {
      // lanes = edges: geometry of up to 64 edges of this row at once
      const int32_t em = cs + lane;
      const bool ev = em < e1;
      const int32_t emc = ev ? em : (e1 - 1);
      const int jl = (int)a.idx_j[emc];
      const float rx = a.rij[3 * (int64_t)emc], ry = a.rij[3 * (int64_t)emc + 1], rz = a.rij[3 * (int64_t)emc + 2];
      const float dl = sqrtf(rx * rx + ry * ry + rz * rz);
      const float inv = 1.0f / dl;
      const float uxl = rx * inv, uyl = ry * inv, uzl = rz * inv;
      float fcl, dfcl;
      spk_cutoff_eval(a.rb.cutoff, dl, fcl, dfcl);
      float grx = 0.f, gry = 0.f, grz = 0.f;  // bwd: geometry gradient of this lane's edge
      if (LT) {
        spk_wave_lds_sync();      // the previous chunk's reads are done
#pragma unroll
        for (int k4 = 0; k4 < NRBF / 4; ++k4) {
          f32x4 pv, dv;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float p_, dp_;
            spk_rbf_eval(a.rb, 4 * k4 + q, dl, p_, dp_);        // 0 for k >= n_rbf
            pv[q] = p_; dv[q] = dp_;
          }
          *(f32x4*)(tabw + lane * (2 * NRBF) + 4 * k4) = pv;
          if (BWD && !TS) *(f32x4*)(tabw + lane * (2 * NRBF) + NRBF + 4 * k4) = dv;
        }
        spk_wave_lds_sync();
      }
      // neighbour rows are requested one edge ahead (explicit double buffer): the row kernel is
      // bound by the latency of these dependent gathers, not by their bandwidth
      constexpr bool PF = true;
      VT cjr[PF ? 2 : 1][3] = {}, mujr[PF ? 2 : 1][3] = {}, gqbr[PF ? 2 : 1] = {}, gmbr[PF ? 2 : 1][3] = {};
      auto load_rows = [&](int slot, int t) {
        const int64_t jj = __builtin_amdgcn_readlane(jl, t);
        const float* cj = a.c + jj * 3 * F + fo;
        const float* muj = a.mu + jj * 3 * F + fo;
#pragma unroll
        for (int p = 0; p < 3 && !TS; ++p) { cjr[slot][p] = MV::load(cj + p * F); if (!MU0) mujr[slot][p] = MV::load(muj + p * F); }
        if (BWD && !GEOM) {
          gqbr[slot] = MV::load(a.gq_out + jj * F + fo);
#pragma unroll
          for (int p = 0; p < 3; ++p) gmbr[slot][p] = MV::load(a.gmu_out + jj * 3 * F + fo + p * F);
        }
      };
      // Pairs at or beyond the cutoff (skin / buffer pairs of MD lists) contribute exactly zero to every sum
      // and gradient (f_c = f_c' = 0): they are dropped here, before their rows are fetched.  The row's live
      // edges are walked through a bit mask (wave-uniform), so the one-edge-ahead prefetch stays intact.
      uint64_t live = __ballot(ev && dl < a.rb.cutoff);
      const int n_live = __popcll(live);
      auto next_live = [&]() { const int t = __ffsll((long long)live) - 1; live &= live - 1; return t; };
      int t_next = n_live > 0 ? next_live() : 0;
      if (PF && n_live > 0) load_rows(0, t_next);
      for (int t2 = 0; t2 < n_live; t2 += 2) {
#pragma unroll
        for (int par0 = 0; par0 < 2; ++par0) {
          const int par = PF ? par0 : 0;
          if (t2 + par0 < n_live) {
            const int t = t_next;
            if (t2 + par0 + 1 < n_live) t_next = next_live();
            if (PF) { if (t2 + par0 + 1 < n_live) load_rows(par ^ 1, t_next); }
            else load_rows(0, t);
            const float d = spk_readlane_f(dl, t);
            const float ux = spk_readlane_f(uxl, t), uy = spk_readlane_f(uyl, t), uz = spk_readlane_f(uzl, t);
            const float fc = spk_readlane_f(fcl, t), dfc = spk_readlane_f(dfcl, t);
            VT P[3], Pd[3] = {MV::zero(), MV::zero(), MV::zero()};
            if (TAB) {
              // raw filter and slope of the lane's channels from the table: interval n of [n_knots][3F][4]
              const float u = d * a.tab_inv_step;
              int n = (int)u;
              n = n < a.tab_knots - 2 ? n : a.tab_knots - 2;
              const float sx = u - (float)n, s2 = sx * sx, s3 = s2 * sx;
              const float h10 = s3 - 2.f * s2 + sx, h01 = -2.f * s3 + 3.f * s2, h11 = s3 - s2;
              const float g10 = (3.f * s2 - 4.f * sx + 1.f) * a.tab_inv_step, g01 = (6.f * sx - 6.f * s2) * a.tab_inv_step,
                          g11 = (3.f * s2 - 2.f * sx) * a.tab_inv_step;
              const float* t0 = a.tab + ((size_t)n * 3 * F + fo) * 4;
#pragma unroll
              for (int p = 0; p < 3; ++p) {
                if (MU0 && p == 2) { P[p] = MV::zero(); continue; }
                float pv[VPL], pdv[VPL];
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                  const f32x4 kk = *(const f32x4*)(t0 + (size_t)(p * F + v) * 4);          // (v_n, m_n, v_n+1 - v_n, m_n+1)
                  pv[v] = kk.x + h10 * kk.y + h01 * kk.z + h11 * kk.w;
                  pdv[v] = g10 * kk.y + g01 * kk.z + g11 * kk.w;
                }
                P[p] = MV::load(pv);
                if (BWD && !TS) Pd[p] = MV::load(pdv);
              }
            } else if (LT) {
              const float* te = tabw + t * (2 * NRBF);          // wave-uniform address: broadcast reads
              P[0] = bias[0]; P[1] = bias[1]; P[2] = bias[2];
#pragma unroll
              for (int k4 = 0; k4 < NRBF / 4; ++k4) {
                const f32x4 s4 = *(const f32x4*)(te + 4 * k4);
                f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                if (BWD && !TS) d4 = *(const f32x4*)(te + NRBF + 4 * k4);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                  for (int p = 0; p < 3; ++p) {
                    P[p] = w[p][4 * k4 + q] * s4[q] + P[p];
                    if (BWD && !TS) Pd[p] = w[p][4 * k4 + q] * d4[q] + Pd[p];
                  }
              }
            } else {
            // lane k evaluates phi_k(d)
            float pl, dpl;
            spk_rbf_eval(a.rb, lane, d, pl, dpl);
            P[0] = bias[0]; P[1] = bias[1]; P[2] = bias[2];
#pragma unroll
            for (int k = 0; k < (TAB ? 1 : NRBF); ++k) {
              const float s = spk_readlane_f(pl, k);
              const float sd = (BWD && !TS) ? spk_readlane_f(dpl, k) : 0.f;
#pragma unroll
              for (int p = 0; p < 3; ++p) {
                P[p] = w[p][k] * s + P[p];
                if (BWD && !TS) Pd[p] = w[p][k] * sd + Pd[p];
              }
            }
            }
            if (!BWD) {
              const VT mq = P[0] * fc * cjr[par][0];
              const VT mR = P[1] * fc * cjr[par][1];
              const VT mm = P[2] * fc * cjr[par][2];
              accq += mq;
              accv[0] += mR * ux + mm * mujr[par][0];
              accv[1] += mR * uy + mm * mujr[par][1];
              accv[2] += mR * uz + mm * mujr[par][2];
            } else {
              const VT Fq = P[0] * fc, FR = P[1] * fc, Fm = P[2] * fc;
              const VT dFq = Pd[0] * fc + P[0] * dfc;
              const VT dFR = Pd[1] * fc + P[1] * dfc;
              const VT dFm = Pd[2] * fc + P[2] * dfc;
              const VT cq = cjr[par][0], cR = cjr[par][1], cm = cjr[par][2];
              const VT mb0 = mujr[par][0], mb1 = mujr[par][1], mb2 = mujr[par][2];
              const VT gb0 = gmbr[par][0], gb1 = gmbr[par][1], gb2 = gmbr[par][2];
              // (1) transposed sums for the centre atom acting as neighbour of b (reverse edge)
              if (!GEOM) {
                accq += Fq * gqbr[par];
                accR -= FR * (gb0 * ux + gb1 * uy + gb2 * uz);
                accv[0] += Fm * gb0; accv[1] += Fm * gb1; accv[2] += Fm * gb2;
              }
              // (2) geometry gradient of edge (atom <- b)
              if (!TS) {
              const VT gu = gma[0] * ux + gma[1] * uy + gma[2] * uz;
              const VT gm = gma[0] * mb0 + gma[1] * mb1 + gma[2] * mb2;
              const VT ddv = cq * gqa * dFq + cR * gu * dFR + cm * gm * dFm;
              const VT mR = FR * cR;
              float dd = MV::sum(ddv), tux = MV::sum(gma[0] * mR), tuy = MV::sum(gma[1] * mR), tuz = MV::sum(gma[2] * mR);
              dd = spk_wave_sum_dpp(dd); tux = spk_wave_sum_dpp(tux); tuy = spk_wave_sum_dpp(tuy); tuz = spk_wave_sum_dpp(tuz);
              if (lane == t) {
                const float dot = tux * ux + tuy * uy + tuz * uz;
                const float invd = 1.0f / d;
                grx = dd * ux + (tux - dot * ux) * invd;
                gry = dd * uy + (tuy - dot * uy) * invd;
                grz = dd * uz + (tuz - dot * uz) * invd;
              }
              }
            }
          }
        }
      }
      if (BWD && !TS && ev && dl > 0.f) {
        a.gr[3 * (int64_t)em] += grx; a.gr[3 * (int64_t)em + 1] += gry; a.gr[3 * (int64_t)em + 2] += grz;
      }
    }
    if (!BWD) {
      MV::store(a.q_out + atom * F + fo, MV::load(a.q + atom * F + fo) + accq);
#pragma unroll
      for (int x = 0; x < 3; ++x)
        MV::store(a.mu_out + (atom * 3 + x) * F + fo, MV::load(a.mu + (atom * 3 + x) * F + fo) + accv[x]);
    } else if (!GEOM) {
      const VT ma0 = MV::load(a.mu + (atom * 3 + 0) * F + fo), ma1 = MV::load(a.mu + (atom * 3 + 1) * F + fo),
               ma2 = MV::load(a.mu + (atom * 3 + 2) * F + fo);
      const VT cma = MV::load(a.c + atom * 3 * F + 2 * F + fo);
      MV::store(a.gc + atom * 3 * F + fo, accq);
      MV::store(a.gc + atom * 3 * F + F + fo, accR);
      MV::store(a.gc + atom * 3 * F + 2 * F + fo, ma0 * accv[0] + ma1 * accv[1] + ma2 * accv[2]);
#pragma unroll
      for (int x = 0; x < 3; ++x) MV::store(a.gmu + (atom * 3 + x) * F + fo, gma[x] + cma * accv[x]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// simple kernels: one workgroup per edge, one thread per channel, float atomics; any F / n_rbf,
// any edge order, no symmetry assumption.  Outputs must be pre-initialised by the launcher
// (fwd: q_out = q, mu_out = mu; bwd: gc = 0, gmu = gmu_out).
// ------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void k_painn_msg_simple(MsgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.rb.n_rbf, F = a.F;
  float* sphi = sm;
  float* sdphi = sm + K;
  float* sred = sdphi + K;  // 4 * (blockDim/64)
  const int f = threadIdx.x;
  const int nw = blockDim.x >> 6;
  for (int64_t e = blockIdx.x; e < a.E; e += gridDim.x) {
    const int64_t i = a.idx_i[e], j = a.idx_j[e];
    const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    const float inv = 1.0f / d;
    const float ux = rx * inv, uy = ry * inv, uz = rz * inv;
    float fc, dfc;
    spk_cutoff_eval(a.rb.cutoff, d, fc, dfc);
    __syncthreads();
    if (f < K) { float p, dp; spk_rbf_eval(a.rb, f, d, p, dp); sphi[f] = p; sdphi[f] = dp; }
    __syncthreads();
    float dd = 0.f, tux = 0.f, tuy = 0.f, tuz = 0.f;
    if (f < F) {
      float P[3], Pd[3];
      for (int p = 0; p < 3; ++p) {
        const int row = p * F + f;
        float s = a.bf[row], sd = 0.f;
        for (int k = 0; k < K; ++k) {
          const float wv = a.wf[(int64_t)row * K + k];
          s = fmaf(wv, sphi[k], s);
          sd = fmaf(wv, sdphi[k], sd);
        }
        P[p] = s; Pd[p] = sd;
      }
      const float cq = a.c[j * 3 * F + f], cR = a.c[j * 3 * F + F + f], cm = a.c[j * 3 * F + 2 * F + f];
      const float mb0 = a.mu[(j * 3 + 0) * F + f], mb1 = a.mu[(j * 3 + 1) * F + f], mb2 = a.mu[(j * 3 + 2) * F + f];
      const float Fq = P[0] * fc, FR = P[1] * fc, Fm = P[2] * fc;
      if (!BWD) {
        const float mq = Fq * cq, mR = FR * cR, mm = Fm * cm;
        unsafeAtomicAdd(&a.q_out[i * F + f], mq);
        unsafeAtomicAdd(&a.mu_out[(i * 3 + 0) * F + f], mR * ux + mm * mb0);
        unsafeAtomicAdd(&a.mu_out[(i * 3 + 1) * F + f], mR * uy + mm * mb1);
        unsafeAtomicAdd(&a.mu_out[(i * 3 + 2) * F + f], mR * uz + mm * mb2);
      } else {
        const float gq = a.gq_out[i * F + f];
        const float g0 = a.gmu_out[(i * 3 + 0) * F + f], g1 = a.gmu_out[(i * 3 + 1) * F + f], g2 = a.gmu_out[(i * 3 + 2) * F + f];
        const float gu = g0 * ux + g1 * uy + g2 * uz;
        const float gm = g0 * mb0 + g1 * mb1 + g2 * mb2;
        // gc[j] += Phi * mbar ;  gmu[j] += m_mu * gmu_out[i]
        unsafeAtomicAdd(&a.gc[j * 3 * F + f], Fq * gq);
        unsafeAtomicAdd(&a.gc[j * 3 * F + F + f], FR * gu);
        unsafeAtomicAdd(&a.gc[j * 3 * F + 2 * F + f], Fm * gm);
        const float mm = Fm * cm;
        unsafeAtomicAdd(&a.gmu[(j * 3 + 0) * F + f], mm * g0);
        unsafeAtomicAdd(&a.gmu[(j * 3 + 1) * F + f], mm * g1);
        unsafeAtomicAdd(&a.gmu[(j * 3 + 2) * F + f], mm * g2);
        const float dFq = Pd[0] * fc + P[0] * dfc, dFR = Pd[1] * fc + P[1] * dfc, dFm = Pd[2] * fc + P[2] * dfc;
        dd = cq * gq * dFq + cR * gu * dFR + cm * gm * dFm;
        const float mR = FR * cR;
        tux = g0 * mR; tuy = g1 * mR; tuz = g2 * mR;
      }
    }
    if (BWD) {
      dd = spk_wave_sum(dd); tux = spk_wave_sum(tux); tuy = spk_wave_sum(tuy); tuz = spk_wave_sum(tuz);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) {
        const int wq = threadIdx.x >> 6;
        sred[4 * wq] = dd; sred[4 * wq + 1] = tux; sred[4 * wq + 2] = tuy; sred[4 * wq + 3] = tuz;
      }
      __syncthreads();
      if (threadIdx.x == 0 && d > 0.f) {
        float D = 0.f, X = 0.f, Y = 0.f, Z = 0.f;
        for (int wq = 0; wq < nw; ++wq) { D += sred[4 * wq]; X += sred[4 * wq + 1]; Y += sred[4 * wq + 2]; Z += sred[4 * wq + 3]; }
        const float dot = X * ux + Y * uy + Z * uz;
        a.gr[3 * e] += D * ux + (X - dot * ux) * inv;
        a.gr[3 * e + 1] += D * uy + (Y - dot * uy) * inv;
        a.gr[3 * e + 2] += D * uz + (Z - dot * uz) * inv;
      }
    }
  }
}

// out[k] = -r[perm[k]]: the pair vectors of the by-neighbour copy of a list, seen from the neighbour (R_i - R_j)
__global__ void k_gather_rows3_neg(const float* __restrict__ r, const int32_t* __restrict__ perm, int64_t E, float* __restrict__ out) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < E; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = perm[k];
    out[3 * k] = -r[3 * e]; out[3 * k + 1] = -r[3 * e + 1]; out[3 * k + 2] = -r[3 * e + 2];
  }
}

// ------------------------------------------------------------------------------------------
// launchers of the message
// ------------------------------------------------------------------------------------------
static int check_msg(const spk_graph_t* g, const spk_radial_t* rb, int F, const char* who) {
  SPK_CHECK_ARG(g != nullptr && rb != nullptr, "%s: null graph/radial", who);
  SPK_CHECK_ARG(g->n_atoms >= 0 && g->n_edges >= 0 && g->n_atoms < (1LL << 31) && g->n_edges < (1LL << 31), "%s: bad graph sizes", who);
  SPK_CHECK_ARG(g->n_edges == 0 || (g->idx_i && g->idx_j), "%s: null index arrays", who);
  SPK_CHECK_ARG(F >= 1 && F <= 1024, "%s: n_atom_basis=%d unsupported", who, F);
  SPK_CHECK_ARG(rb->n_rbf >= 1 && rb->n_rbf <= 256, "%s: n_rbf=%d unsupported", who, rb->n_rbf);
  return SPK_OK;
}

// EXPERIMENT (spk_tabfilter.hip): filter tables attached to an interaction by the address of its filter rows
bool spk_filter_table_lookup(const float* key, const float** table, int* n_knots, float* d_max);

static int g_row_table = -1;     // -1: rule (large lists), 0 / 1: never / whenever the shape has the instance
extern "C" void spk_painn_set_row_table(int32_t mode) { g_row_table = mode; }

template <bool BWD>
static int msg_dispatch(const MsgArgs& a_in, bool row_ok, hipStream_t stream, const char* who) {
  const int variant = spk_get_variant();
  MsgArgs a = a_in;
  const int F = a.F, K = a.rb.n_rbf;
  const bool shape_ok = row_ok && (F == 64 || F == 128) && K <= 32;
  {
    const float* tab; int nk; float dmax;
    if (shape_ok && F == 128 && variant != SPK_VARIANT_SIMPLE && spk_filter_table_lookup(a.wf, &tab, &nk, &dmax)) {
      a.tab = tab; a.tab_knots = nk; a.tab_inv_step = (float)(nk - 1) / dmax;
      const int grid = spk_grid_for(a.N, 4, spk_num_cus() * 4);
      SpkProfScope prof(BWD ? (a.geom_only ? "painn_msg_bwd_tab_geom" : "painn_msg_bwd_tab") : (a.mu_zero ? "painn_msg_fwd_tab_mu0" : "painn_msg_fwd_tab"), stream);
      if (BWD && a.geom_only && a.mu_zero) hipLaunchKernelGGL((k_painn_msg_row<2, 20, BWD, BWD, true, true>), dim3(grid), dim3(256), 0, stream, a);
      else if (BWD && a.geom_only) hipLaunchKernelGGL((k_painn_msg_row<2, 20, BWD, BWD, false, true>), dim3(grid), dim3(256), 0, stream, a);
      else if (!BWD && a.mu_zero) hipLaunchKernelGGL((k_painn_msg_row<2, 20, BWD, false, !BWD, true>), dim3(grid), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((k_painn_msg_row<2, 20, BWD, false, false, true>), dim3(grid), dim3(256), 0, stream, a);
      SPK_LAUNCH_CHECK();
      return SPK_OK;
    }
  }
  SPK_CHECK_ARG(variant != SPK_VARIANT_MFMA || shape_ok, "%s: shape F=%d n_rbf=%d (or unsorted/asymmetric list) not supported by the row kernel", who, F, K);
  if (row_ok && variant != SPK_VARIANT_SIMPLE && spk_painn_blk_ok(a, BWD))       // large lists with a block plan (spk_painn_blk.hip)
    return BWD ? spk_painn_blk_bwd(a, stream) : spk_painn_blk_fwd(a, stream);
  if (!BWD && shape_ok && variant != SPK_VARIANT_SIMPLE && spk_painn_msg_rowtile_fwd_ok(a))
    return spk_painn_msg_rowtile_fwd(a, stream);      // a wavefront per row, split-precision filter GEMM, no atomics (spk_painn_tile.hip, round 6)
  if (!BWD && shape_ok && variant != SPK_VARIANT_SIMPLE && spk_painn_msg_tile_ok(a)) {
    // forward on large lists: filter GEMM on the matrix cores, 32-edge tiles (spk_painn_tile.hip); the profile scope
    // covers the init launch too
    SpkProfScope prof(a.mu_zero ? "painn_msg_fwd_tile_mu0" : "painn_msg_fwd_tile", stream);
    return spk_painn_msg_tile_fwd(a, stream);
  }
  if (BWD && shape_ok && variant != SPK_VARIANT_SIMPLE && spk_painn_msg_rowtile_bwd_ok(a)) {
    // backward with a wavefront per row and the filter GEMMs on the f16 matrix instructions (spk_painn_tile.hip, round 6)
    return spk_painn_msg_rowtile_bwd(a, stream);
  }
  if (BWD && shape_ok && variant != SPK_VARIANT_SIMPLE && spk_painn_msg_tile_bwd_ok(a)) {
    // backward with the value and derivative filter GEMMs on the matrix cores (spk_painn_tile.hip)
    SpkProfScope prof(a.geom_only ? "painn_msg_bwd_tile_geom" : "painn_msg_bwd_tile", stream);
    return spk_painn_msg_tile_bwd(a, stream);
  }
  if (shape_ok && variant != SPK_VARIANT_SIMPLE) {
    // persistent waves: every wave walks several CSR rows, so the per-wave weight set-up is amortised
    a.xcd_map = (spk_xcd_walk_default() && a.N >= (1 << 14)) ? 1 : 0;        // large lists only: a molecule batch fits every L2
    const int grid = a.xcd_map ? (spk_grid_for(a.N, 4, spk_num_cus() * 2) + 7) / 8 * 8 : spk_grid_for(a.N, 4, spk_num_cus() * 2);
    size_t lds = (size_t)K * (3 * F + 1) * sizeof(float);
    // LDS radial table (LT) where it pays: F = 128, n_rbf <= 20, large lists (the instruction-bound regime); SPK_ROW_LT=0 / 1 forces
    static const int lt_env = [] { const char* e = getenv("SPK_ROW_LT"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    const int lt_mode = g_row_table >= 0 ? g_row_table : lt_env;
    // measured on the 32k-atom water box (profiles/r04_tile_experiments.txt): forward 824 -> 643 us, backward 1 423 -> 1 597 us (at
    // 255 VGPRs the backward's schedule loses more than the table saves) => the rule switches it on for the forward only
    const bool lt = F == 128 && K <= 20 && (lt_mode >= 0 ? lt_mode == 1 : (!BWD && a.N >= (1 << 14)));
    if (lt && lds < 4 * 64 * 2 * 20 * sizeof(float)) lds = 4 * 64 * 2 * 20 * sizeof(float);
    // the first-interaction specialisations are timed under their own tags (they move fewer bytes)
    SpkProfScope prof(BWD ? (a.geom_only ? "painn_msg_bwd_row_geom" : "painn_msg_bwd_row") : (a.mu_zero ? "painn_msg_fwd_row_mu0" : "painn_msg_fwd_row"), stream);
#define SPK_MSG_CASE(VPLv, NRBFv)                                                                                             \
  do {                                                                                                                        \
    if (BWD && a.geom_only && a.mu_zero) hipLaunchKernelGGL((k_painn_msg_row<VPLv, NRBFv, BWD, BWD, true>), dim3(grid), dim3(256), lds, stream, a);  \
    else if (BWD && a.geom_only) hipLaunchKernelGGL((k_painn_msg_row<VPLv, NRBFv, BWD, BWD, false>), dim3(grid), dim3(256), lds, stream, a);      \
    else if (!BWD && a.mu_zero) hipLaunchKernelGGL((k_painn_msg_row<VPLv, NRBFv, BWD, false, !BWD>), dim3(grid), dim3(256), lds, stream, a);      \
    else hipLaunchKernelGGL((k_painn_msg_row<VPLv, NRBFv, BWD, false, false>), dim3(grid), dim3(256), lds, stream, a);                           \
  } while (0)
    if (lt) {
      if (BWD && a.geom_only && a.mu_zero) hipLaunchKernelGGL((k_painn_msg_row<2, 20, BWD, BWD, true, false, true>), dim3(grid), dim3(256), lds, stream, a);
      else if (BWD && a.geom_only) hipLaunchKernelGGL((k_painn_msg_row<2, 20, BWD, BWD, false, false, true>), dim3(grid), dim3(256), lds, stream, a);
      else if (!BWD && a.mu_zero) hipLaunchKernelGGL((k_painn_msg_row<2, 20, BWD, false, !BWD, false, true>), dim3(grid), dim3(256), lds, stream, a);
      else hipLaunchKernelGGL((k_painn_msg_row<2, 20, BWD, false, false, false, true>), dim3(grid), dim3(256), lds, stream, a);
    }
    else if (F == 64) { if (K <= 20) SPK_MSG_CASE(1, 20); else SPK_MSG_CASE(1, 32); }
    else { if (K <= 20) SPK_MSG_CASE(2, 20); else SPK_MSG_CASE(2, 32); }
#undef SPK_MSG_CASE
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  // simple path: initialise outputs, then atomics
  const size_t nf = (size_t)a.N * F;
  if (!BWD) {
    SPK_HIP_TRY(hipMemcpyAsync(a.q_out, a.q, nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
    SPK_HIP_TRY(hipMemcpyAsync(a.mu_out, a.mu, 3 * nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
  } else {
    { int _zr = spk_zero_async(a.gc, 3 * nf * sizeof(float), stream); if (_zr) return _zr; }
    SPK_HIP_TRY(hipMemcpyAsync(a.gmu, a.gmu_out, 3 * nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
  }
  if (a.E == 0) return SPK_OK;
  int threads = ((F > K ? F : K) + 63) / 64 * 64;
  const size_t lds = (size_t)(2 * K + 4 * 16 + 8) * sizeof(float);
  int grid = (int)(a.E < 65535 * 16 ? a.E : 65535 * 16);
  SpkProfScope prof(BWD ? "painn_msg_bwd_simple" : "painn_msg_fwd_simple", stream);
  hipLaunchKernelGGL(k_painn_msg_simple<BWD>, dim3(grid), dim3(threads), lds, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

int spk_painn_message_fwd_internal(const spk_graph_t* g, const spk_radial_t* rb, const float* c,
                                   const float* q, const float* mu, const float* r_ij,
                                   const float* wf, const float* bf, int F, float* q_out,
                                   float* mu_out, hipStream_t stream, bool mu_zero = false, bool blocks_prepared = false) {
  const char* who = "spk_painn_message_fwd_f32";
  int rc = check_msg(g, rb, F, who);
  if (rc) return rc;
  if (g->n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(c && q && mu && wf && bf && q_out && mu_out && (g->n_edges == 0 || r_ij), "%s: null pointer", who);
  MsgArgs a = {};
  a.c = c; a.q = q; a.mu = mu; a.rij = r_ij; a.idx_i = g->idx_i; a.idx_j = g->idx_j; a.rowptr = g->rowptr;
  a.wf = wf; a.bf = bf; a.q_out = q_out; a.mu_out = mu_out; a.E = g->n_edges; a.N = g->n_atoms; a.F = F;
  a.rb = spk_radial_dev(rb); a.mu_zero = mu_zero ? 1 : 0; a.skin_list = g->filter_pairs ? 1 : 0;
  a.blocks = g->blocks; a.blocks_prepared = blocks_prepared ? 1 : 0;
  return msg_dispatch<false>(a, g->sorted && g->rowptr, stream, who);
}

int spk_painn_message_bwd_internal(const spk_graph_t* g, const spk_radial_t* rb, const float* c,
                                   const float* mu, const float* gq_out, const float* gmu_out,
                                   const float* r_ij, const float* wf, const float* bf, int F,
                                   float* gc, float* gmu, float* gr, hipStream_t stream, bool geom_only = false, bool mu_zero = false,
                                   bool blocks_prepared = false) {
  const char* who = "spk_painn_message_bwd_f32";
  int rc = check_msg(g, rb, F, who);
  if (rc) return rc;
  if (g->n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(c && mu && gq_out && gmu_out && wf && bf && gc && gmu && (g->n_edges == 0 || (r_ij && gr)), "%s: null pointer", who);
  MsgArgs a = {};
  a.c = c; a.mu = mu; a.gq_out = gq_out; a.gmu_out = gmu_out; a.rij = r_ij; a.idx_i = g->idx_i; a.idx_j = g->idx_j;
  a.rowptr = g->rowptr; a.wf = wf; a.bf = bf; a.gc = gc; a.gmu = gmu; a.gr = gr; a.E = g->n_edges; a.N = g->n_atoms;
  a.F = F; a.rb = spk_radial_dev(rb); a.geom_only = geom_only ? 1 : 0; a.mu_zero = mu_zero ? 1 : 0; a.skin_list = g->filter_pairs ? 1 : 0;
  a.blocks = g->blocks; a.blocks_prepared = blocks_prepared ? 1 : 0;
  // Sorted but ASYMMETRIC list that carries its by-neighbour copy (spk_transposed_t; LAMMPS-ordered and vesin lists,
  // interfaces/lammps/pair_schnetpack.cpp:240-267, transform/neighborlist.py:446-456): two row passes instead of the edge-parallel kernel
  // with its 7 F float atomics per pair -- (1) the transposed sums as the TS row form over the by-neighbour list, (2) the geometry gradient
  // as the GEOM row form over the list itself.  No atomics, fixed summation order: bit-reproducible.
  if (g->sorted && !g->symmetric && g->rowptr && g->transposed && g->transposed->r_perm && spk_get_variant() == SPK_VARIANT_AUTO && (F == 128 || F == 64) &&
      rb->n_rbf <= 32 && g->n_edges > 0 && !getenv("SPK_NO_TRANSPOSED")) {
    // (T->r_perm is a workspace of the PLAN, rewritten by every call: two message backwards of the same plan must not run concurrently on different
    //  streams -- the operator library and the drivers issue a plan's launches on one stream)
    const spk_transposed_t* T = g->transposed;
    const int K = rb->n_rbf;
    const int grid = spk_grid_for(a.N, 4, spk_num_cus() * 2);
    const size_t lds = (size_t)K * (3 * F + 1) * sizeof(float);
    // large lists (round 6): the same two passes in row-tile form (spk_painn_tile.hip: filter from the split-precision GEMM of 32-pair chunks) --
    // the sums pass over the by-neighbour list, the geometry pass over the list itself; neither assumes symmetry
    const bool rowtile = spk_painn_msg_rowtile_bwd_ok(a);
    if (!geom_only) {
      hipLaunchKernelGGL(k_gather_rows3_neg, dim3(spk_grid_for(a.E, 256, spk_num_cus() * 8)), dim3(256), 0, stream, r_ij, T->perm, a.E, T->r_perm);
      SPK_LAUNCH_CHECK();
      MsgArgs t = a;
      t.idx_i = T->idx_i; t.idx_j = T->idx_j; t.rowptr = T->rowptr; t.rij = T->r_perm; t.gr = nullptr; t.blocks = nullptr; t.geom_only = 0;
      if (rowtile) { t.geom_only = 2; int rc2 = spk_painn_msg_rowtile_bwd(t, stream); if (rc2) return rc2; }
      else {
      SpkProfScope prof("painn_msg_bwd_row_tsum", stream);
#define SPK_TS_CASE(VPLv, NRBFv)                                                                                                               \
  do {                                                                                                                                         \
    if (t.mu_zero) hipLaunchKernelGGL((k_painn_msg_row<VPLv, NRBFv, true, false, true, false, false, true>), dim3(grid), dim3(256), lds, stream, t);  \
    else hipLaunchKernelGGL((k_painn_msg_row<VPLv, NRBFv, true, false, false, false, false, true>), dim3(grid), dim3(256), lds, stream, t);         \
  } while (0)
      if (F == 64) { if (K <= 20) SPK_TS_CASE(1, 20); else SPK_TS_CASE(1, 32); }
      else { if (K <= 20) SPK_TS_CASE(2, 20); else SPK_TS_CASE(2, 32); }
#undef SPK_TS_CASE
      SPK_LAUNCH_CHECK();
      }
    }
    MsgArgs gm = a;
    gm.geom_only = 1; gm.blocks = nullptr;
    if (rowtile) return spk_painn_msg_rowtile_bwd(gm, stream);
    SpkProfScope prof("painn_msg_bwd_row_geom", stream);
#define SPK_GM_CASE(VPLv, NRBFv)                                                                                                               \
  do {                                                                                                                                         \
    if (gm.mu_zero) hipLaunchKernelGGL((k_painn_msg_row<VPLv, NRBFv, true, true, true>), dim3(grid), dim3(256), lds, stream, gm);               \
    else hipLaunchKernelGGL((k_painn_msg_row<VPLv, NRBFv, true, true, false>), dim3(grid), dim3(256), lds, stream, gm);                         \
  } while (0)
    if (F == 64) { if (K <= 20) SPK_GM_CASE(1, 20); else SPK_GM_CASE(1, 32); }
    else { if (K <= 20) SPK_GM_CASE(2, 20); else SPK_GM_CASE(2, 32); }
#undef SPK_GM_CASE
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  return msg_dispatch<true>(a, g->sorted && g->symmetric && g->rowptr, stream, who);
}

extern "C" int spk_painn_message_fwd_f32(const spk_graph_t* g, const spk_radial_t* rb, const float* c,
                                         const float* q, const float* mu, const float* r_ij,
                                         const float* wf, const float* bf, int32_t F, float* q_out,
                                         float* mu_out, void* stream) {
  return spk_painn_message_fwd_internal(g, rb, c, q, mu, r_ij, wf, bf, F, q_out, mu_out, (hipStream_t)stream);
}

extern "C" int spk_painn_message_bwd_f32(const spk_graph_t* g, const spk_radial_t* rb, const float* c,
                                         const float* mu, const float* gq_out, const float* gmu_out,
                                         const float* r_ij, const float* wf, const float* bf, int32_t F,
                                         float* gc, float* gmu, float* gr, void* stream) {
  return spk_painn_message_bwd_internal(g, rb, c, mu, gq_out, gmu_out, r_ij, wf, bf, F, gc, gmu, gr, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// mixing block: elementwise parts (per atom, per channel)
// ------------------------------------------------------------------------------------------
// ctx[n] = (q[n] | sqrt(sum_x V^2 + eps)),  mix [N,3,2F] = (V | W)
__global__ void k_mix_ctx(const float* __restrict__ q, const float* __restrict__ mix, int64_t N,
                          int F, float eps, float* __restrict__ ctx) {
  const int64_t total = N * (int64_t)F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = t / F;
    const int f = (int)(t % F);
    const float v0 = mix[(n * 3 + 0) * 2 * F + f], v1 = mix[(n * 3 + 1) * 2 * F + f], v2 = mix[(n * 3 + 2) * 2 * F + f];
    ctx[n * 2 * F + f] = q[t];
    ctx[n * 2 * F + F + f] = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + eps);
  }
}

// q_out = q + a_q + a_qmu sum_x V W ; mu_out = mu + a_mu W
__global__ void k_mix_update(const float* __restrict__ q, const float* __restrict__ mu,
                             const float* __restrict__ mix, const float* __restrict__ a, int64_t N,
                             int F, float* q_out, float* mu_out) {
  const int64_t total = N * (int64_t)F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = t / F;
    const int f = (int)(t % F);
    const float aq = a[n * 3 * F + f], am = a[n * 3 * F + F + f], aqm = a[n * 3 * F + 2 * F + f];
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      const float V = mix[(n * 3 + x) * 2 * F + f], W = mix[(n * 3 + x) * 2 * F + F + f];
      s += V * W;
      mu_out[(n * 3 + x) * F + f] = mu[(n * 3 + x) * F + f] + am * W;
    }
    q_out[t] = q[t] + aq + aqm * s;
  }
}

// backward of k_mix_update w.r.t. a and mix (direct paths into q, mu are the caller's residuals)
__global__ void k_mix_update_bwd(const float* __restrict__ mix, const float* __restrict__ a,
                                 const float* __restrict__ gq_out, const float* __restrict__ gmu_out,
                                 int64_t N, int F, float* __restrict__ ga, float* __restrict__ gmix) {
  const int64_t total = N * (int64_t)F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = t / F;
    const int f = (int)(t % F);
    const float am = a[n * 3 * F + F + f], aqm = a[n * 3 * F + 2 * F + f];
    const float gq = gq_out[t];
    float s = 0.f, gam = 0.f;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      const float V = mix[(n * 3 + x) * 2 * F + f], W = mix[(n * 3 + x) * 2 * F + F + f];
      const float gm = gmu_out[(n * 3 + x) * F + f];
      s += V * W;
      gam += gm * W;
      gmix[(n * 3 + x) * 2 * F + f] = gq * aqm * W;                 // dL/dV (norm term added later)
      gmix[(n * 3 + x) * 2 * F + F + f] = gq * aqm * V + gm * am;   // dL/dW
    }
    ga[n * 3 * F + f] = gq;
    ga[n * 3 * F + F + f] = gam;
    ga[n * 3 * F + 2 * F + f] = gq * s;
  }
}

// backward of k_mix_ctx: gV += g_ctx[:,F:] V / Vn ; gq = gq_out + g_ctx[:, :F]
__global__ void k_mix_ctx_bwd(const float* __restrict__ mix, const float* __restrict__ g_ctx,
                              const float* __restrict__ gq_out, int64_t N, int F, float eps,
                              float* gmix, float* gq) {
  const int64_t total = N * (int64_t)F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = t / F;
    const int f = (int)(t % F);
    const float v0 = mix[(n * 3 + 0) * 2 * F + f], v1 = mix[(n * 3 + 1) * 2 * F + f], v2 = mix[(n * 3 + 2) * 2 * F + f];
    const float vn = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + eps);
    const float s = g_ctx[n * 2 * F + F + f] / vn;
    gmix[(n * 3 + 0) * 2 * F + f] += s * v0;
    gmix[(n * 3 + 1) * 2 * F + f] += s * v1;
    gmix[(n * 3 + 2) * 2 * F + f] += s * v2;
    gq[t] = gq_out[t] + g_ctx[n * 2 * F + f];
  }
}

// ------------------------------------------------------------------------------------------
// Fused PaiNNMixing forward (representation/painn.py:99-117) for n_atom_basis = 128: channel mix of mu
// ([3N, F] x [F, 2F]), norm, the two Dense layers of the intra-atomic context net and the update of (q, mu)
// in ONE launch.  Everything is local to an atom, so a workgroup (4 waves) owns 16 atoms and walks the
// stages with activations in LDS; weights come from the packed images (spk_pack_weight_f32) and the matrix
// work runs on v_mfma_f32_16x16x4_f32.  Wave w owns the 32 features [32 w, 32 w + 32) in EVERY stage -- the
// V and W halves of the channel mix for all three components, and the (q | mu | q mu) parts of the context
// output -- so |V|, sum_x V W and the final update are register-local: the mixed tensor never comes back
// from memory.  Replaces 4 launches (mix chain over 3N rows, mix_ctx, context chain, mix_update).
// mix, preB and a are still written once: the backward reads them.
// ------------------------------------------------------------------------------------------
struct MixFwdArgs {
  const float* q1; const float* mu1;
  const float* wmix;            // packed, contraction F,  width 2F
  const float* w1; const float* b1;   // packed, contraction 2F, width F
  const float* w2; const float* b2;   // packed, contraction F,  width 3F
  float eps;
  int64_t N;
  float* mix; float* preB; float* a; float* q_out; float* mu_out;
  const float* wmix_s; const float* w1_s; const float* w2_s;   // split-precision images of the same layers (spk_split.h; null: fp32 matrix path)
};

// A operands of pair p (32 output features) for u-steps [u0, u0 + 4) from a packed image with KB k-blocks.
// p must be wave-uniform (callers pass readfirstlane'd wave indices): the address is then a SCALAR base plus one
// per-lane byte offset shared by every load of the kernel (global_load ... v_off, s[base]) -- with per-lane 64-bit
// addresses the compiler hoisted ~40 address pairs out of the tile loop and the kernel needed 340 registers.
__device__ __forceinline__ void mix_load_a(f32x4 (&av)[2][4], const float* __restrict__ w, int KB, int p, int u0, int el, int h) {
  const char* base = (const char*)w + (((int64_t)p * KB + 2 * u0) * 64) * 16;
  const uint32_t off = (uint32_t)(((h >> 1) * 64 + (h & 1) * 32 + el) * 16);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    av[0][u] = *(const f32x4*)(base + off + u * 2048);
    av[1][u] = *(const f32x4*)(base + off + u * 2048 + 256);
  }
}
__device__ __forceinline__ void mix_mfma4(const f32x4 (&av)[2][4], const f32x4 (&bv)[4], f32x4& acc0, f32x4& acc1) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
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
// B operands of one 64-deep chunk c from an LDS row
__device__ __forceinline__ void mix_load_b(f32x4 (&bv)[4], const float* __restrict__ brow, int c, int h) {
#pragma unroll
  for (int u = 0; u < 4; ++u) bv[u] = *(const f32x4*)(brow + 64 * c + 16 * u + 4 * h);
}

// MINB = 2: two workgroups per CU, 256 VGPRs (spills 268 B/lane to scratch); MINB = 1: one workgroup per CU with the full
// 512-register budget (no scratch; the overflow sits in AGPRs).  Which one wins depends on how many 16-atom tiles a CU gets.
template <int F, int MINB>
__global__ __launch_bounds__(256, MINB) void k_painn_mixing_fwd(MixFwdArgs a) {
  static_assert(F == 128, "the weight stream below is written out for n_atom_basis = 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDA = F + 4, LDC = 2 * F + 4, LDH = F + 4;
  constexpr int KB1 = F / 8, KB2 = 2 * F / 8;   // k-blocks of the packed images (contraction F resp. 2F)
  float* sMu = smem;                 // [3][16][LDA]  mu entering the mixing
  float* sCt = sMu + 48 * LDA;       // [16][LDC]     context input [q | |V|]
  float* sHd = sCt + 16 * LDC;       // [16][LDH]     hidden layer
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 4, el = lane & 15;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (a.N + 15) / 16;
  // The weights do not depend on the data: the 14 chunks (64 contraction indices of one 32-feature pair) a wave
  // needs per tile are requested ONE CHUNK AHEAD through two register sets, across stage and tile boundaries.
  f32x4 wa[2][4], wb[2][4];
  if ((int64_t)blockIdx.x < ntiles) mix_load_a(wa, a.wmix, KB1, wv, 0, el, h);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * 16;
    const int64_t m = m0 + el;
    const bool valid = m < a.N;
    // ---- stage 0: mu tile (rows x-major: row = 16 x + atom) and q into LDS
    constexpr int Q4 = F / 4;
    for (int s = threadIdx.x; s < 48 * Q4; s += 256) {
      const int row = s / Q4, c4 = s - row * Q4;
      const int x = row >> 4, n = row & 15;
      int64_t mm = m0 + n;
      if (mm >= a.N) mm = a.N - 1;
      *(f32x4*)(sMu + row * LDA + 4 * c4) = *(const f32x4*)(a.mu1 + (mm * 3 + x) * F + 4 * c4);
    }
    for (int s = threadIdx.x; s < 16 * Q4; s += 256) {
      const int n = s / Q4, c4 = s - n * Q4;
      int64_t mm = m0 + n;
      if (mm >= a.N) mm = a.N - 1;
      *(f32x4*)(sCt + n * LDC + 4 * c4) = *(const f32x4*)(a.q1 + mm * F + 4 * c4);
    }
    __syncthreads();
    // ---- stage 1: channel mix; this wave: V = features [32 wv, +32), W = F + the same; the three components
    //      x share every weight chunk (3 x 32 MFMAs per chunk)
    f32x4 V[3][2], W[3][2];
#pragma unroll
    for (int x = 0; x < 3; ++x) { V[x][0] = z4; V[x][1] = z4; W[x][0] = z4; W[x][1] = z4; }
    f32x4 bv[4];
#define MIX_X3(WSET, C, ACC)                                                                      \
    _Pragma("unroll") for (int x = 0; x < 3; ++x) {                                               \
      mix_load_b(bv, sMu + (16 * x + el) * LDA, C, h);                                            \
      mix_mfma4(WSET, bv, ACC[x][0], ACC[x][1]);                                                  \
    }
    mix_load_a(wb, a.wmix, KB1, wv, 4, el, h);             MIX_X3(wa, 0, V) __builtin_amdgcn_sched_barrier(0);
    mix_load_a(wa, a.wmix, KB1, wv + F / 32, 0, el, h);    MIX_X3(wb, 1, V) __builtin_amdgcn_sched_barrier(0);
    mix_load_a(wb, a.wmix, KB1, wv + F / 32, 4, el, h);    MIX_X3(wa, 0, W) __builtin_amdgcn_sched_barrier(0);
    mix_load_a(wa, a.w1, KB2, wv, 0, el, h);               MIX_X3(wb, 1, W) __builtin_amdgcn_sched_barrier(0);
#undef MIX_X3
    f32x4 sVW[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int f0 = 32 * wv + 16 * k + 4 * h;
      f32x4 n2 = {a.eps, a.eps, a.eps, a.eps};
      sVW[k] = z4;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        n2 += V[x][k] * V[x][k];
        sVW[k] += V[x][k] * W[x][k];
        if (valid) {
          *(f32x4*)(a.mix + (m * 3 + x) * 2 * F + f0) = V[x][k];
          *(f32x4*)(a.mix + (m * 3 + x) * 2 * F + F + f0) = W[x][k];
        }
      }
      f32x4 vn;
      vn.x = sqrtf(n2.x); vn.y = sqrtf(n2.y); vn.z = sqrtf(n2.z); vn.w = sqrtf(n2.w);
      *(f32x4*)(sCt + el * LDC + F + f0) = vn;
    }
    __syncthreads();
    // ---- stage 2: hidden = silu([q | |V|] W1^T + b1), features [32 wv, +32), contraction 2F = 4 chunks
    {
      f32x4 acc0 = a.b1 ? *(const f32x4*)(a.b1 + 32 * wv + 4 * h) : z4;
      f32x4 acc1 = a.b1 ? *(const f32x4*)(a.b1 + 32 * wv + 16 + 4 * h) : z4;
      const float* brow = sCt + el * LDC;
      mix_load_a(wb, a.w1, KB2, wv, 4, el, h);   mix_load_b(bv, brow, 0, h); mix_mfma4(wa, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wa, a.w1, KB2, wv, 8, el, h);   mix_load_b(bv, brow, 1, h); mix_mfma4(wb, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wb, a.w1, KB2, wv, 12, el, h);  mix_load_b(bv, brow, 2, h); mix_mfma4(wa, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wa, a.w2, KB1, wv, 0, el, h);   mix_load_b(bv, brow, 3, h); mix_mfma4(wb, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int f0 = 32 * wv + 16 * k + 4 * h;
        f32x4 o = k ? acc1 : acc0;
        if (valid) *(f32x4*)(a.preB + m * F + f0) = o;
        o.x = o.x * spk_sigmoid(o.x); o.y = o.y * spk_sigmoid(o.y); o.z = o.z * spk_sigmoid(o.z); o.w = o.w * spk_sigmoid(o.w);
        *(f32x4*)(sHd + el * LDH + f0) = o;
      }
    }
    __syncthreads();
    // ---- stage 3: a = hidden W2^T + b2, parts (q | mu | q mu) of features [32 wv, +32)
    f32x4 A[3][2];
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      const int p = wv + part * (F / 32);
      A[part][0] = a.b2 ? *(const f32x4*)(a.b2 + 32 * p + 4 * h) : z4;
      A[part][1] = a.b2 ? *(const f32x4*)(a.b2 + 32 * p + 16 + 4 * h) : z4;
    }
    {
      const float* brow = sHd + el * LDH;
      const bool more = tile + gridDim.x < ntiles;
      mix_load_a(wb, a.w2, KB1, wv, 4, el, h);                  mix_load_b(bv, brow, 0, h); mix_mfma4(wa, bv, A[0][0], A[0][1]); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wa, a.w2, KB1, wv + F / 32, 0, el, h);         mix_load_b(bv, brow, 1, h); mix_mfma4(wb, bv, A[0][0], A[0][1]); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wb, a.w2, KB1, wv + F / 32, 4, el, h);         mix_load_b(bv, brow, 0, h); mix_mfma4(wa, bv, A[1][0], A[1][1]); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wa, a.w2, KB1, wv + 2 * (F / 32), 0, el, h);   mix_load_b(bv, brow, 1, h); mix_mfma4(wb, bv, A[1][0], A[1][1]); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wb, a.w2, KB1, wv + 2 * (F / 32), 4, el, h);   mix_load_b(bv, brow, 0, h); mix_mfma4(wa, bv, A[2][0], A[2][1]); __builtin_amdgcn_sched_barrier(0);
      if (more) mix_load_a(wa, a.wmix, KB1, wv, 0, el, h);      mix_load_b(bv, brow, 1, h); mix_mfma4(wb, bv, A[2][0], A[2][1]); __builtin_amdgcn_sched_barrier(0);
    }
    // ---- stage 4: q += a_q + a_qmu sum_x V W ;  mu += a_mu W      (painn.py:111-116); a is kept for the backward
    if (valid) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int f0 = 32 * wv + 16 * k + 4 * h;
#pragma unroll
        for (int part = 0; part < 3; ++part) *(f32x4*)(a.a + m * 3 * F + part * F + f0) = A[part][k];
        const f32x4 q = *(const f32x4*)(sCt + el * LDC + f0);
        *(f32x4*)(a.q_out + m * F + f0) = q + A[0][k] + A[2][k] * sVW[k];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          const f32x4 mu = *(const f32x4*)(sMu + (16 * x + el) * LDA + f0);
          *(f32x4*)(a.mu_out + (m * 3 + x) * F + f0) = mu + A[1][k] * W[x][k];
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// The same forward with EIGHT wavefronts per 16-atom tile: wave w owns the 16 features [16 w, 16 w + 16) in every stage, i.e. one
// 16x16 accumulator where the four-wave form has a pair.  Half the accumulators and half the weight registers per wave: the kernel
// fits the 256-register budget of two waves per SIMD without scratch, so the serial stages of a tile are walked by two waves per
// SIMD (one fills the other's load / LDS / barrier gaps) and a tile finishes in about half the time -- which is what the
// 336-tiles-on-256-CUs launch of configs[2] is bound by (one tile per workgroup, two rounds).
// ------------------------------------------------------------------------------------------
// A operands of the 16-feature tile (pair p, half k) for u-steps [u0, u0 + 4)
__device__ __forceinline__ void mix_load_a16(f32x4 (&av)[4], const float* __restrict__ w, int KB, int p, int k, int u0, int el, int h) {
  const char* base = (const char*)w + (((int64_t)p * KB + 2 * u0) * 64) * 16 + k * 256;
  const uint32_t off = (uint32_t)(((h >> 1) * 64 + (h & 1) * 32 + el) * 16);
#pragma unroll
  for (int u = 0; u < 4; ++u) av[u] = *(const f32x4*)(base + off + u * 2048);
}
__device__ __forceinline__ void mix_mfma4_16(const f32x4 (&av)[4], const f32x4 (&bv)[4], f32x4& acc) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u].w, acc, 0, 0, 0);
  }
}

template <int F>
__global__ __launch_bounds__(512, 2) void k_painn_mixing_fwd8(MixFwdArgs a) {
  static_assert(F == 128, "the weight stream below is written out for n_atom_basis = 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDA = F + 4, LDC = 2 * F + 4, LDH = F + 4;
  constexpr int KB1 = F / 8, KB2 = 2 * F / 8;
  constexpr int PW = F / 32;          // weight pairs per F output features
  float* sMu = smem;                 // [3][16][LDA]
  float* sCt = sMu + 48 * LDA;       // [16][LDC]
  float* sHd = sCt + 16 * LDC;       // [16][LDH]
  const int lane = threadIdx.x & 63, w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pw = w8 >> 1, kw = w8 & 1;             // this wave's pair and half inside a block of F output features
  const int h = lane >> 4, el = lane & 15;
  const int f0 = 16 * w8 + 4 * h;                  // this lane's four features
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (a.N + 15) / 16;
  f32x4 wa[4], wb[4];
  if ((int64_t)blockIdx.x < ntiles) mix_load_a16(wa, a.wmix, KB1, pw, kw, 0, el, h);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * 16;
    const int64_t m = m0 + el;
    const bool valid = m < a.N;
    constexpr int Q4 = F / 4;
    for (int s = threadIdx.x; s < 48 * Q4; s += 512) {
      const int row = s / Q4, c4 = s - row * Q4;
      const int x = row >> 4, n = row & 15;
      int64_t mm = m0 + n;
      if (mm >= a.N) mm = a.N - 1;
      *(f32x4*)(sMu + row * LDA + 4 * c4) = *(const f32x4*)(a.mu1 + (mm * 3 + x) * F + 4 * c4);
    }
    for (int s = threadIdx.x; s < 16 * Q4; s += 512) {
      const int n = s / Q4, c4 = s - n * Q4;
      int64_t mm = m0 + n;
      if (mm >= a.N) mm = a.N - 1;
      *(f32x4*)(sCt + n * LDC + 4 * c4) = *(const f32x4*)(a.q1 + mm * F + 4 * c4);
    }
    __syncthreads();
    // ---- stage 1: channel mix
    f32x4 V[3], W[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) { V[x] = z4; W[x] = z4; }
    f32x4 bv[4];
#define MIX8_X3(WSET, C, ACC)                                                                     \
    _Pragma("unroll") for (int x = 0; x < 3; ++x) {                                               \
      mix_load_b(bv, sMu + (16 * x + el) * LDA, C, h);                                            \
      mix_mfma4_16(WSET, bv, ACC[x]);                                                             \
    }
    mix_load_a16(wb, a.wmix, KB1, pw, kw, 4, el, h);          MIX8_X3(wa, 0, V) __builtin_amdgcn_sched_barrier(0);
    mix_load_a16(wa, a.wmix, KB1, pw + PW, kw, 0, el, h);     MIX8_X3(wb, 1, V) __builtin_amdgcn_sched_barrier(0);
    mix_load_a16(wb, a.wmix, KB1, pw + PW, kw, 4, el, h);     MIX8_X3(wa, 0, W) __builtin_amdgcn_sched_barrier(0);
    mix_load_a16(wa, a.w1, KB2, pw, kw, 0, el, h);            MIX8_X3(wb, 1, W) __builtin_amdgcn_sched_barrier(0);
#undef MIX8_X3
    f32x4 sVW = z4;
    {
      f32x4 n2 = {a.eps, a.eps, a.eps, a.eps};
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        n2 += V[x] * V[x];
        sVW += V[x] * W[x];
        if (valid) {
          *(f32x4*)(a.mix + (m * 3 + x) * 2 * F + f0) = V[x];
          *(f32x4*)(a.mix + (m * 3 + x) * 2 * F + F + f0) = W[x];
        }
      }
      f32x4 vn;
      vn.x = sqrtf(n2.x); vn.y = sqrtf(n2.y); vn.z = sqrtf(n2.z); vn.w = sqrtf(n2.w);
      *(f32x4*)(sCt + el * LDC + F + f0) = vn;
    }
    __syncthreads();
    // ---- stage 2: hidden = silu([q | |V|] W1^T + b1)
    {
      f32x4 acc = a.b1 ? *(const f32x4*)(a.b1 + f0) : z4;
      const float* brow = sCt + el * LDC;
      mix_load_a16(wb, a.w1, KB2, pw, kw, 4, el, h);   mix_load_b(bv, brow, 0, h); mix_mfma4_16(wa, bv, acc); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wa, a.w1, KB2, pw, kw, 8, el, h);   mix_load_b(bv, brow, 1, h); mix_mfma4_16(wb, bv, acc); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wb, a.w1, KB2, pw, kw, 12, el, h);  mix_load_b(bv, brow, 2, h); mix_mfma4_16(wa, bv, acc); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wa, a.w2, KB1, pw, kw, 0, el, h);   mix_load_b(bv, brow, 3, h); mix_mfma4_16(wb, bv, acc); __builtin_amdgcn_sched_barrier(0);
      f32x4 o = acc;
      if (valid) *(f32x4*)(a.preB + m * F + f0) = o;
      o.x = o.x * spk_sigmoid(o.x); o.y = o.y * spk_sigmoid(o.y); o.z = o.z * spk_sigmoid(o.z); o.w = o.w * spk_sigmoid(o.w);
      *(f32x4*)(sHd + el * LDH + f0) = o;
    }
    __syncthreads();
    // ---- stage 3: a = hidden W2^T + b2, parts (q | mu | q mu)
    f32x4 A[3];
#pragma unroll
    for (int part = 0; part < 3; ++part) A[part] = a.b2 ? *(const f32x4*)(a.b2 + part * F + f0) : z4;
    {
      const float* brow = sHd + el * LDH;
      const bool more = tile + gridDim.x < ntiles;
      mix_load_a16(wb, a.w2, KB1, pw, kw, 4, el, h);             mix_load_b(bv, brow, 0, h); mix_mfma4_16(wa, bv, A[0]); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wa, a.w2, KB1, pw + PW, kw, 0, el, h);        mix_load_b(bv, brow, 1, h); mix_mfma4_16(wb, bv, A[0]); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wb, a.w2, KB1, pw + PW, kw, 4, el, h);        mix_load_b(bv, brow, 0, h); mix_mfma4_16(wa, bv, A[1]); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wa, a.w2, KB1, pw + 2 * PW, kw, 0, el, h);    mix_load_b(bv, brow, 1, h); mix_mfma4_16(wb, bv, A[1]); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wb, a.w2, KB1, pw + 2 * PW, kw, 4, el, h);    mix_load_b(bv, brow, 0, h); mix_mfma4_16(wa, bv, A[2]); __builtin_amdgcn_sched_barrier(0);
      if (more) mix_load_a16(wa, a.wmix, KB1, pw, kw, 0, el, h); mix_load_b(bv, brow, 1, h); mix_mfma4_16(wb, bv, A[2]); __builtin_amdgcn_sched_barrier(0);
    }
    // ---- stage 4: q += a_q + a_qmu sum_x V W ;  mu += a_mu W
    if (valid) {
#pragma unroll
      for (int part = 0; part < 3; ++part) *(f32x4*)(a.a + m * 3 * F + part * F + f0) = A[part];
      const f32x4 q = *(const f32x4*)(sCt + el * LDC + f0);
      *(f32x4*)(a.q_out + m * F + f0) = q + A[0] + A[2] * sVW;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const f32x4 mu = *(const f32x4*)(sMu + (16 * x + el) * LDA + f0);
        *(f32x4*)(a.mu_out + (m * 3 + x) * F + f0) = mu + A[1] * W[x];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Fused first half of the PaiNNMixing backward (n_atom_basis = 128): gradient of the update w.r.t. the context
// output a and the mixed tensor (k_mix_update_bwd), the two transposed Dense layers of the intra-atomic context net,
// and the gradient of the [q | |V|] context input (k_mix_ctx_bwd) in ONE launch -- 3 launches before.  Same
// ownership as the forward: wave w owns features [32 w, 32 w + 32), so dL/dV, dL/dW, |V| stay in registers
// between the stages.  Outputs: gq1 (dL/dq entering the mixing) and gmix (dL/d mixed tensor); the channel-mix
// chain over 3N rows follows as before.
// ------------------------------------------------------------------------------------------
struct MixBwdArgs {
  const float* gq; const float* gmu;      // dL/dq_out [N,F], dL/dmu_out [N,3,F]
  const float* mix; const float* a; const float* preB;   // saved by the forward
  const float* w2t;                       // packed input-gradient image of ictx_w2: contraction 3F, width F
  const float* w1t;                       // packed input-gradient image of ictx_w1: contraction F,  width 2F
  float eps;
  int64_t N;
  float* gq1; float* gmix;
  const float* w2t_s; const float* w1t_s;     // split-precision images (null: fp32 matrix path)
};

template <int F, int MINB>
__global__ __launch_bounds__(256, MINB) void k_painn_mixing_bwd(MixBwdArgs a) {
  static_assert(F == 128, "the weight stream below is written out for n_atom_basis = 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDG = 3 * F + 4, LDT = F + 4;
  constexpr int KB3 = 3 * F / 8, KB1 = F / 8;
  float* sGa = smem;                 // [16][LDG]  dL/da  (q | mu | q mu)
  float* sT = sGa + 16 * LDG;        // [16][LDT]  dL/d hidden pre-activation
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 4, el = lane & 15;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (a.N + 15) / 16;
  f32x4 wa[2][4], wb[2][4], bv[4];
  if ((int64_t)blockIdx.x < ntiles) mix_load_a(wa, a.w2t, KB3, wv, 0, el, h);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * 16;
    int64_t m = m0 + el;
    const bool valid = m < a.N;
    if (!valid) m = a.N - 1;
    // ---- stage 0: gradient of the update (painn.py:111-116) for this wave's features
    f32x4 V[3][2], gV[3][2], gq4[2], invn[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int f0 = 32 * wv + 16 * k + 4 * h;
      const f32x4 amu = *(const f32x4*)(a.a + m * 3 * F + F + f0), aqm = *(const f32x4*)(a.a + m * 3 * F + 2 * F + f0);
      gq4[k] = *(const f32x4*)(a.gq + m * F + f0);
      f32x4 s = z4, gam = z4, n2 = {a.eps, a.eps, a.eps, a.eps};
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const f32x4 v = *(const f32x4*)(a.mix + (m * 3 + x) * 2 * F + f0);
        const f32x4 w = *(const f32x4*)(a.mix + (m * 3 + x) * 2 * F + F + f0);
        const f32x4 gm = *(const f32x4*)(a.gmu + (m * 3 + x) * F + f0);
        V[x][k] = v;
        s += v * w; gam += gm * w; n2 += v * v;
        gV[x][k] = gq4[k] * aqm * w;                                   // dL/dV, the norm term follows in stage 2
        if (valid) *(f32x4*)(a.gmix + (m * 3 + x) * 2 * F + F + f0) = gq4[k] * aqm * v + gm * amu;   // dL/dW
      }
      invn[k].x = 1.0f / sqrtf(n2.x); invn[k].y = 1.0f / sqrtf(n2.y); invn[k].z = 1.0f / sqrtf(n2.z); invn[k].w = 1.0f / sqrtf(n2.w);
      *(f32x4*)(sGa + el * LDG + f0) = gq4[k];
      *(f32x4*)(sGa + el * LDG + F + f0) = gam;
      *(f32x4*)(sGa + el * LDG + 2 * F + f0) = gq4[k] * s;
    }
    __syncthreads();
    // ---- stage 1: t = (ga W2) * silu'(preB), features [32 wv, +32), contraction 3F = 6 chunks
    {
      f32x4 acc0 = z4, acc1 = z4;
      const float* brow = sGa + el * LDG;
      mix_load_a(wb, a.w2t, KB3, wv, 4, el, h);   mix_load_b(bv, brow, 0, h); mix_mfma4(wa, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wa, a.w2t, KB3, wv, 8, el, h);   mix_load_b(bv, brow, 1, h); mix_mfma4(wb, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wb, a.w2t, KB3, wv, 12, el, h);  mix_load_b(bv, brow, 2, h); mix_mfma4(wa, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wa, a.w2t, KB3, wv, 16, el, h);  mix_load_b(bv, brow, 3, h); mix_mfma4(wb, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wb, a.w2t, KB3, wv, 20, el, h);  mix_load_b(bv, brow, 4, h); mix_mfma4(wa, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wa, a.w1t, KB1, wv, 0, el, h);   mix_load_b(bv, brow, 5, h); mix_mfma4(wb, bv, acc0, acc1); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int f0 = 32 * wv + 16 * k + 4 * h;
        const f32x4 pb = *(const f32x4*)(a.preB + m * F + f0);
        f32x4 o = k ? acc1 : acc0;
        o.x *= spk_act_grad<SPK_ACT_SILU>(pb.x); o.y *= spk_act_grad<SPK_ACT_SILU>(pb.y);
        o.z *= spk_act_grad<SPK_ACT_SILU>(pb.z); o.w *= spk_act_grad<SPK_ACT_SILU>(pb.w);
        *(f32x4*)(sT + el * LDT + f0) = o;
      }
    }
    __syncthreads();
    // ---- stage 2: g_ctx = t W1: the q part feeds dL/dq, the |V| part the norm term of dL/dV
    {
      f32x4 q0 = z4, q1 = z4, n0 = z4, n1 = z4;
      const float* brow = sT + el * LDT;
      const bool more = tile + gridDim.x < ntiles;
      mix_load_a(wb, a.w1t, KB1, wv, 4, el, h);              mix_load_b(bv, brow, 0, h); mix_mfma4(wa, bv, q0, q1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wa, a.w1t, KB1, wv + F / 32, 0, el, h);     mix_load_b(bv, brow, 1, h); mix_mfma4(wb, bv, q0, q1); __builtin_amdgcn_sched_barrier(0);
      mix_load_a(wb, a.w1t, KB1, wv + F / 32, 4, el, h);     mix_load_b(bv, brow, 0, h); mix_mfma4(wa, bv, n0, n1); __builtin_amdgcn_sched_barrier(0);
      if (more) mix_load_a(wa, a.w2t, KB3, wv, 0, el, h);    mix_load_b(bv, brow, 1, h); mix_mfma4(wb, bv, n0, n1); __builtin_amdgcn_sched_barrier(0);
      if (valid) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int f0 = 32 * wv + 16 * k + 4 * h;
          *(f32x4*)(a.gq1 + m * F + f0) = gq4[k] + (k ? q1 : q0);
          const f32x4 sc = (k ? n1 : n0) * invn[k];
#pragma unroll
          for (int x = 0; x < 3; ++x) *(f32x4*)(a.gmix + (m * 3 + x) * 2 * F + f0) = gV[x][k] + sc * V[x][k];
        }
      }
    }
    __syncthreads();
  }
}

// The same backward with eight wavefronts per tile (see k_painn_mixing_fwd8): wave w owns the features [16 w, 16 w + 16).
template <int F>
__global__ __launch_bounds__(512, 2) void k_painn_mixing_bwd8(MixBwdArgs a) {
  static_assert(F == 128, "the weight stream below is written out for n_atom_basis = 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDG = 3 * F + 4, LDT = F + 4;
  constexpr int KB3 = 3 * F / 8, KB1 = F / 8;
  constexpr int PW = F / 32;
  float* sGa = smem;                 // [16][LDG]  dL/da  (q | mu | q mu)
  float* sT = sGa + 16 * LDG;        // [16][LDT]  dL/d hidden pre-activation
  const int lane = threadIdx.x & 63, w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pw = w8 >> 1, kw = w8 & 1;
  const int h = lane >> 4, el = lane & 15;
  const int f0 = 16 * w8 + 4 * h;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (a.N + 15) / 16;
  f32x4 wa[4], wb[4], bv[4];
  if ((int64_t)blockIdx.x < ntiles) mix_load_a16(wa, a.w2t, KB3, pw, kw, 0, el, h);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * 16;
    int64_t m = m0 + el;
    const bool valid = m < a.N;
    if (!valid) m = a.N - 1;
    // ---- stage 0: gradient of the update (painn.py:111-116) for this wave's features
    f32x4 V[3], gV[3], gq4, invn;
    {
      const f32x4 amu = *(const f32x4*)(a.a + m * 3 * F + F + f0), aqm = *(const f32x4*)(a.a + m * 3 * F + 2 * F + f0);
      gq4 = *(const f32x4*)(a.gq + m * F + f0);
      f32x4 s = z4, gam = z4, n2 = {a.eps, a.eps, a.eps, a.eps};
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const f32x4 v = *(const f32x4*)(a.mix + (m * 3 + x) * 2 * F + f0);
        const f32x4 w = *(const f32x4*)(a.mix + (m * 3 + x) * 2 * F + F + f0);
        const f32x4 gm = *(const f32x4*)(a.gmu + (m * 3 + x) * F + f0);
        V[x] = v;
        s += v * w; gam += gm * w; n2 += v * v;
        gV[x] = gq4 * aqm * w;                                         // dL/dV, the norm term follows in stage 2
        if (valid) *(f32x4*)(a.gmix + (m * 3 + x) * 2 * F + F + f0) = gq4 * aqm * v + gm * amu;   // dL/dW
      }
      invn.x = 1.0f / sqrtf(n2.x); invn.y = 1.0f / sqrtf(n2.y); invn.z = 1.0f / sqrtf(n2.z); invn.w = 1.0f / sqrtf(n2.w);
      *(f32x4*)(sGa + el * LDG + f0) = gq4;
      *(f32x4*)(sGa + el * LDG + F + f0) = gam;
      *(f32x4*)(sGa + el * LDG + 2 * F + f0) = gq4 * s;
    }
    __syncthreads();
    // ---- stage 1: t = (ga W2) * silu'(preB), contraction 3F = 6 chunks
    {
      f32x4 acc = z4;
      const float* brow = sGa + el * LDG;
      mix_load_a16(wb, a.w2t, KB3, pw, kw, 4, el, h);   mix_load_b(bv, brow, 0, h); mix_mfma4_16(wa, bv, acc); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wa, a.w2t, KB3, pw, kw, 8, el, h);   mix_load_b(bv, brow, 1, h); mix_mfma4_16(wb, bv, acc); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wb, a.w2t, KB3, pw, kw, 12, el, h);  mix_load_b(bv, brow, 2, h); mix_mfma4_16(wa, bv, acc); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wa, a.w2t, KB3, pw, kw, 16, el, h);  mix_load_b(bv, brow, 3, h); mix_mfma4_16(wb, bv, acc); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wb, a.w2t, KB3, pw, kw, 20, el, h);  mix_load_b(bv, brow, 4, h); mix_mfma4_16(wa, bv, acc); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wa, a.w1t, KB1, pw, kw, 0, el, h);   mix_load_b(bv, brow, 5, h); mix_mfma4_16(wb, bv, acc); __builtin_amdgcn_sched_barrier(0);
      const f32x4 pb = *(const f32x4*)(a.preB + m * F + f0);
      f32x4 o = acc;
      o.x *= spk_act_grad<SPK_ACT_SILU>(pb.x); o.y *= spk_act_grad<SPK_ACT_SILU>(pb.y);
      o.z *= spk_act_grad<SPK_ACT_SILU>(pb.z); o.w *= spk_act_grad<SPK_ACT_SILU>(pb.w);
      *(f32x4*)(sT + el * LDT + f0) = o;
    }
    __syncthreads();
    // ---- stage 2: g_ctx = t W1: the q part feeds dL/dq, the |V| part the norm term of dL/dV
    {
      f32x4 q0 = z4, n0 = z4;
      const float* brow = sT + el * LDT;
      const bool more = tile + gridDim.x < ntiles;
      mix_load_a16(wb, a.w1t, KB1, pw, kw, 4, el, h);             mix_load_b(bv, brow, 0, h); mix_mfma4_16(wa, bv, q0); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wa, a.w1t, KB1, pw + PW, kw, 0, el, h);        mix_load_b(bv, brow, 1, h); mix_mfma4_16(wb, bv, q0); __builtin_amdgcn_sched_barrier(0);
      mix_load_a16(wb, a.w1t, KB1, pw + PW, kw, 4, el, h);        mix_load_b(bv, brow, 0, h); mix_mfma4_16(wa, bv, n0); __builtin_amdgcn_sched_barrier(0);
      if (more) mix_load_a16(wa, a.w2t, KB3, pw, kw, 0, el, h);   mix_load_b(bv, brow, 1, h); mix_mfma4_16(wb, bv, n0); __builtin_amdgcn_sched_barrier(0);
      if (valid) {
        *(f32x4*)(a.gq1 + m * F + f0) = gq4 + q0;
        const f32x4 sc = n0 * invn;
#pragma unroll
        for (int x = 0; x < 3; ++x) *(f32x4*)(a.gmix + (m * 3 + x) * 2 * F + f0) = gV[x] + sc * V[x];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Split-precision forms of the two eight-wave kernels (round 6; spk_split.h): the same stages, the same ownership (wave w owns the features
// [16 w, 16 w + 16) of every stage), the products on v_mfma_f32_16x16x32_f16 with (high, low) fp16 operand pairs -- 6 instructions of 16 cycles
// per 64-k chunk and tile instead of 16 of 32.  Weights come from the split packed images (32-row tile geometry of k_pack_weight_split: the lane
// of a 16 x 16 x 32 A operand picks its 16 bytes out of the two 16-k steps of a 32-k step); the activations of a tile live in LDS as fp16 images --
// a row of K values is [K high halves | K low halves | pad], the byte length of the fp32 row -- split once by whoever writes them.
// ------------------------------------------------------------------------------------------
#include "spk_split.h"
#define MIXS_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16((A), (B), (C), 0, 0, 0)
// A operands (high, low) x (two 32-k steps) of the 16-feature tile (32-row tile p, half k) for the 64-k chunk starting at 16-k step u0
__device__ __forceinline__ void mixs_load_a(f32x4 (&av)[4], const float* __restrict__ w, int KB, int p, int k, int u0, int el, int g) {
  const char* base = (const char*)w + (((int64_t)p * KB + 2 * u0) * 64) * 16;
  const uint32_t off = (uint32_t)((2 * (g >> 1)) * 1024 + ((g & 1) * 32 + 16 * k + el) * 16);
#pragma unroll
  for (int ds = 0; ds < 2; ++ds) {
    av[2 * ds] = *(const f32x4*)(base + off + (4 * ds) * 1024);          // high parts of 32-k step ds
    av[2 * ds + 1] = *(const f32x4*)(base + off + (4 * ds + 1) * 1024);  // low parts
  }
}
// one 64-k chunk c of the product: B operands from the split LDS row `row` (K = contraction length of the row)
__device__ __forceinline__ void mixs_mfma(const f32x4 (&av)[4], const _Float16* __restrict__ row, int K, int c, int g, f32x4& acc, f32x4& crs) {
#pragma unroll
  for (int ds = 0; ds < 2; ++ds) {
    const h16x8 bh = *(const h16x8*)(row + 64 * c + 32 * ds + 8 * g);
    const h16x8 bl = *(const h16x8*)(row + K + 64 * c + 32 * ds + 8 * g);
    const h16x8 ah = __builtin_bit_cast(h16x8, av[2 * ds]), al = __builtin_bit_cast(h16x8, av[2 * ds + 1]);
    acc = MIXS_MFMA(ah, bh, acc);
    crs = MIXS_MFMA(ah, bl, crs);
    crs = MIXS_MFMA(al, bh, crs);
  }
}
__device__ __forceinline__ f32x4 mixs_fold(f32x4 acc, f32x4 crs) { return acc + crs * SP_DOWN; }
// four consecutive values of a split LDS row
__device__ __forceinline__ void mixs_store4(_Float16* __restrict__ row, int K, int col, f32x4 v) {
  h16x4 h, l;
  sp_split4(v, h, l);
  *(h16x4*)(row + col) = h; *(h16x4*)(row + K + col) = l;
}

template <int F>
__global__ __launch_bounds__(512, 2) void k_painn_mixing_fwd8s(MixFwdArgs a) {
  static_assert(F == 128, "the weight stream below is written out for n_atom_basis = 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDA = 2 * (F + 4), LDC = 2 * (2 * F + 4), LDH = 2 * (F + 4);      // row strides in HALVES = the fp32 rows' bytes / 2
  constexpr int KB1 = F / 8, KB2 = 2 * F / 8;
  constexpr int PW = F / 32;
  _Float16* sMu = (_Float16*)smem;           // [3][16] rows of F
  _Float16* sCt = sMu + 48 * LDA;            // [16] rows of 2F: [q | |V|]
  _Float16* sHd = sCt + 16 * LDC;            // [16] rows of F
  const int lane = threadIdx.x & 63, w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pw = w8 >> 1, kw = w8 & 1;
  const int h = lane >> 4, el = lane & 15;
  const int f0 = 16 * w8 + 4 * h;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (a.N + 15) / 16;
  f32x4 wa[4], wb[4];
  if ((int64_t)blockIdx.x < ntiles) mixs_load_a(wa, a.wmix_s, KB1, pw, kw, 0, el, h);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * 16;
    int64_t m = m0 + el;
    const bool valid = m < a.N;
    if (!valid) m = a.N - 1;
    constexpr int Q4 = F / 4;
    for (int s = threadIdx.x; s < 48 * Q4; s += 512) {
      const int row = s / Q4, c4 = s - row * Q4;
      const int x = row >> 4, n = row & 15;
      int64_t mm = m0 + n;
      if (mm >= a.N) mm = a.N - 1;
      mixs_store4(sMu + row * LDA, F, 4 * c4, *(const f32x4*)(a.mu1 + (mm * 3 + x) * F + 4 * c4));
    }
    for (int s = threadIdx.x; s < 16 * Q4; s += 512) {
      const int n = s / Q4, c4 = s - n * Q4;
      int64_t mm = m0 + n;
      if (mm >= a.N) mm = a.N - 1;
      mixs_store4(sCt + n * LDC, 2 * F, 4 * c4, *(const f32x4*)(a.q1 + mm * F + 4 * c4));
    }
    __syncthreads();
    // ---- stage 1: channel mix
    f32x4 V[3], W[3], Vx[3], Wx[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) { V[x] = z4; W[x] = z4; Vx[x] = z4; Wx[x] = z4; }
#define MIXS_X3(WSET, C, ACC, CRS)                                                                \
    _Pragma("unroll") for (int x = 0; x < 3; ++x) mixs_mfma(WSET, sMu + (16 * x + el) * LDA, F, C, h, ACC[x], CRS[x]);
    mixs_load_a(wb, a.wmix_s, KB1, pw, kw, 4, el, h);          MIXS_X3(wa, 0, V, Vx) __builtin_amdgcn_sched_barrier(0);
    mixs_load_a(wa, a.wmix_s, KB1, pw + PW, kw, 0, el, h);     MIXS_X3(wb, 1, V, Vx) __builtin_amdgcn_sched_barrier(0);
    mixs_load_a(wb, a.wmix_s, KB1, pw + PW, kw, 4, el, h);     MIXS_X3(wa, 0, W, Wx) __builtin_amdgcn_sched_barrier(0);
    mixs_load_a(wa, a.w1_s, KB2, pw, kw, 0, el, h);            MIXS_X3(wb, 1, W, Wx) __builtin_amdgcn_sched_barrier(0);
#undef MIXS_X3
    f32x4 sVW = z4;
    {
      f32x4 n2 = {a.eps, a.eps, a.eps, a.eps};
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        V[x] = mixs_fold(V[x], Vx[x]); W[x] = mixs_fold(W[x], Wx[x]);
        n2 += V[x] * V[x];
        sVW += V[x] * W[x];
        if (valid) {
          *(f32x4*)(a.mix + (m * 3 + x) * 2 * F + f0) = V[x];
          *(f32x4*)(a.mix + (m * 3 + x) * 2 * F + F + f0) = W[x];
        }
      }
      f32x4 vn;
      vn.x = sqrtf(n2.x); vn.y = sqrtf(n2.y); vn.z = sqrtf(n2.z); vn.w = sqrtf(n2.w);
      mixs_store4(sCt + el * LDC, 2 * F, F + f0, vn);
    }
    __syncthreads();
    // ---- stage 2: hidden = silu([q | |V|] W1^T + b1)
    {
      f32x4 acc = a.b1 ? *(const f32x4*)(a.b1 + f0) : z4, crs = z4;
      const _Float16* brow = sCt + el * LDC;
      mixs_load_a(wb, a.w1_s, KB2, pw, kw, 4, el, h);   mixs_mfma(wa, brow, 2 * F, 0, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wa, a.w1_s, KB2, pw, kw, 8, el, h);   mixs_mfma(wb, brow, 2 * F, 1, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wb, a.w1_s, KB2, pw, kw, 12, el, h);  mixs_mfma(wa, brow, 2 * F, 2, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wa, a.w2_s, KB1, pw, kw, 0, el, h);   mixs_mfma(wb, brow, 2 * F, 3, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      f32x4 o = mixs_fold(acc, crs);
      if (valid) *(f32x4*)(a.preB + m * F + f0) = o;
      o.x = o.x * spk_sigmoid(o.x); o.y = o.y * spk_sigmoid(o.y); o.z = o.z * spk_sigmoid(o.z); o.w = o.w * spk_sigmoid(o.w);
      mixs_store4(sHd + el * LDH, F, f0, o);
    }
    __syncthreads();
    // ---- stage 3: a = hidden W2^T + b2, parts (q | mu | q mu)
    f32x4 A[3], Ax[3];
#pragma unroll
    for (int part = 0; part < 3; ++part) { A[part] = a.b2 ? *(const f32x4*)(a.b2 + part * F + f0) : z4; Ax[part] = z4; }
    {
      const _Float16* brow = sHd + el * LDH;
      const bool more = tile + gridDim.x < ntiles;
      mixs_load_a(wb, a.w2_s, KB1, pw, kw, 4, el, h);              mixs_mfma(wa, brow, F, 0, h, A[0], Ax[0]); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wa, a.w2_s, KB1, pw + PW, kw, 0, el, h);         mixs_mfma(wb, brow, F, 1, h, A[0], Ax[0]); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wb, a.w2_s, KB1, pw + PW, kw, 4, el, h);         mixs_mfma(wa, brow, F, 0, h, A[1], Ax[1]); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wa, a.w2_s, KB1, pw + 2 * PW, kw, 0, el, h);     mixs_mfma(wb, brow, F, 1, h, A[1], Ax[1]); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wb, a.w2_s, KB1, pw + 2 * PW, kw, 4, el, h);     mixs_mfma(wa, brow, F, 0, h, A[2], Ax[2]); __builtin_amdgcn_sched_barrier(0);
      if (more) mixs_load_a(wa, a.wmix_s, KB1, pw, kw, 0, el, h);  mixs_mfma(wb, brow, F, 1, h, A[2], Ax[2]); __builtin_amdgcn_sched_barrier(0);
    }
    // ---- stage 4: q += a_q + a_qmu sum_x V W ;  mu += a_mu W   (q and mu re-read in fp32: the LDS copies are split images)
    if (valid) {
#pragma unroll
      for (int part = 0; part < 3; ++part) { A[part] = mixs_fold(A[part], Ax[part]); *(f32x4*)(a.a + m * 3 * F + part * F + f0) = A[part]; }
      const f32x4 q = *(const f32x4*)(a.q1 + m * F + f0);
      *(f32x4*)(a.q_out + m * F + f0) = q + A[0] + A[2] * sVW;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const f32x4 mu = *(const f32x4*)(a.mu1 + (m * 3 + x) * F + f0);
        *(f32x4*)(a.mu_out + (m * 3 + x) * F + f0) = mu + A[1] * W[x];
      }
    }
    __syncthreads();
  }
}

template <int F>
__global__ __launch_bounds__(512, 2) void k_painn_mixing_bwd8s(MixBwdArgs a) {
  static_assert(F == 128, "the weight stream below is written out for n_atom_basis = 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LDG = 2 * (3 * F + 4), LDT = 2 * (F + 4);      // row strides in halves
  constexpr int KB3 = 3 * F / 8, KB1 = F / 8;
  constexpr int PW = F / 32;
  _Float16* sGa = (_Float16*)smem;       // [16] rows of 3F: dL/da (q | mu | q mu)
  _Float16* sT = sGa + 16 * LDG;         // [16] rows of F: dL/d hidden pre-activation
  const int lane = threadIdx.x & 63, w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pw = w8 >> 1, kw = w8 & 1;
  const int h = lane >> 4, el = lane & 15;
  const int f0 = 16 * w8 + 4 * h;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (a.N + 15) / 16;
  f32x4 wa[4], wb[4];
  if ((int64_t)blockIdx.x < ntiles) mixs_load_a(wa, a.w2t_s, KB3, pw, kw, 0, el, h);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * 16;
    int64_t m = m0 + el;
    const bool valid = m < a.N;
    if (!valid) m = a.N - 1;
    // ---- stage 0: gradient of the update (painn.py:111-116) for this wave's features
    f32x4 V[3], gV[3], gq4, invn;
    {
      const f32x4 amu = *(const f32x4*)(a.a + m * 3 * F + F + f0), aqm = *(const f32x4*)(a.a + m * 3 * F + 2 * F + f0);
      gq4 = *(const f32x4*)(a.gq + m * F + f0);
      f32x4 s = z4, gam = z4, n2 = {a.eps, a.eps, a.eps, a.eps};
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const f32x4 v = *(const f32x4*)(a.mix + (m * 3 + x) * 2 * F + f0);
        const f32x4 w = *(const f32x4*)(a.mix + (m * 3 + x) * 2 * F + F + f0);
        const f32x4 gm = *(const f32x4*)(a.gmu + (m * 3 + x) * F + f0);
        V[x] = v;
        s += v * w; gam += gm * w; n2 += v * v;
        gV[x] = gq4 * aqm * w;
        if (valid) *(f32x4*)(a.gmix + (m * 3 + x) * 2 * F + F + f0) = gq4 * aqm * v + gm * amu;
      }
      invn.x = 1.0f / sqrtf(n2.x); invn.y = 1.0f / sqrtf(n2.y); invn.z = 1.0f / sqrtf(n2.z); invn.w = 1.0f / sqrtf(n2.w);
      mixs_store4(sGa + el * LDG, 3 * F, f0, gq4);
      mixs_store4(sGa + el * LDG, 3 * F, F + f0, gam);
      mixs_store4(sGa + el * LDG, 3 * F, 2 * F + f0, gq4 * s);
    }
    __syncthreads();
    // ---- stage 1: t = (ga W2) * silu'(preB), contraction 3F = 6 chunks
    {
      f32x4 acc = z4, crs = z4;
      const _Float16* brow = sGa + el * LDG;
      mixs_load_a(wb, a.w2t_s, KB3, pw, kw, 4, el, h);   mixs_mfma(wa, brow, 3 * F, 0, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wa, a.w2t_s, KB3, pw, kw, 8, el, h);   mixs_mfma(wb, brow, 3 * F, 1, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wb, a.w2t_s, KB3, pw, kw, 12, el, h);  mixs_mfma(wa, brow, 3 * F, 2, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wa, a.w2t_s, KB3, pw, kw, 16, el, h);  mixs_mfma(wb, brow, 3 * F, 3, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wb, a.w2t_s, KB3, pw, kw, 20, el, h);  mixs_mfma(wa, brow, 3 * F, 4, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wa, a.w1t_s, KB1, pw, kw, 0, el, h);   mixs_mfma(wb, brow, 3 * F, 5, h, acc, crs); __builtin_amdgcn_sched_barrier(0);
      const f32x4 pb = *(const f32x4*)(a.preB + m * F + f0);
      f32x4 o = mixs_fold(acc, crs);
      o.x *= spk_act_grad<SPK_ACT_SILU>(pb.x); o.y *= spk_act_grad<SPK_ACT_SILU>(pb.y);
      o.z *= spk_act_grad<SPK_ACT_SILU>(pb.z); o.w *= spk_act_grad<SPK_ACT_SILU>(pb.w);
      mixs_store4(sT + el * LDT, F, f0, o);
    }
    __syncthreads();
    // ---- stage 2: g_ctx = t W1: the q part feeds dL/dq, the |V| part the norm term of dL/dV
    {
      f32x4 q0 = z4, n0 = z4, q0x = z4, n0x = z4;
      const _Float16* brow = sT + el * LDT;
      const bool more = tile + gridDim.x < ntiles;
      mixs_load_a(wb, a.w1t_s, KB1, pw, kw, 4, el, h);              mixs_mfma(wa, brow, F, 0, h, q0, q0x); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wa, a.w1t_s, KB1, pw + PW, kw, 0, el, h);         mixs_mfma(wb, brow, F, 1, h, q0, q0x); __builtin_amdgcn_sched_barrier(0);
      mixs_load_a(wb, a.w1t_s, KB1, pw + PW, kw, 4, el, h);         mixs_mfma(wa, brow, F, 0, h, n0, n0x); __builtin_amdgcn_sched_barrier(0);
      if (more) mixs_load_a(wa, a.w2t_s, KB3, pw, kw, 0, el, h);    mixs_mfma(wb, brow, F, 1, h, n0, n0x); __builtin_amdgcn_sched_barrier(0);
      if (valid) {
        *(f32x4*)(a.gq1 + m * F + f0) = gq4 + mixs_fold(q0, q0x);
        const f32x4 sc = mixs_fold(n0, n0x) * invn;
#pragma unroll
        for (int x = 0; x < 3; ++x) *(f32x4*)(a.gmix + (m * 3 + x) * 2 * F + f0) = gV[x] + sc * V[x];
      }
    }
    __syncthreads();
  }
}

// Measured (profiles/README.md, round 2): the spill-free variant wins at every size tried (cfg 3: forward 45.9 -> 42.9 us,
// backward 34.3 -> 29.7 us; 32 k-atom box: 185 -> 166 us and 151 -> 130 us).  SPK_MIX_OCC=2 selects the two-workgroup form.
static bool mix_one_block_per_cu(int64_t ntiles) {
  static const char* env = getenv("SPK_MIX_OCC");
  (void)ntiles;
  return !(env && env[0] == '2');
}

// Eight waves per tile is the default at every size measured (cfg 3: forward 42.3 -> 38.9 us, backward 29.5 -> 26.8 us; 32 k-atom
// box: 167.6 -> 148.3 us and 128.3 -> 117.5 us); SPK_MIX_OCC = 1 / 2 select the four-wave forms (one / two workgroups per CU).
static bool mix_eight_waves(int64_t ntiles) {
  static const char* env = getenv("SPK_MIX_OCC");
  (void)ntiles;
  return !(env && (env[0] == '1' || env[0] == '2'));
}

static int launch_painn_mixing_bwd(const MixBwdArgs& a, int F, hipStream_t stream) {
  SPK_CHECK_ARG(F == 128, "fused PaiNN mixing backward: n_atom_basis must be 128");
  const size_t lds = sizeof(float) * (16 * (size_t)(3 * F + 4) + 16 * (size_t)(F + 4));
  const int64_t ntiles = (a.N + 15) / 16;
  const int grid = (int)(ntiles < 8192 ? ntiles : 8192);
  SpkProfScope prof("painn_mixing_bwd", stream);
  // few tiles per CU (molecule batches): one workgroup per CU, no scratch; many tiles per CU: two resident workgroups
  if (mix_eight_waves(ntiles) && a.w2t_s && a.w1t_s) hipLaunchKernelGGL((k_painn_mixing_bwd8s<128>), dim3(grid), dim3(512), lds, stream, a);
  else if (mix_eight_waves(ntiles)) hipLaunchKernelGGL((k_painn_mixing_bwd8<128>), dim3(grid), dim3(512), lds, stream, a);
  else if (mix_one_block_per_cu(ntiles)) hipLaunchKernelGGL((k_painn_mixing_bwd<128, 1>), dim3(grid), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL((k_painn_mixing_bwd<128, 2>), dim3(grid), dim3(256), lds, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

static int launch_painn_mixing_fwd(const MixFwdArgs& a, int F, hipStream_t stream) {
  SPK_CHECK_ARG(F == 128, "fused PaiNN mixing: n_atom_basis must be 128");
  const size_t lds = sizeof(float) * (48 * (size_t)(F + 4) + 16 * (size_t)(2 * F + 4) + 16 * (size_t)(F + 4));
  const int64_t ntiles = (a.N + 15) / 16;
  const int grid = (int)(ntiles < 8192 ? ntiles : 8192);
  SpkProfScope prof("painn_mixing_fwd", stream);
  if (mix_eight_waves(ntiles) && a.wmix_s && a.w1_s && a.w2_s) hipLaunchKernelGGL((k_painn_mixing_fwd8s<128>), dim3(grid), dim3(512), lds, stream, a);
  else if (mix_eight_waves(ntiles)) hipLaunchKernelGGL((k_painn_mixing_fwd8<128>), dim3(grid), dim3(512), lds, stream, a);
  else if (mix_one_block_per_cu(ntiles)) hipLaunchKernelGGL((k_painn_mixing_fwd<128, 1>), dim3(grid), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL((k_painn_mixing_fwd<128, 2>), dim3(grid), dim3(256), lds, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

#define SPK_EW_GRID(n) dim3(spk_grid_for((n), 256, spk_num_cus() * 16)), dim3(256), 0, stream

extern "C" int spk_painn_mix_ctx_f32(const float* q, const float* mix, int64_t N, int32_t F, float eps,
                                     float* ctx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(q && mix && ctx && N > 0 && F > 0, "spk_painn_mix_ctx_f32: bad input");
  hipLaunchKernelGGL(k_mix_ctx, SPK_EW_GRID(N * F), q, mix, N, F, eps, ctx);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_painn_mix_update_f32(const float* q, const float* mu, const float* mix,
                                        const float* a, int64_t N, int32_t F, float* q_out,
                                        float* mu_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(q && mu && mix && a && q_out && mu_out && N > 0 && F > 0, "spk_painn_mix_update_f32: bad input");
  hipLaunchKernelGGL(k_mix_update, SPK_EW_GRID(N * F), q, mu, mix, a, N, F, q_out, mu_out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_painn_mix_update_bwd_f32(const float* mu, const float* mix, const float* a,
                                            const float* gq_out, const float* gmu_out, int64_t N,
                                            int32_t F, float* ga, float* gmix, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  (void)mu;
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(mix && a && gq_out && gmu_out && ga && gmix && N > 0 && F > 0, "spk_painn_mix_update_bwd_f32: bad input");
  hipLaunchKernelGGL(k_mix_update_bwd, SPK_EW_GRID(N * F), mix, a, gq_out, gmu_out, N, F, ga, gmix);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_painn_mix_ctx_bwd_f32(const float* mix, const float* g_ctx, const float* gq_out,
                                         int64_t N, int32_t F, float eps, float* gmix_inout,
                                         float* gq, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(mix && g_ctx && gq_out && gmix_inout && gq && N > 0 && F > 0, "spk_painn_mix_ctx_bwd_f32: bad input");
  hipLaunchKernelGGL(k_mix_ctx_bwd, SPK_EW_GRID(N * F), mix, g_ctx, gq_out, N, F, eps, gmix_inout, gq);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------
// whole-representation drivers (representation/painn.py:207-256), eval-mode force path
// ------------------------------------------------------------------------------------------
#define SPK_TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

#include "spk_pack.h"
// order of the packed images in spk_painn_t::wpack: per interaction ctx.0, ctx.1, mu_channel_mix, intra ctx.0, intra ctx.1
static bool painn_pack_shapes_ok(const spk_painn_t* m) { return m->n_atom_basis == 128; }   // widths F, 2F, 3F <= 384, all % 128
static SpkPackTable painn_pack_table(const spk_painn_t* m) {
  SpkPackTable T;
  const int F = m->n_atom_basis;
  for (int l = 0; l < m->n_interactions; ++l) {
    const spk_painn_layer_t& P = m->layers[l];
    T.add(P.ctx_w1, P.ctx_w1T, F, F);
    T.add(P.ctx_w2, P.ctx_w2T, 3 * F, F);
    T.add(P.mix_w, P.mix_wT, 2 * F, F);
    T.add(P.ictx_w1, P.ictx_w1T, F, 2 * F);
    T.add(P.ictx_w2, P.ictx_w2T, 3 * F, F);
  }
  T.base = m->wpack;
  return T;
}
extern "C" int64_t spk_painn_packed_floats(const spk_painn_t* m) {
  if (!m || !m->layers || m->n_interactions <= 0 || !painn_pack_shapes_ok(m)) return 0;
  return painn_pack_table(m).total;
}
extern "C" int spk_painn_pack_weights_f32(const spk_painn_t* m, float* wpack, void* stream) {
  SPK_CHECK_ARG(m && wpack && spk_painn_packed_floats(m) > 0, "spk_painn_pack_weights_f32: model shapes have no packed form (see spk_painn_packed_floats)");
  return spk_pack_all(painn_pack_table(m), wpack, (hipStream_t)stream);
}

static spk_chain_layer_t mk_layer(const float* w, const float* b, const float* res, float* out, float* pre_out,
                                  const float* post_pre, int k, int n_out, int act, int trans, int post_act) {
  spk_chain_layer_t L;
  L.w = w; L.b = b; L.res = res; L.out = out; L.pre_out = pre_out; L.post_pre = post_pre;
  L.k = k; L.n_out = n_out; L.act = act; L.trans = trans; L.post_act = post_act;
  return L;
}

static spk_chain_layer_t mk_fwd(const float* w, const float* wT, const float* b, float* out, float* pre_out,
                                int k, int n_out, int act) {
  return wT ? mk_layer(wT, b, nullptr, out, pre_out, nullptr, k, n_out, act, 1, 0)
            : mk_layer(w, b, nullptr, out, pre_out, nullptr, k, n_out, act, 0, 0);
}

// molecule-resident kernels (spk_painn_mol.hip): block-diagonal lists with <= 32 atoms per block, F = 128
#include "spk_painn_mol.h"

static bool painn_tabulated(const spk_painn_t* m) {
  const float* tab; int nk; float dmax;
  return m && m->layers && m->n_interactions > 0 && spk_filter_table_lookup(m->layers[0].filt_w, &tab, &nk, &dmax);
}

// per layer saved for backward: preA [F] | c [3F] | mu_in [3F] | mix [6F] | preB [F] | a [3F]
static inline int64_t painn_saved_per_atom(int F) { return 17 * (int64_t)F; }

extern "C" int64_t spk_painn_saved_floats(const spk_painn_t* m, int64_t N) {
  if (!m) return 0;
  return (int64_t)m->n_interactions * N * painn_saved_per_atom(m->n_atom_basis);
}
extern "C" int64_t spk_painn_scratch_floats(const spk_painn_t* m, int64_t N) {
  if (!m) return 0;
  return N * 24 * (int64_t)m->n_atom_basis;
}

// per-call edge tables of the block kernels (spk_painn_blk.hip): true if the message launches of this call will use them
static bool painn_blocks_prepare(const spk_graph_t* g, const spk_radial_t* rb, const float* r_ij, int F, bool bwd, hipStream_t stream) {
  if (!g->blocks || !r_ij || g->n_edges == 0 || !(g->sorted && g->rowptr) || (bwd && !g->symmetric)) return false;
  if (spk_get_variant() == SPK_VARIANT_SIMPLE) return false;
  MsgArgs a = {};
  a.rij = r_ij; a.rb = spk_radial_dev(rb); a.blocks = g->blocks; a.E = g->n_edges; a.N = g->n_atoms; a.F = F; a.rowptr = g->rowptr;
  if (!spk_painn_blk_ok(a, bwd)) return false;
  return spk_painn_blk_prep(a, stream) == SPK_OK;
}

extern "C" int spk_painn_forward_f32(const spk_painn_t* m, const spk_graph_t* g,
                                     const spk_radial_t* rb, const float* q0, const float* r_ij,
                                     float* q_out, float* mu_out, float* saved, float* scratch,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_painn_forward_f32";
  SPK_CHECK_ARG(m && m->layers && g && rb, "%s: null argument", who);
  const SpkPackTable ptab = (m->wpack && painn_pack_shapes_ok(m)) ? painn_pack_table(m) : SpkPackTable();
  const int64_t N = g->n_atoms;
  const int F = m->n_atom_basis, L = m->n_interactions;
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(q0 && q_out && mu_out && (L == 0 || (saved && scratch)), "%s: null buffer", who);
  const size_t nf = (size_t)N * F;
  if (L == 0) {
    SPK_HIP_TRY(hipMemcpyAsync(q_out, q0, nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
    { int _zr = spk_zero_async(mu_out, 3 * nf * sizeof(float), stream); if (_zr) return _zr; }
    return SPK_OK;
  }
  if (ptab.base && r_ij && !getenv("SPK_NO_PAINN_MOL_FWD") && !painn_tabulated(m) && spk_painn_mol_eligible(m, g, rb))      // batches of small molecules: the whole forward is ONE launch
    return spk_painn_mol_forward(m, g, rb, ptab, q0, r_ij, q_out, mu_out, saved, stream);
  float* c1 = scratch;            // [N,F]
  float* q1 = c1 + nf;            // [N,F]
  float* mu1 = q1 + nf;           // [N,3F]
  float* ctx = mu1 + 3 * nf;      // [N,2F]
  float* a1 = ctx + 2 * nf;       // [N,F]
  const int64_t per = painn_saved_per_atom(F) * N;
  const bool blk = painn_blocks_prepare(g, rb, r_ij, F, false, stream);     // per-call edge tables of the block kernels, once for all interactions
  // mu entering the first interaction is zero (painn.py:246)
  { int _zr = spk_zero_async(saved + 4 * nf, 3 * nf * sizeof(float), stream); if (_zr) return _zr; }
  for (int l = 0; l < L; ++l) {
    const spk_painn_layer_t& P = m->layers[l];
    float* S = saved + l * per;
    float* preA = S; float* c = S + nf; float* mu_in = S + 4 * nf; float* mix = S + 7 * nf;
    float* preB = S + 13 * nf; float* av = S + 14 * nf;
    const float* qin = (l == 0) ? q0 : q_out;
    {  // c = ctx_w2 silu(ctx_w1 q + b1) + b2   (one launch)
      spk_chain_t ch = {};
      ch.n_layers = 2; ch.m = N; ch.in = qin; ch.tmp[0] = c1; ch.tmp[1] = a1;
      ch.layers[0] = mk_fwd(P.ctx_w1, P.ctx_w1T, P.ctx_b1, nullptr, preA, F, F, SPK_ACT_SILU);
      ch.layers[1] = mk_fwd(P.ctx_w2, P.ctx_w2T, P.ctx_b2, c, nullptr, F, 3 * F, SPK_ACT_NONE);
      spk_apply_pack(ch, ptab);
      SPK_TRY(spk_dense_chain_f32(&ch, stream));
    }
    SPK_TRY(spk_painn_message_fwd_internal(g, rb, c, qin, mu_in, r_ij, P.filt_w, P.filt_b, F, q1, mu1, stream, l == 0, blk));   // mu_in == 0 for l == 0
    float* mu_next = (l == L - 1) ? mu_out : (saved + (l + 1) * per + 4 * nf);
    const float* pk_mix = spk_packed_of(ptab, P.mix_w, 0);
    const float* pk_w1 = spk_packed_of(ptab, P.ictx_w1, 0);
    const float* pk_w2 = spk_packed_of(ptab, P.ictx_w2, 0);
    if (F == 128 && pk_mix && pk_w1 && pk_w2) {   // the whole PaiNNMixing block in one launch
      MixFwdArgs ma;
      ma.q1 = q1; ma.mu1 = mu1; ma.wmix = pk_mix; ma.w1 = pk_w1; ma.b1 = P.ictx_b1; ma.w2 = pk_w2; ma.b2 = P.ictx_b2;
      ma.eps = m->epsilon; ma.N = N; ma.mix = mix; ma.preB = preB; ma.a = av; ma.q_out = q_out; ma.mu_out = mu_next;
      ma.wmix_s = spk_packed_split_of(ptab, P.mix_w, 0); ma.w1_s = spk_packed_split_of(ptab, P.ictx_w1, 0); ma.w2_s = spk_packed_split_of(ptab, P.ictx_w2, 0);
      SPK_TRY(launch_painn_mixing_fwd(ma, F, stream));
      continue;
    }
    {  // mix = mu1 W_mix^T over [3N, F]
      spk_chain_t ch = {};
      ch.n_layers = 1; ch.m = 3 * N; ch.in = mu1;
      ch.layers[0] = mk_fwd(P.mix_w, P.mix_wT, nullptr, mix, nullptr, F, 2 * F, SPK_ACT_NONE);
      spk_apply_pack(ch, ptab);
      SPK_TRY(spk_dense_chain_f32(&ch, stream));
    }
    SPK_TRY(spk_painn_mix_ctx_f32(q1, mix, N, F, m->epsilon, ctx, stream));
    {  // a = ictx_w2 silu(ictx_w1 ctx + b1) + b2
      spk_chain_t ch = {};
      ch.n_layers = 2; ch.m = N; ch.in = ctx; ch.tmp[0] = c1; ch.tmp[1] = a1;
      ch.layers[0] = mk_fwd(P.ictx_w1, P.ictx_w1T, P.ictx_b1, nullptr, preB, 2 * F, F, SPK_ACT_SILU);
      ch.layers[1] = mk_fwd(P.ictx_w2, P.ictx_w2T, P.ictx_b2, av, nullptr, F, 3 * F, SPK_ACT_NONE);
      spk_apply_pack(ch, ptab);
      SPK_TRY(spk_dense_chain_f32(&ch, stream));
    }
    SPK_TRY(spk_painn_mix_update_f32(q1, mu1, mix, av, N, F, q_out, mu_next, stream));
  }
  return SPK_OK;
}

extern "C" int spk_painn_backward_f32(const spk_painn_t* m, const spk_graph_t* g,
                                      const spk_radial_t* rb, const float* gq_out,
                                      const float* gmu_out, const float* r_ij, const float* saved,
                                      float* scratch, float* gr, float* gq0, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_painn_backward_f32";
  SPK_CHECK_ARG(m && m->layers && g && rb, "%s: null argument", who);
  const SpkPackTable ptab = (m->wpack && painn_pack_shapes_ok(m)) ? painn_pack_table(m) : SpkPackTable();
  const int64_t N = g->n_atoms, E = g->n_edges;
  const int F = m->n_atom_basis, L = m->n_interactions;
  if (N > 0 && E > 0 && L > 0 && gr && r_ij && saved && scratch && (gq_out || gmu_out) && ptab.base && !painn_tabulated(m) && spk_painn_mol_bwd_eligible(m, g, rb))
    return spk_painn_mol_backward(m, g, rb, ptab, gq_out, gmu_out, r_ij, saved, scratch, gr, gq0, stream);      // ONE launch, every gr entry written once
  if (E > 0) {
    SPK_CHECK_ARG(gr != nullptr, "%s: null gr", who);
    { int _zr = spk_zero_async(gr, (size_t)E * 3 * sizeof(float), stream); if (_zr) return _zr; }
  }
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG((gq_out || gmu_out) && (L == 0 || (saved && scratch)), "%s: null buffer", who);
  const size_t nf = (size_t)N * F;
  // scratch layout
  float* ga = scratch;             // [N,3F]
  float* gmix = ga + 3 * nf;       // [N,6F]
  float* ga1 = gmix + 6 * nf;      // [N,F]
  float* gctx = ga1 + nf;          // [N,2F]
  float* gq1 = gctx + 2 * nf;      // [N,F]
  float* gmu1 = gq1 + nf;          // [N,3F]
  float* gc = gmu1 + 3 * nf;       // [N,3F]
  float* gc1 = gc + 3 * nf;        // [N,F]
  float* gq = gc1 + nf;            // [N,F]   running dL/dq
  float* gmu = gq + nf;            // [N,3F]  running dL/dmu
  if (gq_out) SPK_HIP_TRY(hipMemcpyAsync(gq, gq_out, nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
  else { int _zr = spk_zero_async(gq, nf * sizeof(float), stream); if (_zr) return _zr; }
  if (gmu_out) SPK_HIP_TRY(hipMemcpyAsync(gmu, gmu_out, 3 * nf * sizeof(float), hipMemcpyDeviceToDevice, stream));
  else { int _zr = spk_zero_async(gmu, 3 * nf * sizeof(float), stream); if (_zr) return _zr; }
  const int64_t per = painn_saved_per_atom(F) * N;
  const bool blk = painn_blocks_prepare(g, rb, r_ij, F, true, stream);
  for (int l = L - 1; l >= 0; --l) {
    const spk_painn_layer_t& P = m->layers[l];
    const float* S = saved + l * per;
    const float* preA = S; const float* c = S + nf; const float* mu_in = S + 4 * nf; const float* mix = S + 7 * nf;
    const float* preB = S + 13 * nf; const float* av = S + 14 * nf;
    // ---- mixing backward
    const float* pk_w2t = spk_packed_of(ptab, P.ictx_w2, 1);
    const float* pk_w1t = spk_packed_of(ptab, P.ictx_w1, 1);
    const bool fused_mix = (F == 128 && pk_w2t && pk_w1t);
    if (fused_mix) {   // update gradient + context net (transposed) + norm term in one launch
      MixBwdArgs mb;
      mb.gq = gq; mb.gmu = gmu; mb.mix = mix; mb.a = av; mb.preB = preB; mb.w2t = pk_w2t; mb.w1t = pk_w1t;
      mb.eps = m->epsilon; mb.N = N; mb.gq1 = gq1; mb.gmix = gmix;
      mb.w2t_s = spk_packed_split_of(ptab, P.ictx_w2, 1); mb.w1t_s = spk_packed_split_of(ptab, P.ictx_w1, 1);
      SPK_TRY(launch_painn_mixing_bwd(mb, F, stream));
    } else {
    SPK_TRY(spk_painn_mix_update_bwd_f32(nullptr, mix, av, gq, gmu, N, F, ga, gmix, stream));
    {  // g_ctx = ((ga W_d) * silu'(preB)) W_c
      spk_chain_t ch = {};
      ch.n_layers = 2; ch.m = N; ch.in = ga; ch.tmp[0] = ga1; ch.tmp[1] = gc1;
      ch.layers[0] = mk_layer(P.ictx_w2, nullptr, nullptr, nullptr, nullptr, preB, 3 * F, F, SPK_ACT_NONE, 1, SPK_ACT_SILU);
      ch.layers[1] = mk_layer(P.ictx_w1, nullptr, nullptr, gctx, nullptr, nullptr, F, 2 * F, SPK_ACT_NONE, 1, 0);
      spk_apply_pack(ch, ptab);
      SPK_TRY(spk_dense_chain_f32(&ch, stream));
    }
    SPK_TRY(spk_painn_mix_ctx_bwd_f32(mix, gctx, gq, N, F, m->epsilon, gmix, gq1, stream));
    }
    {  // mu1 -> mix is a bias-free Dense over [3N, F]; residual path adds gmu
      spk_chain_t ch = {};
      ch.n_layers = 1; ch.m = 3 * N; ch.in = gmix;
      ch.layers[0] = mk_layer(P.mix_w, nullptr, gmu, gmu1, nullptr, nullptr, 2 * F, F, SPK_ACT_NONE, 1, 0);
      spk_apply_pack(ch, ptab);
      SPK_TRY(spk_dense_chain_f32(&ch, stream));
    }
    // ---- message backward: gc, gmu (incl. residual), gr +=
    // (first interaction without dL/dq0, the eval path: only the geometry gradient is formed and the context-net
    //  backward below it is skipped)
    const bool last_geom_only = (l == 0 && !gq0);
    SPK_TRY(spk_painn_message_bwd_internal(g, rb, c, mu_in, gq1, gmu1, r_ij, P.filt_w, P.filt_b, F, gc, gmu, gr, stream, last_geom_only, l == 0, blk));
    if (last_geom_only) break;
    {  // context net backward; residual path adds gq1
      float* out = (l == 0 && gq0) ? gq0 : gq;
      spk_chain_t ch = {};
      ch.n_layers = 2; ch.m = N; ch.in = gc; ch.tmp[0] = ga1; ch.tmp[1] = gc1;
      ch.layers[0] = mk_layer(P.ctx_w2, nullptr, nullptr, nullptr, nullptr, preA, 3 * F, F, SPK_ACT_NONE, 1, SPK_ACT_SILU);
      ch.layers[1] = mk_layer(P.ctx_w1, nullptr, gq1, out, nullptr, nullptr, F, F, SPK_ACT_NONE, 1, 0);
      spk_apply_pack(ch, ptab);
      SPK_TRY(spk_dense_chain_f32(&ch, stream));
    }
  }
  return SPK_OK;
}


// ------------------------------------------------------------------------------------------------------------------------------
// The standard potential  PairwiseDistances -> PaiNN -> Atomwise(sum) -> Forces  on batches of small molecules: TWO launches
// (atomistic/distances.py:14-26, representation/painn.py:207-256, atomistic/atomwise.py:69-88, atomistic/response.py:59-76).
static bool painn_potential_ok(const spk_painn_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb) {
  if (!m || !head || !g || !rb || !m->layers || m->n_interactions <= 0) return false;
  if (painn_tabulated(m)) return false;          // (experiment) registered filter tables: the stage-by-stage path runs the table kernels
  if (!m->wpack || !painn_pack_shapes_ok(m)) return false;
  if (!head->w1 || !head->w1t || !head->b1 || !head->w2 || head->n_hidden != 64) return false;
  if (head->act != SPK_ACT_SSP && head->act != SPK_ACT_SILU) return false;
  if (getenv("SPK_NO_POTENTIAL") || getenv("SPK_NO_PAINN_MOL_FWD") || !g->idx_i || !g->rev) return false;
  return spk_painn_mol_eligible(m, g, rb) && spk_painn_mol_bwd_eligible(m, g, rb);
}
extern "C" int spk_painn_potential_supported(const spk_painn_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb) {
  return painn_potential_ok(m, head, g, rb) ? 1 : 0;
}

// Energies and FORCES (= -dE/dR of the summed energy) in the two launches, nothing else: q0 may be NULL (the rows of the nuclear
// embedding table `emb` [n_types, F] are looked up by Z inside the forward launch).  all_inside != 0: the caller guarantees that every
// molecule's atoms lie inside ONE group of the plan and that every molecule has an atom -- the energies are then stored, not
// accumulated, and E needs no clearing launch.  scratch: spk_painn_scratch_floats() floats.
extern "C" int spk_painn_potential_forces_f32(const spk_painn_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb,
                                              const float* q0, const float* emb, const int64_t* Z, int32_t n_types, const float* R,
                                              const float* offsets, const int64_t* idx_m, int64_t n_mol, int32_t all_inside, float* q_out,
                                              float* mu_out, float* E, float* F, float* pre_h, float* saved, float* scratch, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_painn_potential_forces_f32";
  SPK_CHECK_ARG(painn_potential_ok(m, head, g, rb), "%s: model / list not covered by the fused potential (see spk_painn_potential_supported)", who);
  SPK_CHECK_ARG(n_mol >= 0 && (n_mol == 0 || E), "%s: null energy buffer", who);
  SPK_CHECK_ARG(q0 || (emb && Z && n_types > 0), "%s: neither features nor an embedding table", who);
  if (n_mol > 0 && !all_inside) { int zr = spk_zero_async(E, (size_t)n_mol * sizeof(float), stream); if (zr) return zr; }
  if (g->n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(R && idx_m && q_out && mu_out && F && pre_h && saved && scratch, "%s: null buffer", who);
  PmHeadDev h;
  h.w1 = head->w1; h.w1t = head->w1t; h.b1 = head->b1; h.w2 = head->w2; h.b2 = head->b2; h.H = head->n_hidden; h.act = head->act;
  h.idx_m = idx_m; h.E = E; h.pre_h = pre_h; h.direct_store = all_inside ? 1 : 0;
  const SpkPackTable ptab = painn_pack_table(m);
  // scratch: [2][N, 3F] rows of the message backward, then the pair vectors [E, 3] (3 E <= 93 N floats of the 18 N F left)
  //          and dE/dq_L [N, F], the gradient of the summed energy through the head -- both written by the forward launch
  const size_t nf = (size_t)g->n_atoms * m->n_atom_basis;
  SPK_CHECK_ARG(3 * (size_t)g->n_edges <= nf, "%s: more than n_atom_basis / 3 neighbours per atom on average", who);
  float* rij = scratch + 6 * nf;
  float* gq = scratch + 7 * nf;
  SPK_TRY(spk_painn_mol_forward_ex(m, g, rb, ptab, q0, nullptr, R, offsets, q0 ? nullptr : emb, Z, n_types, &h, rij, gq, q_out, mu_out, saved, stream));
  return spk_painn_mol_backward_ex(m, g, rb, ptab, gq, nullptr, rij, saved, scratch, nullptr, nullptr, F, stream);
}
