// Kernels of the force-matching gradient engine (spk_fm_engine.h): element-wise and row kernels of the four passes
//   A values | B reverse w.r.t. positions (forces) | C tangents along t = -dL/dF | D reverse of the dual graph (weight gradients)
// of SchNet (representation/schnet.py:54-69, 147-173) and PaiNN (representation/painn.py:31-117, 207-256) with the energy head
// (atomistic/atomwise.py:69-88) and Forces (atomistic/response.py:59-76).  Formulas: oracle/fm_oracle.py (pinned to autograd).
//
// Written once for two compilers: hipcc (gfx950: grid-stride threads / one wavefront per row) and, with SPK_FM_EMU defined, a
// plain C++ compiler that runs every kernel as serial loops -- TEST INFRASTRUCTURE (tests/fm_emu): the engine's host
// orchestration and these formulas are checked in float64 on the build box, the product only ever runs the HIP instantiation.
//
// Layout conventions: "stacked" buffers X2 are [2n, W]: n value rows, then n tangent (or d-derivative, or tangent-cotangent)
// rows.  A trailing 1 = derivative w.r.t. the pair distance, t = tangent, g* = cotangent of a value, h* = cotangent of a tangent.
#pragma once
#include <stdint.h>
#include <math.h>

#ifdef SPK_FM_EMU
#define FM_KERNEL static inline
#define FM_HD static inline
#define FM_UNROLL
#define FM_R
#define FM_ATOMIC_MAX(p, v) do { if (*(p) < (v)) *(p) = (v); } while (0)
#define FM_ATOMIC_OR(p, v) do { *(p) |= (v); } while (0)
#define FM_FOR(t, total) for (int64_t t = 0; t < (int64_t)(total); ++t)
#define FM_FOR_ROWS(row, rows) for (int64_t row = 0; row < (int64_t)(rows); ++row)
#define FM_FOR_LANES(f, F) for (int f = 0; f < (int)(F); ++f)
#define FM_WAVE_SUM(x) (x)
#define FM_IF_LANE0
#define FM_LAST_LANE_WITH(cond) (cond)
// slotted row loops (see the device definitions): the emulation has one slot
#define FM_NSLOT 1
#define FM_SLOT 0
#define FM_FOR_SLOTTED(t, total) for (int64_t t = 0, fm_ok = 1; t < (int64_t)(total); ++t)
#define FM_SLOT_SUM(arr, n)
#define FM_SLOT_OWNER (fm_ok != 0)
#else
#define FM_KERNEL __global__ __launch_bounds__(256)
#define FM_HD __device__ __forceinline__
#define FM_UNROLL _Pragma("unroll")
#define FM_R __restrict__
#define FM_ATOMIC_MAX(p, v) atomicMax((p), (v))
#define FM_ATOMIC_OR(p, v) atomicOr((p), (v))
#define FM_FOR(t, total) for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < (int64_t)(total); t += (int64_t)gridDim.x * blockDim.x)
#define FM_FOR_ROWS(row, rows) for (int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6); row < (int64_t)(rows); row += (int64_t)gridDim.x * (blockDim.x >> 6))
#define FM_FOR_LANES(f, F) for (int f = threadIdx.x & 63; f < (int)(F); f += 64)
#define FM_WAVE_SUM(x) fm_wave_sum(x)
#define FM_IF_LANE0 if ((threadIdx.x & 63) == 0)
// true on the highest active lane of the wavefront on which `cond` holds (one atomic per wavefront instead of one per lane: 2 560 atomic
// maxima on ONE address cost k_fm_geom ~10 of its 14 us -- they serialise at the L2)
#define FM_LAST_LANE_WITH(cond) fm_last_lane_with(cond)
// Slotted row loops: the pairs of a CSR row are shared by FM_NSLOT threads -- the four waves of a workgroup; lane l of every wave works on
// item 64 * block + l, wave w takes the chunks w, w + 4, ... of the item's row, the partial sums meet in LDS in slot order (deterministic)
// and wave 0 writes.  An 8-frame training batch has ~15 pairs per atom: with one thread per (atom, channel) the PaiNN message kernels were
// 84 workgroups each walking four chunks of ~25 dependent-stage gathers (10 - 17 us a launch); slotted they are 336 workgroups with one
// chunk per thread.  Items past the end are clamped to the last one (every thread reaches the barriers) and do not write.
#define FM_NSLOT 4
#define FM_SLOT ((int)(threadIdx.x >> 6))
#define FM_FOR_SLOTTED(t, total)                                                                                                       \
  for (int64_t fm_base = (int64_t)blockIdx.x * 64; fm_base < (int64_t)(total); fm_base += (int64_t)gridDim.x * 64)                    \
    for (int64_t fm_t = fm_base + (threadIdx.x & 63), fm_ok = fm_t < (int64_t)(total), t = fm_ok ? fm_t : (int64_t)(total) - 1, fm_once = 1; fm_once; fm_once = 0)
#define FM_SLOT_MAXV 12
#define FM_SLOT_SUM(arr, n)                                                                                                            \
  do {                                                                                                                                 \
    __shared__ float fm_red[FM_SLOT_MAXV][FM_NSLOT][64];                                                                               \
    const int fm_w = threadIdx.x >> 6, fm_l = threadIdx.x & 63;                                                                        \
    _Pragma("unroll") for (int fm_q = 0; fm_q < (n); ++fm_q) fm_red[fm_q][fm_w][fm_l] = (arr)[fm_q];                                 \
    __syncthreads();                                                                                                                   \
    _Pragma("unroll") for (int fm_q = 0; fm_q < (n); ++fm_q)                                                                          \
      (arr)[fm_q] = ((fm_red[fm_q][0][fm_l] + fm_red[fm_q][1][fm_l]) + fm_red[fm_q][2][fm_l]) + fm_red[fm_q][3][fm_l];              \
    __syncthreads();                                                                                                                   \
  } while (0)
#define FM_SLOT_OWNER (fm_ok != 0 && (threadIdx.x >> 6) == 0)
__device__ __forceinline__ bool fm_last_lane_with(bool cond) {
  const unsigned long long m = __ballot(cond);
  return cond && (int)(threadIdx.x & 63) == 63 - __builtin_clzll(m);
}
__device__ __forceinline__ float fm_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
#endif

// Gathers of the chunked loops: the address is made valid (index 0 when the pair does not count) and the LOAD IS UNCONDITIONAL, the
// value is selected afterwards.  Written as `ok ? p[i] : 0` the compiler guards every load with its own branch and waits for memory inside
// them: the gathers of a chunk then leave one after the other (k_fm_cfconv_T_dual: 10.7 us for four chunks of three dependent stages).
template <class T, class I>
FM_HD T fm_ld(bool ok, const T* p, I idx) {
  const T v = p[ok ? idx : I(0)];
  return ok ? v : T(0);
}
template <class T, class I, class D>
FM_HD T fm_ldi(bool ok, const T* p, I idx, D dflt) {
  const T v = p[ok ? idx : I(0)];
  return ok ? v : (T)dflt;
}
#define FM_LD(ok, p, idx) fm_ld((ok), (p), (idx))
// p may be NULL (uniform over the launch): the load then goes to `alt` (any non-NULL array of T) and is masked -- a pointer SELECT, not a
// branch around the load (twelve uniform branches inside a chunk cost the PaiNN message kernels a factor of two)
#define FM_LDN(ok, p, idx, alt) fm_ld((ok) && (p) != nullptr, (p) ? (p) : (alt), (idx))
#define FM_LDI(ok, p, idx, dflt) fm_ldi((ok), (p), (idx), (dflt))
// The PaiNN message kernels gather ~100 values per chunk.  There the unconditional form ALONE made the scheduler interleave loads and uses
// to bound the live ranges (waits with one or two loads in flight: k_fm_painn_msg_T_dual 17 -> 34 us); with a scheduling fence between the
// gathers of a chunk and their uses the ~46 loads leave together (PaiNN step 0.767 -> 0.743 ms).  SPK_FM_PAINN_UNCOND (set by spk_fm.hip)
// selects that form; without it these kernels keep the plain conditional loads.
#if defined(SPK_FM_EMU) || !defined(SPK_FM_PAINN_UNCOND)
#define FM_LDC(ok, p, idx) ((ok) ? (p)[idx] : T(0))
#define FM_LDNC(ok, p, idx) ((ok) && (p) ? (p)[idx] : T(0))
#define FM_LDIC(ok, p, idx, dflt) ((ok) ? (p)[idx] : (dflt))
#define FM_FENCE()
#else
#define FM_LDC(ok, p, idx) fm_ld((ok), (p), (idx))
#define FM_LDNC(ok, p, idx) fm_ld((ok) && (p) != nullptr, (p) ? (p) : (const T*)gathers_alt, (idx))
#define FM_LDIC(ok, p, idx, dflt) fm_ldi((ok), (p), (idx), (dflt))
#define FM_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// Row / column loops keep FM_CH (light kernels) or FM_CH4 (PaiNN message) pairs in flight: index loads, then every gather of the chunk, then
// the sums IN PAIR ORDER (the result does not depend on the chunking).  At training sizes (~15 pairs per atom, everything L2 resident)
// a loop with one dependent gather per iteration is a chain of ~0.5 us round trips; chunked, a row costs two or three.
#define FM_CH 8
#define FM_CH4 4

#define FM_ACT_NONE 0
#define FM_ACT_SSP 1
#define FM_ACT_SILU 2

FM_HD float fm_exp(float x) { return expf(x); }
FM_HD double fm_exp(double x) { return exp(x); }
FM_HD float fm_log1p(float x) { return log1pf(x); }
FM_HD double fm_log1p(double x) { return log1p(x); }
FM_HD float fm_sqrt(float x) { return sqrtf(x); }
FM_HD double fm_sqrt(double x) { return sqrt(x); }
FM_HD void fm_sincos(float x, float& s, float& c) { s = sinf(x); c = cosf(x); }
FM_HD void fm_sincos(double x, double& s, double& c) { s = sin(x); c = cos(x); }

template <class T>
FM_HD T fm_sigmoid(T x) {
  const T t = fm_exp(x >= 0 ? -x : x);
  return x >= 0 ? T(1) / (T(1) + t) : t / (T(1) + t);
}
// k-th derivative (k = 0, 1, 2) of shifted softplus (nn/activations.py:9-22) / SiLU
template <class T>
FM_HD T fm_act(int act, int order, T z) {
  if (act == FM_ACT_NONE) return order == 0 ? z : (order == 1 ? T(1) : T(0));
  const T s = fm_sigmoid(z);
  if (act == FM_ACT_SSP) {
    if (order == 0) return (z > 0 ? z : T(0)) + fm_log1p(fm_exp(z >= 0 ? -z : z)) - T(0.69314718055994530942);
    return order == 1 ? s : s * (T(1) - s);
  }
  if (order == 0) return z * s;
  if (order == 1) return s * (T(1) + z * (T(1) - s));
  return s * (T(1) - s) * (T(2) + z * (T(1) - T(2) * s));
}

template <class T>
struct FmRadial {   // nn/radial.py:18-48 (Gaussian: p0 = offsets, p1 = widths), :82-110 (Bessel: p0 = freqs); nn/cutoff.py:14-57
  int kind, n_rbf;
  const T* p0;
  const T* p1;
  T cutoff;
};

// ------------------------------------------------------------------------------------------------ geometry
// d, u = r / d, f_c, f_c', phi2 = [phi ; phi'] for every pair (atomistic/distances.py:14-26, nn/radial.py, nn/cutoff.py)
template <class T>
FM_KERNEL void k_fm_geom(const T* FM_R R, const T* FM_R off, const int64_t* FM_R ii, const int64_t* FM_R jj, int64_t E, int64_t N, FmRadial<T> rb, T* FM_R d_out,
                         T* FM_R u_out, T* FM_R fc_out, T* FM_R fc1_out, T* FM_R phi2, int32_t* e_act) {
  const T* FM_R p0 = rb.p0;
  const T* FM_R p1 = rb.p1;
  FM_FOR(e, E) {
    int64_t i = ii[e], j = jj[e];
    const bool ok = (uint64_t)i < (uint64_t)N && (uint64_t)j < (uint64_t)N;
    if (!ok) { i = 0; j = 0; }
    T r[3];
    T d2 = 0;
    for (int x = 0; x < 3; ++x) {
      r[x] = R[j * 3 + x] - R[i * 3 + x] + (off ? off[e * 3 + x] : T(0));
      d2 += r[x] * r[x];
    }
    const T d = fm_sqrt(d2);
    const T inv = d > 0 ? T(1) / d : T(0);
    d_out[e] = d;
    for (int x = 0; x < 3; ++x) u_out[e * 3 + x] = r[x] * inv;
    T f = 0, f1 = 0;
    if (ok && d < rb.cutoff) {
      const T a = T(3.14159265358979323846) / rb.cutoff;
      T s, c;
      fm_sincos(a * d, s, c);
      f = T(0.5) * (c + T(1));
      f1 = -T(0.5) * a * s;
    }
    fc_out[e] = f;
    fc1_out[e] = f1;
    // e_act = 1 + the last pair inside the cutoff: the row / column loops stop there (pairs behind it -- the inert tail that pads a
    // static-shape batch, train.pad_edges -- contribute exactly zero and would otherwise make the last atom's row hundreds of pairs long)
    if (FM_LAST_LANE_WITH(ok && d < rb.cutoff)) FM_ATOMIC_MAX(e_act, (int32_t)(e + 1));      // (pair indices ascend with the lane)
  }
  // basis values and d-derivatives: one thread per (pair, k) -- with the basis loop inside the per-pair thread the 2 560 pairs of a training
  // batch were 40 wavefronts walking 20 exponentials and 40 strided stores each (12.6 us); d is recomputed here from cached loads
  const int K = rb.n_rbf;
  FM_FOR(t, E * K) {
    const int64_t e = t / K;
    const int k = (int)(t % K);
    int64_t i = ii[e], j = jj[e];
    if (!((uint64_t)i < (uint64_t)N && (uint64_t)j < (uint64_t)N)) { i = 0; j = 0; }
    T d2 = 0;
    for (int x = 0; x < 3; ++x) {
      const T r = R[j * 3 + x] - R[i * 3 + x] + (off ? off[e * 3 + x] : T(0));
      d2 += r * r;
    }
    const T d = fm_sqrt(d2);
    const T inv = d > 0 ? T(1) / d : T(0);
    T phi, phi1;
    if (rb.kind == 0) {
      const T w = p1[k];
      const T c = -T(0.5) / (w * w);
      const T tt = d - p0[k];
      phi = fm_exp(c * tt * tt);
      phi1 = T(2) * c * tt * phi;
    } else {
      const T fr = p0[k];
      T s, c;
      fm_sincos(fr * d, s, c);
      if (d == 0) { phi = s; phi1 = 0; }
      else { phi = s * inv; phi1 = (fr * c - phi) * inv; }
    }
    phi2[t] = phi;
    phi2[E * K + t] = phi1;
  }
}

// tangent of the geometry along t = -gF:  rt = t_j - t_i,  dt = u . rt,  ut = (rt - u dt) / d
template <class T>
FM_KERNEL void k_fm_tgeom(const T* gF, const int64_t* ii, const int64_t* jj, const T* d, const T* u, int64_t E, int64_t N, T* dt, T* ut) {
  FM_FOR(e, E) {
    int64_t i = ii[e], j = jj[e];
    if (!((uint64_t)i < (uint64_t)N && (uint64_t)j < (uint64_t)N)) { i = 0; j = 0; }
    T rt[3], s = 0;
    for (int x = 0; x < 3; ++x) {
      rt[x] = gF[i * 3 + x] - gF[j * 3 + x];
      s += u[e * 3 + x] * rt[x];
    }
    dt[e] = s;
    if (ut) {
      const T inv = d[e] > 0 ? T(1) / d[e] : T(0);
      for (int x = 0; x < 3; ++x) ut[e * 3 + x] = (rt[x] - u[e * 3 + x] * s) * inv;
    }
  }
}

// forces: gr_e = gd_e u_e + (gu_e - (gu_e . u_e) u_e) / d_e  (per pair, k_fm_gr);  F_a = -(sum_{e: j(e) = a} gr_e - sum_{e: i(e) = a} gr_e)
template <class T>
FM_KERNEL void k_fm_gr(const T* FM_R gd, const T* FM_R gu, const T* FM_R u, const T* FM_R d, int64_t E, T* FM_R gr) {
  FM_FOR(e, E) {
    const T ux = u[e * 3], uy = u[e * 3 + 1], uz = u[e * 3 + 2];
    T vx = gd[e] * ux, vy = gd[e] * uy, vz = gd[e] * uz;
    if (gu) {
      const T gx = gu[e * 3], gy = gu[e * 3 + 1], gz = gu[e * 3 + 2];
      const T dot = gx * ux + gy * uy + gz * uz;
      const T inv = d[e] > 0 ? T(1) / d[e] : T(0);
      vx += (gx - dot * ux) * inv; vy += (gy - dot * uy) * inv; vz += (gz - dot * uz) * inv;
    }
    gr[e * 3] = vx; gr[e * 3 + 1] = vy; gr[e * 3 + 2] = vz;
  }
}
template <class T>
FM_KERNEL void k_fm_force(const T* FM_R gr, const int32_t* FM_R rowptr, const int32_t* FM_R colptr, const int32_t* FM_R perm, const int32_t* FM_R e_act, int64_t N,
                          T* FM_R Fo) {
  const int ea = *e_act;
  FM_FOR(t, N * 3) {
    const int64_t a = t / 3;
    const int x = (int)(t % 3);
    T acc = 0;
    const int k1 = colptr[a + 1];
    for (int kb = colptr[a]; kb < k1; kb += FM_CH) {
      int e[FM_CH];
      T v[FM_CH];
      FM_UNROLL for (int q = 0; q < FM_CH; ++q) e[q] = FM_LDI(kb + q < k1, perm, kb + q, ea);
      FM_UNROLL for (int q = 0; q < FM_CH; ++q) v[q] = FM_LD(e[q] < ea, gr, (int64_t)e[q] * 3 + x);
      FM_UNROLL for (int q = 0; q < FM_CH; ++q) acc += v[q];
      if (e[FM_CH - 1] >= ea) break;
    }
    int e1 = rowptr[a + 1];
    if (e1 > ea) e1 = ea;
    for (int eb = rowptr[a]; eb < e1; eb += FM_CH) {
      T v[FM_CH];
      FM_UNROLL for (int q = 0; q < FM_CH; ++q) v[q] = FM_LD(eb + q < e1, gr, (int64_t)(eb + q) * 3 + x);
      FM_UNROLL for (int q = 0; q < FM_CH; ++q) acc -= v[q];
    }
    Fo[t] = -acc;
  }
}
// source atom of every entry of the by-neighbour CSR: csrc[k] = idx_i[perm[k]] (-1: out of range) -- one dependent load less in every column loop
template <class T>
FM_KERNEL void k_fm_colsrc(const int32_t* FM_R perm, const int64_t* FM_R ii, int64_t E, int64_t N, int32_t* FM_R csrc, T* unused) {
  (void)unused;
  FM_FOR(k, E) {
    const int64_t i = ii[perm[k]];
    csrc[k] = (uint64_t)i < (uint64_t)N ? (int32_t)i : -1;
  }
}

// ------------------------------------------------------------------------------------------------ generic element-wise
template <class T>
FM_KERNEL void k_fm_axpy(const T* x, int64_t n, T* y) { FM_FOR(t, n) y[t] += x[t]; }

// out_t = act'(pre) pre_t  (tangent through an activation)
template <class T>
FM_KERNEL void k_fm_act_tangent(const T* pre, const T* pret, int64_t n, int act, T* out) { FM_FOR(t, n) out[t] = fm_act(act, 1, pre[t]) * pret[t]; }

// reverse of (z = act(a), zt = act'(a) at):  ga = gz act'(a) + hz act''(a) at,  ha = hz act'(a);  stacked [2n] in, [2n] out
template <class T>
FM_KERNEL void k_fm_act_dual_bwd(const T* gz2, const T* pre2, int64_t n, int act, T* ga2) {
  FM_FOR(t, n) {
    const T a = pre2[t], at = pre2[n + t], gz = gz2[t], hz = gz2[n + t];
    const T a1 = fm_act(act, 1, a);
    ga2[t] = gz * a1 + hz * fm_act(act, 2, a) * at;
    ga2[n + t] = hz * a1;
  }
}

// in place on G2 = [g ; g1] ([2E, W]):  Wf = g f_c,  Wf1 = g1 f_c + g f_c'   (schnet.py:62, painn.py:232-236)
template <class T>
FM_KERNEL void k_fm_filter_fc(T* G2, const T* fc, const T* fc1, int64_t E, int W) {
  FM_FOR(t, E * W) {
    const int64_t e = t / W;
    const T g = G2[t], g1 = G2[E * W + t];
    G2[t] = g * fc[e];
    G2[E * W + t] = g1 * fc[e] + g * fc1[e];
  }
}

// out[i, h] = w[h] * (s ? s[idx ? idx[i] : i] : 1)
template <class T>
FM_KERNEL void k_fm_bcast_rows(const T* w, const T* s, const int64_t* idx, int64_t n, int H, int64_t n_s, T* out) {
  FM_FOR(t, n * H) {
    const int64_t i = t / H;
    T sc = 1;
    if (s) {
      const int64_t m = idx ? idx[i] : i;
      sc = (uint64_t)m < (uint64_t)n_s ? s[m] : T(0);
    }
    out[t] = w[t % H] * sc;
  }
}
// reverse of the dual head's last layer and activation in one pass: S = sum_i gE_i e_i + sum_i et_i gives the cotangents g_th = w2 gEa_i,
// h_th = w2 of (th, th_t); through th = act(a), th_t = act'(a) a_t (k_fm_act_dual_bwd):  [g_a ; h_a] stacked [2 n H]
template <class T>
FM_KERNEL void k_fm_head_dual_cot(const T* w2, const T* gEa, const T* pre2, int64_t n, int H, int act, T* gpre2) {
  FM_FOR(t, n * H) {
    const T hz = w2[t % H], gz = hz * gEa[t / H];
    const T a = pre2[t], at = pre2[n * H + t];
    const T a1 = fm_act(act, 1, a);
    gpre2[t] = gz * a1 + hz * fm_act(act, 2, a) * at;
    gpre2[n * H + t] = hz * a1;
  }
}
template <class T>
FM_KERNEL void k_fm_gather1(const T* s, const int64_t* idx, int64_t n, int64_t n_s, T* out, T* ones) {
  FM_FOR(i, n) {
    const int64_t m = idx[i];
    out[i] = (uint64_t)m < (uint64_t)n_s ? s[m] : T(0);
    if (ones) ones[i] = 1;
  }
}

// ------------------------------------------------------------------------------------------------ energy head
// e_i = th_i . w2 + b2   (one wavefront per atom)
template <class T>
FM_KERNEL void k_fm_rowdot_bias(const T* th, const T* w2, const T* b2, int64_t n, int H, T* e_atom) {
  FM_FOR_ROWS(i, n) {
    T acc = 0;
    FM_FOR_LANES(h, H) acc += th[i * H + h] * w2[h];
    acc = FM_WAVE_SUM(acc);
    FM_IF_LANE0 e_atom[i] = acc + (b2 ? b2[0] : T(0));
  }
}
// E_m = sum of e_atom over the atoms of molecule m (ascending idx_m with its row pointers).  err (may be NULL): the word the row-pointer
// kernels OR their findings into (bit 0: idx_i / idx_m not ascending) -- a batch the engine cannot run yields NaN energies instead of
// plausible numbers, without a host round trip
template <class T>
FM_KERNEL void k_fm_segsum1(const T* e_atom, const int32_t* rowptr_m, int64_t M, const int32_t* err, T* Eo) {
  FM_FOR_ROWS(m, M) {
    T acc = 0;
    const int a0 = rowptr_m[m], n = rowptr_m[m + 1] - a0;
    FM_FOR_LANES(k, n) acc += e_atom[a0 + k];
    acc = FM_WAVE_SUM(acc);
    if (err && (err[0] & 1)) acc = (T)NAN;
    FM_IF_LANE0 Eo[m] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ embedding
// x = table[Z], and the one-hot rows of Z beside it ([N, n_types]): the gradient of the table, gtable[z, :] = sum_{i: Z_i = z} gx[i, :], is then
// onehot^T gx -- one more problem of the batched weight-gradient launch instead of a kernel in which every (z, f) thread walks all atoms
// (14.6 us for 168 atoms).  A number outside [0, n_types) is an error, never an out-of-bounds read: its row is NaN (like the eval kernels:
// the energies of its molecule come out NaN), its one-hot row is zero and bit 1 of the validity flag is raised.
template <class T>
FM_KERNEL void k_fm_embed(const T* table, const int64_t* Z, int64_t N, int F, int n_types, T* x, T* onehot, int32_t* err) {
  FM_FOR(t, N * F) {
    const int64_t z = Z[t / F];
    const bool ok = (uint64_t)z < (uint64_t)n_types;
    x[t] = ok ? table[z * F + t % F] : (T)NAN;
    if (!ok && err && t % F == 0) FM_ATOMIC_OR(err, 2);
  }
  if (onehot) {
    FM_FOR(t, N * n_types) onehot[t] = Z[t / n_types] == (int64_t)(t % n_types) ? T(1) : T(0);
  }
}

// ================================================================================================ SchNet cfconv (schnet.py:64-67)
// y_i = sum_{e in row i} h_j Wf_e
template <class T>
FM_KERNEL void k_fm_cfconv(const T* FM_R h, const T* FM_R Wf, const int32_t* FM_R rowptr, const int64_t* FM_R jj, const int32_t* FM_R e_act, int64_t N, int nf,
                           T* FM_R y) {
  const int ea = *e_act;
  FM_FOR_SLOTTED(t, N * nf) {
    const int64_t i = t / nf;
    const int c = (int)(t % nf);
    int e1 = rowptr[i + 1];
    if (e1 > ea) e1 = ea;
    T acc[1] = {0};
    for (int eb = rowptr[i] + FM_SLOT * FM_CH4; eb < e1; eb += FM_NSLOT * FM_CH4) {
      int64_t j[FM_CH4];
      T a[FM_CH4], b[FM_CH4];
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) j[q] = FM_LDI(eb + q < e1, jj, eb + q, -1);
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) {
        const bool ok = (uint64_t)j[q] < (uint64_t)N;
        a[q] = FM_LD(ok, h, j[q] * nf + c);
        b[q] = FM_LD(ok, Wf, (int64_t)(eb + q) * nf + c);
      }
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) acc[0] += a[q] * b[q];
    }
    FM_SLOT_SUM(acc, 1);
    if (FM_SLOT_OWNER) y[t] = acc[0];
  }
}
// yt_i = sum_e (ht_j Wf_e + h_j Wf1_e dt_e)      (ht may be NULL: first interaction)
template <class T>
FM_KERNEL void k_fm_cfconv_t(const T* FM_R h, const T* FM_R ht, const T* FM_R Wf, const T* FM_R Wf1, const T* FM_R dt, const int32_t* FM_R rowptr,
                             const int64_t* FM_R jj, const int32_t* FM_R e_act, int64_t N, int nf, T* FM_R yt) {
  const int ea = *e_act;
  FM_FOR_SLOTTED(t, N * nf) {
    const int64_t i = t / nf;
    const int c = (int)(t % nf);
    int e1 = rowptr[i + 1];
    if (e1 > ea) e1 = ea;
    T acc[1] = {0};
    for (int eb = rowptr[i] + FM_SLOT * FM_CH4; eb < e1; eb += FM_NSLOT * FM_CH4) {
      int64_t j[FM_CH4];
      T a[FM_CH4], a1[FM_CH4], b[FM_CH4], b1[FM_CH4], de[FM_CH4];
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) j[q] = FM_LDI(eb + q < e1, jj, eb + q, -1);
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) {
        const bool ok = (uint64_t)j[q] < (uint64_t)N;
        const int64_t e = eb + q;
        a[q] = FM_LD(ok, h, j[q] * nf + c);
        a1[q] = FM_LDN(ok, ht, j[q] * nf + c, h);
        b[q] = FM_LD(ok && ht != nullptr, Wf, e * nf + c);
        b1[q] = FM_LD(ok, Wf1, e * nf + c);
        de[q] = FM_LD(ok, dt, e);
      }
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) {
        acc[0] += a[q] * b1[q] * de[q];
        if (ht) acc[0] += a1[q] * b[q];
      }
    }
    FM_SLOT_SUM(acc, 1);
    if (FM_SLOT_OWNER) yt[t] = acc[0];
  }
}
// pass B: gd_e += sum_c gy_i h_j Wf1_e   (one wavefront per pair)
template <class T>
FM_KERNEL void k_fm_cfconv_gd(const T* gy, const T* h, const T* Wf1, const int64_t* ii, const int64_t* jj, int64_t E, int64_t N, int nf, T* gd) {
  FM_FOR_ROWS(e, E) {
    const int64_t i = ii[e], j = jj[e];
    const bool ok = (uint64_t)i < (uint64_t)N && (uint64_t)j < (uint64_t)N;
    T acc = 0;
    if (ok) FM_FOR_LANES(c, nf) acc += gy[i * nf + c] * h[j * nf + c] * Wf1[e * nf + c];
    acc = FM_WAVE_SUM(acc);
    FM_IF_LANE0 gd[e] += acc;
  }
}
// pass B: gh_j = sum_{e: j(e) = j} gy_i Wf_e   (by-neighbour CSR: colptr, perm, csrc = source atom of every entry)
template <class T>
FM_KERNEL void k_fm_cfconv_T(const T* FM_R gy, const T* FM_R Wf, const int32_t* FM_R colptr, const int32_t* FM_R perm, const int32_t* FM_R csrc,
                             const int32_t* FM_R e_act, int64_t N, int nf, T* FM_R gh) {
  const int ea = *e_act;
  FM_FOR_SLOTTED(t, N * nf) {
    const int64_t j = t / nf;
    const int c = (int)(t % nf);
    T acc[1] = {0};
    const int k1 = colptr[j + 1];
    for (int kb = colptr[j] + FM_SLOT * FM_CH4; kb < k1; kb += FM_NSLOT * FM_CH4) {
      int e[FM_CH4], i[FM_CH4];
      T a[FM_CH4], b[FM_CH4];
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) {
        e[q] = FM_LDI(kb + q < k1, perm, kb + q, ea);
        i[q] = FM_LDI(kb + q < k1, csrc, kb + q, -1);
      }
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) {
        const bool ok = e[q] < ea && i[q] >= 0;
        a[q] = FM_LD(ok, gy, (int64_t)i[q] * nf + c);
        b[q] = FM_LD(ok, Wf, (int64_t)e[q] * nf + c);
      }
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) acc[0] += a[q] * b[q];
      if (e[FM_CH4 - 1] >= ea) break;
    }
    FM_SLOT_SUM(acc, 1);
    if (FM_SLOT_OWNER) gh[t] = acc[0];
  }
}
// pass D: gh_j = sum (gy_i Wf_e + hy_i Wf1_e dt_e),  hh_j = sum hy_i Wf_e ;  gy2 = [gy ; hy] [2N, nf], Wf2 = [Wf ; Wf1] [2E, nf]
template <class T>
FM_KERNEL void k_fm_cfconv_T_dual(const T* FM_R gy2, const T* FM_R Wf2, const T* FM_R dt, const int32_t* FM_R colptr, const int32_t* FM_R perm,
                                  const int32_t* FM_R csrc, const int32_t* FM_R e_act, int64_t N, int64_t E, int nf, T* FM_R gh2) {
  const int ea = *e_act;
  FM_FOR_SLOTTED(t, N * nf) {
    const int64_t j = t / nf;
    const int c = (int)(t % nf);
    T acc[2] = {0, 0};      // ag, ah
    const int k1 = colptr[j + 1];
    for (int kb = colptr[j] + FM_SLOT * FM_CH4; kb < k1; kb += FM_NSLOT * FM_CH4) {
      int e[FM_CH4], i[FM_CH4];
      T g[FM_CH4], hh[FM_CH4], w[FM_CH4], w1[FM_CH4], de[FM_CH4];
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) {
        e[q] = FM_LDI(kb + q < k1, perm, kb + q, ea);
        i[q] = FM_LDI(kb + q < k1, csrc, kb + q, -1);
      }
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) {
        const bool ok = e[q] < ea && i[q] >= 0;
        g[q] = FM_LD(ok, gy2, (int64_t)i[q] * nf + c);
        hh[q] = FM_LD(ok, gy2, (N + i[q]) * nf + c);
        w[q] = FM_LD(ok, Wf2, (int64_t)e[q] * nf + c);
        w1[q] = FM_LD(ok, Wf2, (E + e[q]) * nf + c);
        de[q] = FM_LD(ok, dt, e[q]);
      }
      FM_UNROLL for (int q = 0; q < FM_CH4; ++q) {
        acc[0] += g[q] * w[q] + hh[q] * w1[q] * de[q];
        acc[1] += hh[q] * w[q];
      }
      if (e[FM_CH4 - 1] >= ea) break;
    }
    FM_SLOT_SUM(acc, 2);
    if (FM_SLOT_OWNER) {
      gh2[t] = acc[0];
      gh2[N * nf + t] = acc[1];
    }
  }
}
// pass D, per pair and channel: cotangents of the raw filter outputs (g, g1):
//   gWf = gy_i h_j + hy_i ht_j,  Q = hy_i h_j dt  ->  gg = gWf f_c + Q f_c',  gg1 = Q f_c          (ht NULL: first interaction)
template <class T>
FM_KERNEL void k_fm_filter_cot(const T* gy2, const T* h, const T* ht, const T* dt, const T* fc, const T* fc1, const int64_t* ii, const int64_t* jj, int64_t N,
                               int64_t E, int nf, T* gg2) {
  FM_FOR(t, E * nf) {
    const int64_t e = t / nf;
    const int c = (int)(t % nf);
    const int64_t i = ii[e], j = jj[e];
    T o = 0, o1 = 0;
    if ((uint64_t)i < (uint64_t)N && (uint64_t)j < (uint64_t)N) {
      const T g = gy2[i * nf + c], hh = gy2[(N + i) * nf + c], hj = h[j * nf + c];
      const T gWf = g * hj + (ht ? hh * ht[j * nf + c] : T(0));
      const T Q = hh * hj * dt[e];
      o = gWf * fc[e] + Q * fc1[e];
      o1 = Q * fc[e];
    }
    gg2[t] = o;
    gg2[E * nf + t] = o1;
  }
}

// ================================================================================================ PaiNN message (painn.py:50-66)
// Phi: this interaction's filter slice, row stride ld (value rows; the d-derivative rows lie E * ld further)
template <class T>
FM_KERNEL void k_fm_painn_msg(const T* FM_R q, const T* FM_R mu, const T* FM_R c, const T* FM_R Phi, int ld, const T* FM_R u, const int32_t* FM_R rowptr,
                              const int64_t* FM_R jj, const int32_t* FM_R e_act, int64_t N, int F, T* FM_R q1, T* FM_R mu1) {
  const T* gathers_alt = c; (void)gathers_alt;
  const int ea = *e_act;
  FM_FOR_SLOTTED(t, N * F) {
    const int64_t i = t / F;
    const int f = (int)(t % F);
    T acc[4] = {0, 0, 0, 0};      // dq, dm[3]
    int e1 = rowptr[i + 1];
    if (e1 > ea) e1 = ea;
    for (int eb = rowptr[i] + FM_SLOT * FM_CH4; eb < e1; eb += FM_NSLOT * FM_CH4) {
      int64_t j[FM_CH4];
      T P[FM_CH4][3], cj[FM_CH4][3], uu[FM_CH4][3], mj[FM_CH4][3];
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) j[k] = FM_LDIC(eb + k < e1, jj, eb + k, -1);
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        const bool ok = (uint64_t)j[k] < (uint64_t)N;
        const int64_t e = eb + k;
        FM_UNROLL for (int p = 0; p < 3; ++p) {
          P[k][p] = FM_LDC(ok, Phi, e * ld + p * F + f);
          cj[k][p] = FM_LDC(ok, c, j[k] * 3 * F + p * F + f);
          uu[k][p] = FM_LDC(ok, u, e * 3 + p);
          mj[k][p] = FM_LDNC(ok, mu, (j[k] * 3 + p) * F + f);
        }
      }
      FM_FENCE();
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        acc[0] += P[k][0] * cj[k][0];
        const T mR = P[k][1] * cj[k][1], mm = P[k][2] * cj[k][2];
        FM_UNROLL for (int x = 0; x < 3; ++x) acc[1 + x] += mR * uu[k][x] + mm * mj[k][x];
      }
    }
    FM_SLOT_SUM(acc, 4);
    if (FM_SLOT_OWNER) {
      q1[t] = q[t] + acc[0];
      for (int x = 0; x < 3; ++x) mu1[(i * 3 + x) * F + f] = (mu ? mu[(i * 3 + x) * F + f] : T(0)) + acc[1 + x];
    }
  }
}
// tangent of the message; c2 = [c ; ct] [2N, 3F], mu2 = [mu ; mut] [2 * 3N, F] (NULL: first interaction, then ct = 0 too), qt NULL = 0
template <class T>
FM_KERNEL void k_fm_painn_msg_t(const T* FM_R qt, const T* FM_R c2, const T* FM_R mu2, const T* FM_R Phi, int ld, int64_t E, const T* FM_R dt, const T* FM_R u,
                                const T* FM_R ut, const int32_t* FM_R rowptr, const int64_t* FM_R jj, const int32_t* FM_R e_act, int64_t N, int F, int first,
                                T* FM_R q1t, T* FM_R mu1t) {
  const T* gathers_alt = c2; (void)gathers_alt;
  const int ea = *e_act;
  FM_FOR_SLOTTED(t, N * F) {
    const int64_t i = t / F;
    const int f = (int)(t % F);
    T acc[4] = {0, 0, 0, 0};      // dq, dm[3]
    int e1 = rowptr[i + 1];
    if (e1 > ea) e1 = ea;
    for (int eb = rowptr[i] + FM_SLOT * FM_CH4; eb < e1; eb += FM_NSLOT * FM_CH4) {
      int64_t j[FM_CH4];
      T P[FM_CH4][3], P1[FM_CH4][3], cj[FM_CH4][3], tj[FM_CH4][3], uu[FM_CH4][3], uv[FM_CH4][3], mj[FM_CH4][3], mtj[FM_CH4][3], de[FM_CH4];
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) j[k] = FM_LDIC(eb + k < e1, jj, eb + k, -1);
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        const bool ok = (uint64_t)j[k] < (uint64_t)N;
        const bool ok2 = ok && !first;
        const int64_t e = eb + k;
        de[k] = FM_LDC(ok, dt, e);
        FM_UNROLL for (int p = 0; p < 3; ++p) {
          P[k][p] = FM_LDC(ok, Phi, e * ld + p * F + f);
          P1[k][p] = FM_LDC(ok, Phi, (E + e) * ld + p * F + f);
          cj[k][p] = FM_LDC(ok, c2, j[k] * 3 * F + p * F + f);
          tj[k][p] = FM_LDC(ok2, c2, (N + j[k]) * 3 * F + p * F + f);
          uu[k][p] = FM_LDC(ok, u, e * 3 + p);
          uv[k][p] = FM_LDC(ok, ut, e * 3 + p);
          mj[k][p] = FM_LDNC(ok2, mu2, (j[k] * 3 + p) * F + f);
          mtj[k][p] = FM_LDNC(ok2, mu2, ((N + j[k]) * 3 + p) * F + f);
        }
      }
      FM_FENCE();
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        acc[0] += P1[k][0] * de[k] * cj[k][0] + P[k][0] * tj[k][0];
        const T mR = P[k][1] * cj[k][1], mRt = P1[k][1] * de[k] * cj[k][1] + P[k][1] * tj[k][1];
        const T mm = P[k][2] * cj[k][2], mmt = P1[k][2] * de[k] * cj[k][2] + P[k][2] * tj[k][2];
        FM_UNROLL for (int x = 0; x < 3; ++x) acc[1 + x] += mRt * uu[k][x] + mR * uv[k][x] + mmt * mj[k][x] + mm * mtj[k][x];
      }
    }
    FM_SLOT_SUM(acc, 4);
    if (FM_SLOT_OWNER) {
      q1t[t] = (first ? T(0) : qt[t]) + acc[0];
      for (int x = 0; x < 3; ++x) mu1t[(i * 3 + x) * F + f] = (first ? T(0) : mu2[((N + i) * 3 + x) * F + f]) + acc[1 + x];
    }
  }
}
// pass B, per pair (one wavefront): gd_e += sum gm c_j Phi1,  gu_e[x] += sum_f gmu1_i[x] mR     (gmu1 NULL = 0)
template <class T>
FM_KERNEL void k_fm_painn_msg_gd(const T* gq1, const T* gmu1, const T* c, const T* mu, const T* Phi, int ld, const T* u, const int64_t* ii, const int64_t* jj,
                                 int64_t E, int64_t N, int F, T* gd, T* gu) {
  FM_FOR_ROWS(e, E) {
    const int64_t i = ii[e], j = jj[e];
    const bool ok = (uint64_t)i < (uint64_t)N && (uint64_t)j < (uint64_t)N;
    T ad = 0, au[3] = {0, 0, 0};
    if (ok) {
      const T* P = Phi + e * (int64_t)ld;
      const T* P1 = Phi + (E + e) * (int64_t)ld;
      const T* cj = c + j * 3 * F;
      FM_FOR_LANES(f, F) {
        T g1[3] = {0, 0, 0}, gmR = 0, gmm = 0;
        if (gmu1)
          for (int x = 0; x < 3; ++x) {
            g1[x] = gmu1[(i * 3 + x) * F + f];
            gmR += g1[x] * u[e * 3 + x];
            if (mu) gmm += g1[x] * mu[(j * 3 + x) * F + f];
          }
        ad += gq1[i * F + f] * cj[f] * P1[f] + gmR * cj[F + f] * P1[F + f] + gmm * cj[2 * F + f] * P1[2 * F + f];
        const T mR = P[F + f] * cj[F + f];
        for (int x = 0; x < 3; ++x) au[x] += g1[x] * mR;
      }
    }
    ad = FM_WAVE_SUM(ad);
    for (int x = 0; x < 3; ++x) au[x] = FM_WAVE_SUM(au[x]);
    FM_IF_LANE0 {
      gd[e] += ad;
      for (int x = 0; x < 3; ++x) gu[e * 3 + x] += au[x];
    }
  }
}
// pass B, transposed sums of atom j:  gc_j = sum_{e -> j} Phi_e gm_e,  gmu_j = gmu1_j + sum_{e -> j} mm_e gmu1_i
template <class T>
FM_KERNEL void k_fm_painn_msg_T(const T* FM_R gq1, const T* FM_R gmu1, const T* FM_R c, const T* FM_R mu, const T* FM_R Phi, int ld, const T* FM_R u,
                                const int32_t* FM_R colptr, const int32_t* FM_R perm, const int32_t* FM_R csrc, const int32_t* FM_R e_act, int64_t N, int F,
                                T* FM_R gc, T* FM_R gmu) {
  const T* gathers_alt = gq1; (void)gathers_alt;
  const int ea = *e_act;
  FM_FOR_SLOTTED(t, N * F) {
    const int64_t j = t / F;
    const int f = (int)(t % F);
    T acc[6] = {0, 0, 0, 0, 0, 0}, mj[3];      // aq, aR, am, ag[3]
    const T cm = c[j * 3 * F + 2 * F + f];
    for (int x = 0; x < 3; ++x) mj[x] = mu ? mu[(j * 3 + x) * F + f] : T(0);
    const int k1 = colptr[j + 1];
    for (int kb = colptr[j] + FM_SLOT * FM_CH4; kb < k1; kb += FM_NSLOT * FM_CH4) {
      int e[FM_CH4], i[FM_CH4];
      T P[FM_CH4][3], g1[FM_CH4][3], uu[FM_CH4][3], gq[FM_CH4];
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        e[k] = FM_LDIC(kb + k < k1, perm, kb + k, ea);
        i[k] = FM_LDIC(kb + k < k1, csrc, kb + k, -1);
      }
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        const bool ok = e[k] < ea && i[k] >= 0;
        gq[k] = FM_LDC(ok, gq1, (int64_t)i[k] * F + f);
        FM_UNROLL for (int p = 0; p < 3; ++p) {
          P[k][p] = FM_LDC(ok, Phi, (int64_t)e[k] * ld + p * F + f);
          g1[k][p] = FM_LDNC(ok, gmu1, ((int64_t)i[k] * 3 + p) * F + f);
          uu[k][p] = FM_LDC(ok, u, (int64_t)e[k] * 3 + p);
        }
      }
      FM_FENCE();
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        T gmR = 0, gmm = 0;
        FM_UNROLL for (int x = 0; x < 3; ++x) {
          gmR += g1[k][x] * uu[k][x];
          gmm += g1[k][x] * mj[x];
        }
        acc[0] += P[k][0] * gq[k];
        acc[1] += P[k][1] * gmR;
        acc[2] += P[k][2] * gmm;
        const T mm = P[k][2] * cm;
        FM_UNROLL for (int x = 0; x < 3; ++x) acc[3 + x] += mm * g1[k][x];
      }
      if (e[FM_CH4 - 1] >= ea) break;
    }
    FM_SLOT_SUM(acc, 6);
    if (FM_SLOT_OWNER) {
      gc[j * 3 * F + f] = acc[0];
      gc[j * 3 * F + F + f] = acc[1];
      gc[j * 3 * F + 2 * F + f] = acc[2];
      for (int x = 0; x < 3; ++x) gmu[(j * 3 + x) * F + f] = (gmu1 ? gmu1[(j * 3 + x) * F + f] : T(0)) + acc[3 + x];
    }
  }
}
// per-pair cotangents of the message, shared by the two pass-D kernels below
template <class T>
struct FmMsgCot { T gm[3], hm[3]; };
template <class T>
FM_HD FmMsgCot<T> fm_msg_cot(const T* gq1_2, const T* gmu1_2, const T* mu2, const T* u, const T* ut, int64_t e, int64_t i, int64_t j, int64_t N, int F, int f,
                             int first, T* g1, T* h1) {
  FmMsgCot<T> r;
  r.gm[0] = gq1_2[i * F + f];
  r.hm[0] = gq1_2[(N + i) * F + f];
  r.gm[1] = r.gm[2] = r.hm[1] = r.hm[2] = 0;
  for (int x = 0; x < 3; ++x) {
    g1[x] = gmu1_2[(i * 3 + x) * F + f];
    h1[x] = gmu1_2[((N + i) * 3 + x) * F + f];
    r.gm[1] += g1[x] * u[e * 3 + x] + h1[x] * ut[e * 3 + x];
    r.hm[1] += h1[x] * u[e * 3 + x];
    if (!first) {
      const T m = mu2[(j * 3 + x) * F + f], mt = mu2[((N + j) * 3 + x) * F + f];
      r.gm[2] += g1[x] * m + h1[x] * mt;
      r.hm[2] += h1[x] * m;
    }
  }
  return r;
}
// pass D, transposed dual sums of atom j -> gc2 = [gc ; hc] [2N, 3F], gmu2 = [gmu ; hmu] (in: gmu1_2, out may not alias)
template <class T>
FM_KERNEL void k_fm_painn_msg_T_dual(const T* FM_R gq1_2, const T* FM_R gmu1_2, const T* FM_R c2, const T* FM_R mu2, const T* FM_R Phi, int ld, int64_t E,
                                     const T* FM_R dt, const T* FM_R u, const T* FM_R ut, const int32_t* FM_R colptr, const int32_t* FM_R perm,
                                     const int32_t* FM_R csrc, const int32_t* FM_R e_act, int64_t N, int F, int first, T* FM_R gc2, T* FM_R gmu2) {
  const int ea = *e_act;
  FM_FOR_SLOTTED(t, N * F) {
    const int64_t j = t / F;
    const int f = (int)(t % F);
    T acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mj[3], mtj[3];      // ag[3], ah[3], mg[3], mh[3]
    const T cm = c2[j * 3 * F + 2 * F + f], ctm = first ? T(0) : c2[(N + j) * 3 * F + 2 * F + f];
    for (int x = 0; x < 3; ++x) {
      mj[x] = first ? T(0) : mu2[(j * 3 + x) * F + f];
      mtj[x] = first ? T(0) : mu2[((N + j) * 3 + x) * F + f];
    }
    const int k1 = colptr[j + 1];
    for (int kb = colptr[j] + FM_SLOT * FM_CH4; kb < k1; kb += FM_NSLOT * FM_CH4) {
      int e[FM_CH4], i[FM_CH4];
      T P[FM_CH4][3], P1[FM_CH4][3], g1[FM_CH4][3], h1[FM_CH4][3], uu[FM_CH4][3], uv[FM_CH4][3], gq[FM_CH4], hq[FM_CH4], de[FM_CH4];
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        e[k] = FM_LDIC(kb + k < k1, perm, kb + k, ea);
        i[k] = FM_LDIC(kb + k < k1, csrc, kb + k, -1);
      }
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        const bool ok = e[k] < ea && i[k] >= 0;
        const int64_t ee = e[k], iv = i[k];
        gq[k] = FM_LDC(ok, gq1_2, iv * F + f);
        hq[k] = FM_LDC(ok, gq1_2, (N + iv) * F + f);
        de[k] = FM_LDC(ok, dt, ee);
        FM_UNROLL for (int p = 0; p < 3; ++p) {
          P[k][p] = FM_LDC(ok, Phi, ee * ld + p * F + f);
          P1[k][p] = FM_LDC(ok, Phi, (E + ee) * ld + p * F + f);
          g1[k][p] = FM_LDC(ok, gmu1_2, (iv * 3 + p) * F + f);
          h1[k][p] = FM_LDC(ok, gmu1_2, ((N + iv) * 3 + p) * F + f);
          uu[k][p] = FM_LDC(ok, u, ee * 3 + p);
          uv[k][p] = FM_LDC(ok, ut, ee * 3 + p);
        }
      }
      FM_FENCE();
      FM_UNROLL for (int k = 0; k < FM_CH4; ++k) {
        T gm[3], hm[3];
        gm[0] = gq[k]; hm[0] = hq[k];
        gm[1] = gm[2] = hm[1] = hm[2] = 0;
        FM_UNROLL for (int x = 0; x < 3; ++x) {
          gm[1] += g1[k][x] * uu[k][x] + h1[k][x] * uv[k][x];
          hm[1] += h1[k][x] * uu[k][x];
          gm[2] += g1[k][x] * mj[x] + h1[k][x] * mtj[x];
          hm[2] += h1[k][x] * mj[x];
        }
        FM_UNROLL for (int p = 0; p < 3; ++p) {
          acc[p] += gm[p] * P[k][p] + hm[p] * P1[k][p] * de[k];
          acc[3 + p] += hm[p] * P[k][p];
        }
        const T mm = P[k][2] * cm, mmt = P1[k][2] * de[k] * cm + P[k][2] * ctm;
        FM_UNROLL for (int x = 0; x < 3; ++x) {
          acc[6 + x] += g1[k][x] * mm + h1[k][x] * mmt;
          acc[9 + x] += h1[k][x] * mm;
        }
      }
      if (e[FM_CH4 - 1] >= ea) break;
    }
    FM_SLOT_SUM(acc, 12);
    if (FM_SLOT_OWNER) {
      for (int p = 0; p < 3; ++p) {
        gc2[j * 3 * F + p * F + f] = acc[p];
        gc2[(N + j) * 3 * F + p * F + f] = acc[3 + p];
      }
      for (int x = 0; x < 3; ++x) {
        gmu2[(j * 3 + x) * F + f] = gmu1_2[(j * 3 + x) * F + f] + acc[6 + x];
        gmu2[((N + j) * 3 + x) * F + f] = gmu1_2[((N + j) * 3 + x) * F + f] + acc[9 + x];
      }
    }
  }
}
// pass D, per pair and channel: cotangents of the raw filter rows  gP2 = [gPraw ; gPraw1] [2E, 3F]
template <class T>
FM_KERNEL void k_fm_painn_filter_cot(const T* gq1_2, const T* gmu1_2, const T* c2, const T* mu2, const T* dt, const T* u, const T* ut, const T* fc, const T* fc1,
                                     const int64_t* ii, const int64_t* jj, int64_t N, int64_t E, int F, int first, T* gP2) {
  FM_FOR(t, E * F) {
    const int64_t e = t / F;
    const int f = (int)(t % F);
    const int64_t i = ii[e], j = jj[e];
    const bool ok = (uint64_t)i < (uint64_t)N && (uint64_t)j < (uint64_t)N;
    T g1[3], h1[3];
    FmMsgCot<T> m;
    if (ok) m = fm_msg_cot(gq1_2, gmu1_2, mu2, u, ut, e, i, j, N, F, f, first, g1, h1);
    for (int p = 0; p < 3; ++p) {
      T o = 0, o1 = 0;
      if (ok) {
        const T cj = c2[j * 3 * F + p * F + f], ctj = first ? T(0) : c2[(N + j) * 3 * F + p * F + f];
        const T gPhi = m.gm[p] * cj + m.hm[p] * ctj;
        const T gPhi1 = m.hm[p] * cj * dt[e];
        o = gPhi * fc[e] + gPhi1 * fc1[e];
        o1 = gPhi1 * fc[e];
      }
      gP2[e * 3 * F + p * F + f] = o;
      gP2[(E + e) * 3 * F + p * F + f] = o1;
    }
  }
}

// ================================================================================================ PaiNN mixing (painn.py:99-116)
// The eight element-wise kernels of the mixing block are ATOM-LOCAL: item t = (atom i, channel f) reads and writes rows of atom i only.  Each is
// written as a per-item function `fm_painn_*_at(t, ...)`; the kernels below run it over all N F items, the row-chain kernel of spk_fm_chain.h runs
// it over the items of a workgroup's own atoms between two Dense stages (one launch for what is a chain of launches otherwise).
// n = sqrt(sum_x V^2 + eps), svw = sum_x V W, ctx = [q1 | n]
template <class T>
FM_HD void fm_painn_mix_at(int64_t t, const T* q1, const T* VW, T eps, int64_t N, int F, T* n, T* svw, T* ctx) {
  const int64_t i = t / F;
  const int f = (int)(t % F);
  T s2 = 0, sv = 0;
  for (int x = 0; x < 3; ++x) {
    const T V = VW[(i * 3 + x) * 2 * F + f], W = VW[(i * 3 + x) * 2 * F + F + f];
    s2 += V * V;
    sv += V * W;
  }
  const T nn = fm_sqrt(s2 + eps);
  n[t] = nn;
  svw[t] = sv;
  ctx[i * 2 * F + f] = q1[t];
  ctx[i * 2 * F + F + f] = nn;
}
template <class T>
FM_HD void fm_painn_mix_t_at(int64_t t, const T* q1t, const T* VW, const T* VWt, const T* n, int64_t N, int F, T* nt, T* svwt, T* ctxt) {
  const int64_t i = t / F;
  const int f = (int)(t % F);
  T a = 0, sv = 0;
  for (int x = 0; x < 3; ++x) {
    const int64_t o = (i * 3 + x) * 2 * F + f;
    a += VW[o] * VWt[o];
    sv += VWt[o] * VW[o + F] + VW[o] * VWt[o + F];
  }
  const T v = a / n[t];
  nt[t] = v;
  svwt[t] = sv;
  ctxt[i * 2 * F + f] = q1t[t];
  ctxt[i * 2 * F + F + f] = v;
}
// q2 = q1 + a_q + a_qmu svw,  mu2 = mu1 + a_mu W
// (every function below requests ALL its operands before its first store: inside a row chain a load issued behind a store waits until that
//  store has been acknowledged -- loads and stores share one in-order counter -- and a workgroup alone on its compute unit has nobody to hide
//  ~2 us of that behind)
template <class T>
FM_HD void fm_painn_update_at(int64_t t, const T* q1, const T* mu1, const T* VW, const T* a, const T* svw, int64_t N, int F, T* q2, T* mu2) {
  const int64_t i = t / F;
  const int f = (int)(t % F);
  const T am = a[i * 3 * F + F + f];
  const T q = q1[t] + a[i * 3 * F + f] + a[i * 3 * F + 2 * F + f] * svw[t];
  T m[3];
  for (int x = 0; x < 3; ++x) m[x] = mu1[(i * 3 + x) * F + f] + am * VW[(i * 3 + x) * 2 * F + F + f];
  q2[t] = q;
  for (int x = 0; x < 3; ++x) mu2[(i * 3 + x) * F + f] = m[x];
}
template <class T>
FM_HD void fm_painn_update_t_at(int64_t t, const T* q1t, const T* mu1t, const T* VW, const T* VWt, const T* a, const T* at, const T* svw, const T* svwt, int64_t N,
                                int F, T* q2t, T* mu2t) {
  const int64_t i = t / F;
  const int f = (int)(t % F);
  const T q = q1t[t] + at[i * 3 * F + f] + at[i * 3 * F + 2 * F + f] * svw[t] + a[i * 3 * F + 2 * F + f] * svwt[t];
  const T am = a[i * 3 * F + F + f], amt = at[i * 3 * F + F + f];
  T m[3];
  for (int x = 0; x < 3; ++x) {
    const int64_t o = (i * 3 + x) * 2 * F + F + f;
    m[x] = mu1t[(i * 3 + x) * F + f] + amt * VW[o] + am * VWt[o];
  }
  q2t[t] = q;
  for (int x = 0; x < 3; ++x) mu2t[(i * 3 + x) * F + f] = m[x];
}
// pass B: ga = (gq | sum gmu W | gq svw);  gs = gq a_qmu;  gV = gs W;  gW = gmu a_mu + gs V      (gmu NULL = 0)
template <class T>
FM_HD void fm_painn_update_bwd_at(int64_t t, const T* gq, const T* gmu, const T* VW, const T* a, const T* svw, int64_t N, int F, T* ga, T* gVW) {
  const int64_t i = t / F;
  const int f = (int)(t % F);
  const T g = gq[t], am = a[i * 3 * F + F + f], gs = g * a[i * 3 * F + 2 * F + f], sv = svw[t];
  T s = 0, gV[3], gW[3];
  for (int x = 0; x < 3; ++x) {
    const int64_t o = (i * 3 + x) * 2 * F + f;
    const T gm = gmu ? gmu[(i * 3 + x) * F + f] : T(0);
    const T V = VW[o], W = VW[o + F];
    s += gm * W;
    gV[x] = gs * W;
    gW[x] = gm * am + gs * V;
  }
  for (int x = 0; x < 3; ++x) {
    const int64_t o = (i * 3 + x) * 2 * F + f;
    gVW[o] = gV[x];
    gVW[o + F] = gW[x];
  }
  ga[i * 3 * F + f] = g;
  ga[i * 3 * F + F + f] = s;
  ga[i * 3 * F + 2 * F + f] = g * sv;
}
// pass B: gq1 = gq + gctx[:, :F];  gV += (gctx[:, F:] / n) V
template <class T>
FM_HD void fm_painn_mix_bwd_at(int64_t t, const T* gq, const T* gctx, const T* VW, const T* n, int64_t N, int F, T* gq1, T* gVW) {
  const int64_t i = t / F;
  const int f = (int)(t % F);
  const T q = gq[t] + gctx[i * 2 * F + f];
  const T s = gctx[i * 2 * F + F + f] / n[t];
  T v[3];
  for (int x = 0; x < 3; ++x) {
    const int64_t o = (i * 3 + x) * 2 * F + f;
    v[x] = gVW[o] + s * VW[o];
  }
  gq1[t] = q;
  for (int x = 0; x < 3; ++x) gVW[(i * 3 + x) * 2 * F + f] = v[x];
}
// pass D: stacked cotangents in (gq2 = [gq ; hq], gmu2 = [gmu ; hmu] or NULL), values VW2 = [VW ; VWt], a2 = [a ; at], svw2 = [svw ; svwt]
//   -> ga2 = [ga ; ha] [2N, 3F], gVW2 = [gVW ; hVW] [2 * 3N, 2F]       (equations (1)-(6) of oracle/fm_oracle.py)
template <class T>
FM_HD void fm_painn_update_dual_bwd_at(int64_t t, const T* gq2, const T* gmu2, const T* VW2, const T* a2, const T* svw2, int64_t N, int F, T* ga2, T* gVW2) {
  const int64_t i = t / F;
  const int f = (int)(t % F);
  const T gq = gq2[t], hq = gq2[N * F + t];
  const T am = a2[i * 3 * F + F + f], aqm = a2[i * 3 * F + 2 * F + f];
  const T amt = a2[(N + i) * 3 * F + F + f], aqmt = a2[(N + i) * 3 * F + 2 * F + f];
  const T sv = svw2[t], svt = svw2[N * F + t];
  const T gs = gq * aqm + hq * aqmt, hs = hq * aqm;
  T sg = 0, sh = 0, oV[3], oW[3], oVt[3], oWt[3];
  for (int x = 0; x < 3; ++x) {
    const int64_t o = (i * 3 + x) * 2 * F + f, ot = ((N + i) * 3 + x) * 2 * F + f;
    const T V = VW2[o], W = VW2[o + F], Vt = VW2[ot], Wt = VW2[ot + F];
    const T gm = gmu2 ? gmu2[(i * 3 + x) * F + f] : T(0), hm = gmu2 ? gmu2[((N + i) * 3 + x) * F + f] : T(0);
    sg += gm * W + hm * Wt;
    sh += hm * W;
    oV[x] = gs * W + hs * Wt;                       // gV
    oW[x] = gm * am + hm * amt + gs * V + hs * Vt;  // gW
    oVt[x] = hs * W;                                // hV
    oWt[x] = hm * am + hs * V;                      // hW
  }
  for (int x = 0; x < 3; ++x) {
    const int64_t o = (i * 3 + x) * 2 * F + f, ot = ((N + i) * 3 + x) * 2 * F + f;
    gVW2[o] = oV[x];
    gVW2[o + F] = oW[x];
    gVW2[ot] = oVt[x];
    gVW2[ot + F] = oWt[x];
  }
  ga2[i * 3 * F + f] = gq;
  ga2[i * 3 * F + F + f] = sg;
  ga2[i * 3 * F + 2 * F + f] = gq * sv + hq * svt;
  ga2[(N + i) * 3 * F + f] = hq;
  ga2[(N + i) * 3 * F + F + f] = sh;
  ga2[(N + i) * 3 * F + 2 * F + f] = hq * sv;
}
// pass D: gq1 = gq + gctx[:, :F], hq1 = hq + hctx[:, :F];  gV += gn/n V + hn/n (Vt - nt/n V),  hV += hn/n V      (equations (7), (8))
template <class T>
FM_HD void fm_painn_mix_dual_bwd_at(int64_t t, const T* gq2, const T* gctx2, const T* VW2, const T* n2, int64_t N, int F, T* gq1_2, T* gVW2) {
  const int64_t i = t / F;
  const int f = (int)(t % F);
  const T q = gq2[t] + gctx2[i * 2 * F + f];
  const T qt = gq2[N * F + t] + gctx2[(N + i) * 2 * F + f];
  const T n = n2[t], nt = n2[N * F + t];
  const T gn = gctx2[i * 2 * F + F + f] / n, hn = gctx2[(N + i) * 2 * F + F + f] / n;
  T v[3], vt[3];
  for (int x = 0; x < 3; ++x) {
    const int64_t o = (i * 3 + x) * 2 * F + f, ot = ((N + i) * 3 + x) * 2 * F + f;
    const T V = VW2[o], Vt = VW2[ot];
    v[x] = gVW2[o] + gn * V + hn * (Vt - nt / n * V);
    vt[x] = gVW2[ot] + hn * V;
  }
  gq1_2[t] = q;
  gq1_2[N * F + t] = qt;
  for (int x = 0; x < 3; ++x) {
    gVW2[(i * 3 + x) * 2 * F + f] = v[x];
    gVW2[((N + i) * 3 + x) * 2 * F + f] = vt[x];
  }
}

// One descriptor for the eight: kind + up to eight inputs and two outputs (the order of each function's arguments above), so that the engine's
// backends can record them as stages of a row chain (spk_fm_chain.h) or run them as one generic launch.
enum { FM_EW_MIX = 0, FM_EW_MIX_T, FM_EW_UPDATE, FM_EW_UPDATE_T, FM_EW_UPDATE_BWD, FM_EW_MIX_BWD, FM_EW_UPDATE_DUAL_BWD, FM_EW_MIX_DUAL_BWD, FM_EW_KINDS };
template <class T>
struct FmEwArgs {
  int kind, F;
  int64_t N;
  T eps;
  const T* in[8];
  T* out[3];
};
template <class T>
FM_HD void fm_ew_at(const FmEwArgs<T>& a, int64_t t) {
  const T* const* i = a.in;
  T* const* o = a.out;
  switch (a.kind) {
    case FM_EW_MIX: fm_painn_mix_at<T>(t, i[0], i[1], a.eps, a.N, a.F, o[0], o[1], o[2]); break;
    case FM_EW_MIX_T: fm_painn_mix_t_at<T>(t, i[0], i[1], i[2], i[3], a.N, a.F, o[0], o[1], o[2]); break;
    case FM_EW_UPDATE: fm_painn_update_at<T>(t, i[0], i[1], i[2], i[3], i[4], a.N, a.F, o[0], o[1]); break;
    case FM_EW_UPDATE_T: fm_painn_update_t_at<T>(t, i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], a.N, a.F, o[0], o[1]); break;
    case FM_EW_UPDATE_BWD: fm_painn_update_bwd_at<T>(t, i[0], i[1], i[2], i[3], i[4], a.N, a.F, o[0], o[1]); break;
    case FM_EW_MIX_BWD: fm_painn_mix_bwd_at<T>(t, i[0], i[1], i[2], i[3], a.N, a.F, o[0], o[1]); break;
    case FM_EW_UPDATE_DUAL_BWD: fm_painn_update_dual_bwd_at<T>(t, i[0], i[1], i[2], i[3], i[4], a.N, a.F, o[0], o[1]); break;
    default: fm_painn_mix_dual_bwd_at<T>(t, i[0], i[1], i[2], i[3], a.N, a.F, o[0], o[1]); break;
  }
}
template <class T>
FM_KERNEL void k_fm_ew(FmEwArgs<T> a) { FM_FOR(t, a.N * a.F) fm_ew_at<T>(a, t); }
