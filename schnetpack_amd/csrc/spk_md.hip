// MD step kernels: the elementwise half/main steps of the reference's
// integrators fused, and the ring-polymer main step as one bead-mixing kernel.  HBM-bound; every
// array is read and written once per step.
#include "spk_common.h"
#include <atomic>

// p += 1/2 dt F                                   (md/integrators.py:59-70, Integrator.half_step)
__global__ void k_md_half_step(float* __restrict__ p, const float* __restrict__ F, float half_dt, int64_t n) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
    p[t] = fmaf(half_dt, F[t], p[t]);
}

// first half step + main step of velocity Verlet in one pass (md/integrators.py:59-70, :97-110):
//   p += 1/2 dt F ;  R += dt p / m
// and, for the neighbour-list skin (md/neighborlist_md.py:80-90), flag[0] |= any |R - R_ref|^2 > max_disp2;
// flag[1] = running maximum of the squared displacement of ONE step (bits of a non-negative float, atomicMax),
// which lets the caller keep a safety margin when it reads the flag one step late.
__global__ void k_md_kick_drift(float* __restrict__ R, float* __restrict__ p, const float* __restrict__ F,
                                const float* __restrict__ masses, float dt, int64_t n_atoms,
                                const float* __restrict__ R_ref, float max_disp2, int32_t* __restrict__ flag) {
  bool moved = false;
  float step2 = 0.f;
  for (int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; a < n_atoms; a += (int64_t)gridDim.x * blockDim.x) {
    const float dtm = dt / masses[a];
    float d2 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int64_t t = 3 * a + c;
      const float pn = F ? fmaf(0.5f * dt, F[t], p[t]) : p[t];
      const float dr = dtm * pn;
      const float rn = R[t] + dr;
      p[t] = pn;
      R[t] = rn;
      s2 = fmaf(dr, dr, s2);
      if (R_ref) { const float d = rn - R_ref[t]; d2 = fmaf(d, d, d2); }
    }
    step2 = fmaxf(step2, s2);
    moved |= (R_ref != nullptr) && (d2 > max_disp2);
  }
  if (flag) {
    if (__any(moved)) { if ((threadIdx.x & 63) == 0) atomicOr((int*)flag, 1); }
    step2 = spk_wave_max(step2);
    if ((threadIdx.x & 63) == 0) atomicMax((int*)flag + 1, __float_as_int(step2));
  }
}

// Ring-polymer main step (md/integrators.py:204-229 with the normal-mode matrix of
// md/utils/normal_model_transformation.py:38-98).  Transform, propagate and back-transform are linear in
// (p, q), so they fold into four n_beads x n_beads matrices prepared once on the host:
//   p'_b = sum_n ( App[b][n] p_n + m Apq[b][n] q_n ),   q'_b = sum_n ( Aqp[b][n] p_n / m + Aqq[b][n] q_n )
// with A.. = C^T diag(propagator[:, i, j]) C.  One thread per (atom, component) reads all beads once
// and writes the beads [bead0, bead0 + n_local) owned by this rank.
__global__ __launch_bounds__(256) void k_md_ring_polymer(const float* __restrict__ q_all, const float* __restrict__ p_all,
                                                         const float* __restrict__ masses, const float* __restrict__ A,
                                                         int B, int64_t n_atoms, int bead0, int n_local,
                                                         float* __restrict__ q_out, float* __restrict__ p_out,
                                                         const float* __restrict__ R_ref, float max_disp2,
                                                         int32_t* __restrict__ flag) {
  extern __shared__ float sA[];   // [4][B][B]
  for (int t = threadIdx.x; t < 4 * B * B; t += blockDim.x) sA[t] = A[t];
  __syncthreads();
  const int64_t n3 = 3 * n_atoms;
  bool moved = false;
  float step2 = 0.f;
  // one thread per atom: the skin test needs the three components of its displacement
  for (int64_t at = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; at < n_atoms; at += (int64_t)gridDim.x * blockDim.x) {
    const float m = masses[at];
    const float im = 1.0f / m;
    for (int bl = 0; bl < n_local; ++bl) {
      const int b = bead0 + bl;
      float d2 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int64_t t = 3 * at + c;
        float pn = 0.f, qn = 0.f;
        for (int n = 0; n < B; ++n) {
          const float pv = p_all[(int64_t)n * n3 + t], qv = q_all[(int64_t)n * n3 + t];
          pn = fmaf(sA[(0 * B + b) * B + n], pv, pn);
          pn = fmaf(sA[(1 * B + b) * B + n] * m, qv, pn);
          qn = fmaf(sA[(2 * B + b) * B + n] * im, pv, qn);
          qn = fmaf(sA[(3 * B + b) * B + n], qv, qn);
        }
        p_out[(int64_t)bl * n3 + t] = pn;
        q_out[(int64_t)bl * n3 + t] = qn;
        const float ds = qn - q_all[(int64_t)b * n3 + t];
        s2 = fmaf(ds, ds, s2);
        if (R_ref) { const float d = qn - R_ref[(int64_t)bl * n3 + t]; d2 = fmaf(d, d, d2); }
      }
      step2 = fmaxf(step2, s2);
      moved |= (R_ref != nullptr) && (d2 > max_disp2);
    }
  }
  if (flag) {
    if (__any(moved)) { if ((threadIdx.x & 63) == 0) atomicOr((int*)flag, 1); }
    step2 = spk_wave_max(step2);
    if ((threadIdx.x & 63) == 0) atomicMax((int*)flag + 1, __float_as_int(step2));
  }
}

extern "C" int spk_md_half_step_f32(float* p, const float* F, float half_dt, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n >= 0, "spk_md_half_step_f32: bad size");
  if (n == 0) return SPK_OK;
  SPK_CHECK_ARG(p && F, "spk_md_half_step_f32: null pointer");
  hipLaunchKernelGGL(k_md_half_step, dim3(spk_grid_for(n, 256, spk_num_cus() * 8)), dim3(256), 0, stream, p, F, half_dt, n);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_md_kick_drift_f32(float* R, float* p, const float* F, const float* masses, float dt,
                                     int64_t n_atoms, const float* R_ref, float max_disp2, int32_t* flag,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n_atoms >= 0, "spk_md_kick_drift_f32: bad size");
  if (n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(R && p && masses, "spk_md_kick_drift_f32: null pointer");
  SPK_CHECK_ARG((R_ref == nullptr) == (flag == nullptr), "spk_md_kick_drift_f32: R_ref and flag go together");
  hipLaunchKernelGGL(k_md_kick_drift, dim3(spk_grid_for(n_atoms, 256, spk_num_cus() * 8)), dim3(256), 0, stream,
                     R, p, F, masses, dt, n_atoms, R_ref, max_disp2, flag);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_md_ring_polymer_step_f32(const float* q_all, const float* p_all, const float* masses,
                                            const float* A, int32_t n_beads, int64_t n_atoms, int32_t bead0,
                                            int32_t n_local, float* q_out, float* p_out, const float* R_ref,
                                            float max_disp2, int32_t* flag, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n_beads >= 1 && n_beads <= 96 && n_atoms >= 0, "spk_md_ring_polymer_step_f32: bad sizes (n_beads <= 96)");
  if (n_beads > 64) {   // 16 B^2 bytes of dynamic LDS: above 64 KB the launch needs the attribute -- a property of the function ON ONE
                        // DEVICE, so it is tracked per device (once each, not a stream operation)
    static SpkPerDevice attr_set;
    int attr_dev;
    if (attr_set.pending(&attr_dev)) {
      SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_md_ring_polymer, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 96 * 96));
      attr_set.mark(attr_dev);
    }
  }
  SPK_CHECK_ARG(bead0 >= 0 && n_local >= 0 && bead0 + n_local <= n_beads, "spk_md_ring_polymer_step_f32: bead range outside [0, n_beads)");
  if (n_atoms == 0 || n_local == 0) return SPK_OK;
  SPK_CHECK_ARG(q_all && p_all && masses && A && q_out && p_out, "spk_md_ring_polymer_step_f32: null pointer");
  SPK_CHECK_ARG(q_out != q_all && p_out != p_all, "spk_md_ring_polymer_step_f32: outputs must not alias the inputs");
  const size_t lds = sizeof(float) * 4 * (size_t)n_beads * n_beads;
  SPK_CHECK_ARG((R_ref == nullptr) || (flag != nullptr), "spk_md_ring_polymer_step_f32: R_ref needs a flag buffer");
  hipLaunchKernelGGL(k_md_ring_polymer, dim3(spk_grid_for(n_atoms, 256, spk_num_cus() * 8)), dim3(256), lds, stream,
                     q_all, p_all, masses, A, n_beads, n_atoms, bead0, n_local, q_out, p_out, R_ref, max_disp2, flag);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------
// PILE-L thermostat (md/simulation_hooks/thermostats_rpmd.py:33-119): Langevin thermostat on the NORMAL-MODE momenta of a
// ring polymer,  p_nm' = c1_k p_nm + sqrt(m kB n_beads T) c2_k xi,  xi ~ N(0, 1).  Transform, scale and back-transform are
// linear, so in bead space
//     p'_b = sum_b' M1[b][b'] p_b'  +  sqrt(m) s  sum_k M2[b][k] xi_k ,      M1 = C^T diag(c1) C,   M2 = C^T diag(c2)
// One thread per (atom, component).  The noise of mode k comes from a counter-based generator (Philox-4x32-10 keyed by the
// seed, counter = (atom component, mode pair, step, application)): every rank of a bead-parallel run regenerates the SAME
// xi_k for all modes from the counter alone, so the only exchange of a thermostat application is the all-gather of the
// momenta ("identical RNG streams"); the result does not depend on how beads are spread over ranks.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void spk_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// two standard normals from two 32-bit words (Box-Muller; u in (0, 1])
__device__ __forceinline__ void spk_box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float v = (float)(b >> 8) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * logf(u));
  float s, c;
  sincosf(6.283185307179586f * v, &s, &c);
  n0 = r * c; n1 = r * s;
}

#define PILE_CHUNK 8   // local beads per pass (accumulators per thread); more local beads = more passes, noise regenerated per pass
__global__ void k_md_pile(const float* __restrict__ p_all, const float* __restrict__ masses, const float* __restrict__ M, float scale,
                          uint32_t seed_lo, uint32_t seed_hi, uint64_t step_host, const int64_t* __restrict__ step_dev, uint32_t which,
                          int B, int64_t n_atoms, int bead0, int n_local, float* __restrict__ p_out) {
  extern __shared__ float sM[];   // [2][B][B]
  for (int s = threadIdx.x; s < 2 * B * B; s += blockDim.x) sM[s] = M[s];
  __syncthreads();
  const uint64_t step = step_dev ? (uint64_t)step_dev[0] : step_host;
  const int64_t n3 = 3 * n_atoms;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n3; t += (int64_t)gridDim.x * blockDim.x) {
    const float sm = sqrtf(masses[t / 3]) * scale;
    for (int b0 = 0; b0 < n_local; b0 += PILE_CHUNK) {
      float det[PILE_CHUNK], noi[PILE_CHUNK];
#pragma unroll
      for (int u = 0; u < PILE_CHUNK; ++u) { det[u] = 0.f; noi[u] = 0.f; }
      for (int n = 0; n < B; ++n) {
        const float pv = p_all[(int64_t)n * n3 + t];
#pragma unroll
        for (int u = 0; u < PILE_CHUNK; ++u)
          if (b0 + u < n_local) det[u] = fmaf(sM[(bead0 + b0 + u) * B + n], pv, det[u]);
      }
      for (int k2 = 0; k2 < (B + 1) / 2; ++k2) {       // modes 2 k2 and 2 k2 + 1 from one Philox block
        uint32_t w[4];
        spk_philox4x32_10((uint32_t)t, (uint32_t)((uint64_t)t >> 32) ^ ((uint32_t)k2 << 8) ^ which, (uint32_t)step, (uint32_t)(step >> 32), seed_lo, seed_hi, w);
        float x0, x1;
        spk_box_muller(w[0], w[1], x0, x1);
#pragma unroll
        for (int u = 0; u < PILE_CHUNK; ++u)
          if (b0 + u < n_local) {
            const float* row = sM + B * B + (bead0 + b0 + u) * B;
            noi[u] = fmaf(row[2 * k2], x0, noi[u]);
            if (2 * k2 + 1 < B) noi[u] = fmaf(row[2 * k2 + 1], x1, noi[u]);
          }
      }
#pragma unroll
      for (int u = 0; u < PILE_CHUNK; ++u)
        if (b0 + u < n_local) p_out[(int64_t)(b0 + u) * n3 + t] = det[u] + sm * noi[u];
    }
  }
}

extern "C" int spk_md_pile_f32(const float* p_all, const float* masses, const float* M, float noise_scale, uint64_t seed,
                               uint64_t step, const int64_t* step_dev, int32_t which, int32_t n_beads, int64_t n_atoms,
                               int32_t bead0, int32_t n_local, float* p_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n_beads >= 1 && n_beads <= 64 && n_atoms >= 0, "spk_md_pile_f32: bad sizes (n_beads <= 64)");
  SPK_CHECK_ARG(bead0 >= 0 && n_local >= 0 && bead0 + n_local <= n_beads, "spk_md_pile_f32: bead range outside [0, n_beads)");
  if (n_atoms == 0 || n_local == 0) return SPK_OK;
  SPK_CHECK_ARG(p_all && masses && M && p_out && p_out != p_all, "spk_md_pile_f32: null pointer / output aliases the input");
  const size_t lds = sizeof(float) * 2 * (size_t)n_beads * n_beads;      // <= 32 KB
  hipLaunchKernelGGL(k_md_pile, dim3(spk_grid_for(3 * n_atoms, 256, spk_num_cus() * 8)), dim3(256), lds, stream, p_all, masses, M,
                     noise_scale, (uint32_t)seed, (uint32_t)(seed >> 32), step, step_dev, (uint32_t)which, n_beads, n_atoms, bead0, n_local, p_out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
