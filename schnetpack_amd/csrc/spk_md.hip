// MD step kernels: the elementwise half/main steps of the reference's
// integrators fused, and the ring-polymer main step as one bead-mixing kernel.  HBM-bound; every
// array is read and written once per step.
#include "spk_common.h"

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
