// Shared device/host helpers for the gfx950 kernels.  Wavefront = 64 lanes everywhere.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include "../../include/spk_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SPK_PI_F 3.14159265358979323846f
#define SPK_LN2_F 0.69314718055994530942f

// ---------------------------------------------------------------- host error handling
void spk_set_error(const char* fmt, ...);

#define SPK_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      spk_set_error(__VA_ARGS__);       \
      return SPK_ERR_ARG;               \
    }                                   \
  } while (0)

#define SPK_HIP_TRY(expr)                                                             \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      spk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                    __LINE__);                                                        \
      return SPK_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)

#define SPK_LAUNCH_CHECK() SPK_HIP_TRY(hipGetLastError())

static inline int spk_grid_for(int64_t work_items, int per_block, int max_blocks) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

int spk_num_cus();

// Zero `bytes` (a multiple of 4) bytes with a KERNEL launch.  hipMemsetAsync must not be used in this library:
// captured into a HIP graph it becomes a memset node, and on ROCm 7.2 replays of such graphs left the target
// un-cleared from the second replay on (NaN energies / forces in replayed force calls).
int spk_zero_async(void* p, size_t bytes, hipStream_t stream);

// ---------------------------------------------------------------- per-kernel HIP-event profiling
// (off by default; bench.py enables it for a separate pass to time individual kernels on the
// stream they are launched on)
bool spk_prof_enabled();
void spk_prof_begin(const char* tag, hipStream_t stream);
void spk_prof_end(hipStream_t stream);
struct SpkProfScope {
  hipStream_t s; bool on;
  SpkProfScope(const char* tag, hipStream_t stream) : s(stream), on(spk_prof_enabled()) { if (on) spk_prof_begin(tag, s); }
  ~SpkProfScope() { if (on) spk_prof_end(s); }
};

// ---------------------------------------------------------------- device math
// softplus(x) = max(x,0) + log(1 + exp(-|x|)) and sigmoid(x) from one exp.
// Equivalent to torch's softplus(beta=1, threshold=20) to fp32 round-off.
__device__ __forceinline__ void spk_softplus_sigmoid(float x, float& sp, float& sg) {
  float t = __expf(-fabsf(x));
  float one_t = 1.0f + t;
  float inv = __frcp_rn(one_t);
  sp = fmaxf(x, 0.0f) + __logf(one_t);
  sg = (x >= 0.0f) ? inv : t * inv;
}
__device__ __forceinline__ float spk_ssp(float x) {
  float t = __expf(-fabsf(x));
  return fmaxf(x, 0.0f) + __logf(1.0f + t) - SPK_LN2_F;
}
__device__ __forceinline__ float spk_sigmoid(float x) {
  float t = __expf(-fabsf(x));
  float inv = __frcp_rn(1.0f + t);
  return (x >= 0.0f) ? inv : t * inv;
}
// Hardware-transcendental variant for the MFMA kernels (64 activations per lane and tile):
// v_exp_f32 / v_log_f32 / v_rcp_f32 directly (1 ulp each, no denormal fix-ups: the arguments are
// exp2 of a non-positive number and log2 of a value in [1, 2]).
__device__ __forceinline__ void spk_fast_softplus_sigmoid(float x, float& sp, float& sg) {
  const float t = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
  const float one_t = 1.0f + t;
  const float inv = __builtin_amdgcn_rcpf(one_t);
  sp = fmaf(0.69314718055994531f, __builtin_amdgcn_logf(one_t), fmaxf(x, 0.0f));
  sg = (x >= 0.0f) ? inv : t * inv;
}
__device__ __forceinline__ float spk_fast_ssp(float x) {
  const float t = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
  return fmaf(0.69314718055994531f, __builtin_amdgcn_logf(1.0f + t), fmaxf(x, 0.0f)) - SPK_LN2_F;
}

template <int ACT>
__device__ __forceinline__ float spk_act(float x) {
  if (ACT == SPK_ACT_SSP) return spk_ssp(x);
  if (ACT == SPK_ACT_SILU) return x * spk_sigmoid(x);
  return x;
}
// derivative of the activation at pre-activation x
template <int ACT>
__device__ __forceinline__ float spk_act_grad(float x) {
  if (ACT == SPK_ACT_SSP) return spk_sigmoid(x);
  if (ACT == SPK_ACT_SILU) {
    float s = spk_sigmoid(x);
    return s * (1.0f + x * (1.0f - s));
  }
  return 1.0f;
}

// Device-side copy of the radial description (passed by value to kernels).
struct RadialDev {
  int kind;
  int n_rbf;
  const float* p0;
  const float* p1;
  float cutoff;
};
static inline RadialDev spk_radial_dev(const spk_radial_t* rb) {
  RadialDev r;
  r.kind = rb->kind; r.n_rbf = rb->n_rbf; r.p0 = rb->p0; r.p1 = rb->p1; r.cutoff = rb->cutoff;
  return r;
}

// phi_k(d) and d phi_k / dd  (nn/radial.py:11-15 gaussian, :105-110 bessel)
__device__ __forceinline__ void spk_rbf_eval(const RadialDev& rb, int k, float d, float& phi,
                                             float& dphi) {
  if (k >= rb.n_rbf) { phi = 0.f; dphi = 0.f; return; }
  if (rb.kind == SPK_RBF_GAUSSIAN) {
    float w = rb.p1[k];
    float c = -0.5f / (w * w);
    float t = d - rb.p0[k];
    phi = expf(c * t * t);
    dphi = 2.0f * c * t * phi;
  } else {
    float om = rb.p0[k];
    float s, co;
    sincosf(om * d, &s, &co);
    if (d == 0.0f) { phi = s; dphi = 0.f; }
    else { float inv = 1.0f / d; phi = s * inv; dphi = (om * co - phi) * inv; }
  }
}
// cosine cutoff and derivative (nn/cutoff.py:14-33)
__device__ __forceinline__ void spk_cutoff_eval(float rc, float d, float& f, float& df) {
  if (d < rc) {
    float s, c;
    float a = SPK_PI_F / rc;
    sincosf(d * a, &s, &c);
    f = 0.5f * (c + 1.0f);
    df = -0.5f * a * s;
  } else { f = 0.f; df = 0.f; }
}

// Hardware-transcendental variants for the MFMA edge kernels (v_exp_f32, v_sin_f32, v_cos_f32: ~1e-6
// absolute; arguments are bounded: Gaussian exponent <= 0, angles < n_rbf/2 revolutions).  They keep
// the per-tile set-up short and register-light; the simple kernels and the element-wise entry points
// use the accurate library versions (the two families are cross-checked by the tests).
__device__ __forceinline__ void spk_rbf_eval_fast(const RadialDev& rb, int k, float d, float& phi, float& dphi) {
  if (k >= rb.n_rbf) { phi = 0.f; dphi = 0.f; return; }
  if (rb.kind == SPK_RBF_GAUSSIAN) {
    const float w = rb.p1[k];
    const float c = -0.5f * __builtin_amdgcn_rcpf(w * w);
    const float t = d - rb.p0[k];
    phi = __builtin_amdgcn_exp2f(1.4426950408889634f * c * t * t);
    dphi = 2.0f * c * t * phi;
  } else {
    const float om = rb.p0[k];
    const float rev = om * d * 0.15915494309189535f;   // angle in revolutions
    const float s = __builtin_amdgcn_sinf(rev), co = __builtin_amdgcn_cosf(rev);
    if (d == 0.0f) { phi = s; dphi = 0.f; }
    else { const float inv = __builtin_amdgcn_rcpf(d); phi = s * inv; dphi = (om * co - phi) * inv; }
  }
}
__device__ __forceinline__ void spk_cutoff_eval_fast(float rc, float d, float& f, float& df) {
  if (d < rc) {
    const float inv = __builtin_amdgcn_rcpf(rc);
    const float rev = 0.5f * d * inv;                   // pi d / rc in revolutions
    f = 0.5f * (__builtin_amdgcn_cosf(rev) + 1.0f);
    df = -0.5f * SPK_PI_F * inv * __builtin_amdgcn_sinf(rev);
  } else { f = 0.f; df = 0.f; }
}

// XCD-contiguous walk of `ntiles` work items by persistent workgroups (needs gridDim.x % 8 == 0).  Workgroups are handed round-robin to
// the 8 XCDs, each with its own 4 MiB L2: with tile = blockIdx.x + n gridDim.x every XCD sees every 8th tile of the window the chip is
// working on, so the rows that neighbouring tiles gather are fetched into up to 8 L2s; here the workgroups of XCD x walk the eighth
// [x per, (x + 1) per) of the tiles.  Returns ntiles when the workgroup has run out.  Measured on the 32k-atom water box
// (profiles/r04_tile_experiments.txt): PaiNN tile forward 666 -> 630 us, row forward 859 -> 821 us.
__device__ __forceinline__ int64_t spk_xcd_tile(int nidx, int64_t ntiles) {
  const int64_t per = (ntiles + 7) / 8;
  const int64_t tl = (int64_t)(blockIdx.x >> 3) + (int64_t)nidx * (gridDim.x >> 3);
  const int64_t t = (int64_t)(blockIdx.x & 7) * per + tl;
  return (tl < per && t < ntiles) ? t : ntiles;
}
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a function ON ONE DEVICE: a launcher keeps one static SpkPerDevice per kernel
// and sets the attribute the first time it runs on each device (a process that drives several GPUs -- bead-parallel MD, one process per
// node in tests -- would otherwise fail its first large-LDS launch on the second one).  Host side, not a stream operation.
struct SpkPerDevice {
  std::atomic<uint64_t> word[4];      // bit d of word d / 64 (static storage: zero)
  bool pending(int* dev) {
    if (hipGetDevice(dev) != hipSuccess || *dev < 0) *dev = 0;
    return !(word[(*dev >> 6) & 3].load(std::memory_order_acquire) & (1ull << (*dev & 63)));
  }
  void mark(int dev) { word[(dev >> 6) & 3].fetch_or(1ull << (dev & 63), std::memory_order_release); }
};

// a pointer the caller knows to be the same on every lane of the wavefront, moved to scalar registers (loads through it take the
// "scalar base + 32-bit lane offset" form)
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* spk_uniform_ptr(const T* p) {      // (typed as global memory: a plain pointer rebuilt from integers loads as "flat")
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const __attribute__((address_space(1))) T*)(((uint64_t)hi << 32) | lo);
}
static inline int spk_xcd_walk_default() {
  static const int v = [] { const char* e = getenv("SPK_XCD_WALK"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}

__device__ __forceinline__ float spk_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float spk_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__device__ __forceinline__ float spk_readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// wave-local LDS hand-off between lanes of ONE wavefront (no other wave touches the buffer)
__device__ __forceinline__ void spk_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
