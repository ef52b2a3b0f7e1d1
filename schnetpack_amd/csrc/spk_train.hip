// Primitives of the TRAINING regime (force matching: loss on -dE/dR => every operator is differentiated twice,
// atomistic/response.py:59-68 with create_graph = training).
//
// The eval regime runs a whole representation in one or two launches; a training step cannot (its backward is itself
// differentiated), so it is assembled from a small family of kernels that is CLOSED under differentiation -- the
// derivative of each is another member, so the recorded backward and the backward of that backward are the same few
// hand-written launches instead of chains of framework element-wise kernels and library GEMMs for 168-row operands:
//
//   act_mul        a . act^(k)(z) [+ c]                    d/da -> act_mul(k),          d/dz -> act_mul(k + 1)
//   gemm_tn        U^T X  (+ column sums of U)             with the Dense kernels (x W^T, u W) closed under d/d(anything)
//   cfconv_edge    y[out_e] += x[src_e] . W_e              d/dx -> cfconv_edge (roles swapped), d/dW -> edge_mul
//   edge_mul       a[ia_e] . b[ib_e]                       d/da, d/db -> cfconv_edge
//   radial_d       a_e phi_r^(k)(d_e)                      d/da -> radial_c(k),         d/dd -> radial_c(k + 1)
//   radial_c       a_e sum_r G_er phi_r^(k)(d_e)           d/dG -> radial_d(k),         d/dd -> radial_c(k + 1)
//   rowscale/rowdot  W_ef s_e  /  sum_f a_ef b_ef          each other's derivatives
//
// (schnet.py:54-69 = Dense, Dense, rowscale, cfconv_edge, Dense, Dense; nn/radial.py, nn/cutoff.py = radial_d.)
#include "spk_common.h"

// ------------------------------------------------------------------------------------------------ activations, order k
// shifted softplus: ssp' = s, ssp'' = s(1-s), ssp''' = s(1-s)(1-2s), ssp'''' = s(1-s)(1-6s+6s^2)   (s = sigmoid)
// silu = z s: silu' = s(1 + z(1-s)), silu'' = s(1-s)(2 + z(1-2s)), silu''' = s(1-s)(3(1-2s) + z(1-6s+6s^2))
__device__ __forceinline__ float act_order(int act, int order, float z) {
  if (act == SPK_ACT_NONE) return order == 0 ? z : (order == 1 ? 1.f : 0.f);
  const float s = spk_sigmoid(z);
  const float s1 = s * (1.f - s);
  if (act == SPK_ACT_SSP) {
    switch (order) {
      case 0: return spk_ssp(z);
      case 1: return s;
      case 2: return s1;
      case 3: return s1 * (1.f - 2.f * s);
      default: return s1 * (1.f - 6.f * s + 6.f * s * s);
    }
  }
  switch (order) {
    case 0: return z * s;
    case 1: return s * (1.f + z * (1.f - s));
    case 2: return s1 * (2.f + z * (1.f - 2.f * s));
    default: return s1 * (3.f * (1.f - 2.f * s) + z * (1.f - 6.f * s + 6.f * s * s));
  }
}

__global__ void k_act_mul(const float* __restrict__ a, const float* __restrict__ z, const float* __restrict__ c, int64_t n, int act, int order,
                          float* __restrict__ out) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    float v = act_order(act, order, z[t]);
    if (a) v *= a[t];
    if (c) v += c[t];
    out[t] = v;
  }
}

extern "C" int spk_act_mul_f32(const float* a, const float* z, const float* c, int64_t n, int32_t act, int32_t order, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n >= 0 && act >= 0 && act <= 2, "spk_act_mul_f32: bad arguments");
  SPK_CHECK_ARG(order >= 0 && order <= (act == SPK_ACT_SILU ? 3 : 4), "spk_act_mul_f32: derivative order %d not provided", order);
  if (n == 0) return SPK_OK;
  SPK_CHECK_ARG(z && out, "spk_act_mul_f32: null pointer");
  SpkProfScope prof("act_mul", stream);
  hipLaunchKernelGGL(k_act_mul, dim3(spk_grid_for(n, 256, spk_num_cus() * 16)), dim3(256), 0, stream, a, z, c, n, act, order, out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------ G = U^T X (weight gradients)
// U [n, O], X [n, K] row-major, G [O, K]; the contraction runs over the n samples, so both MFMA operands are
// lane-contiguous in memory as they lie (lane el of half hi holds U[n = 2s + hi][o0 + el] and X[2s + hi][k0 + el]).
// One workgroup of 8 waves per (32 x 32 tile of G, slice of n): the waves split the slice, meet in LDS, and -- when n needs
// more than one slice -- the slices meet in a workspace where the LAST workgroup of a tile (ticket counter, self-resetting)
// adds them in slice order: deterministic, one launch.  Tiles with k0 == 0 also carry the column sums of U (bias gradient).
#include "spk_gemm_tn.h"
__global__ __launch_bounds__(64 * TN_WAVES) void k_gemm_tn(GemmTnArgs a) { gemm_tn_block(a, blockIdx.x, blockIdx.y); }

// slices / workspace sizes for a problem (host helper shared with the caller that allocates the workspace)
extern "C" int spk_gemm_tn_plan(int64_t n, int32_t O, int32_t K, int32_t* n_slices, int64_t* ws_floats, int32_t* n_tiles) {
  SPK_CHECK_ARG(n >= 0 && O > 0 && K > 0 && n_slices && ws_floats && n_tiles, "spk_gemm_tn_plan: bad arguments");
  const int tiles = ((O + 31) / 32) * ((K + 31) / 32);
  int64_t S = (n + TN_ROWS_PER_BLOCK - 1) / TN_ROWS_PER_BLOCK;
  int64_t cap = (int64_t)(2 * spk_num_cus()) / tiles;                     // two workgroups per CU over the chip
  if (cap < 1) cap = 1;
  if (S > cap) S = cap;
  if (S > 128) S = 128;
  if (S < 1) S = 1;
  *n_slices = (int32_t)S;
  *n_tiles = tiles;
  *ws_floats = S > 1 ? S * (int64_t)tiles * (1024 + 32) : 0;
  return SPK_OK;
}

extern "C" int spk_gemm_tn_f32(const float* U, const float* X, int64_t n, int32_t O, int32_t K, float* G, float* gb, float* ws, uint32_t* tickets,
                               void* stream_) {
  return spk_gemm_tn_nb_f32(U, X, n, O, K, G, gb, n, ws, tickets, stream_);
}

// the same with the bias gradient (column sums of U) restricted to the rows [0, n_bias): operands stacked as [value rows ; tangent rows]
// carry a bias on the value rows only (force-matching engine, spk_fm_engine.h)
extern "C" int spk_gemm_tn_nb_f32(const float* U, const float* X, int64_t n, int32_t O, int32_t K, float* G, float* gb, int64_t n_bias, float* ws,
                                  uint32_t* tickets, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n_bias >= 0 && n_bias <= n, "spk_gemm_tn_nb_f32: n_bias outside [0, n]");
  int32_t S, tiles;
  int64_t wsf;
  int rc = spk_gemm_tn_plan(n, O, K, &S, &wsf, &tiles);
  if (rc) return rc;
  SPK_CHECK_ARG(G != nullptr && (n == 0 || (U && X)), "spk_gemm_tn_f32: null pointer");
  SPK_CHECK_ARG(S == 1 || (ws && tickets), "spk_gemm_tn_f32: workspace / ticket buffer required for %d slices", S);
  SPK_CHECK_ARG(tiles <= 4096, "spk_gemm_tn_f32: %d output tiles (max 4096)", tiles);
  SpkProfScope prof("gemm_tn", stream);
  GemmTnArgs a = spk_gemm_tn_args(U, X, n, O, K, S, tiles, G, gb, ws, tickets);
  a.nb = n_bias;
  hipLaunchKernelGGL(k_gemm_tn, dim3(tiles, S), dim3(64 * TN_WAVES), 0, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------ cfconv on materialised filters
// y[out_e, :] += x[src_e, :] . W[e, :]   (schnet.py:64-66: x_j * Wij, scatter_add over idx_i)
// one workgroup per output row: threads = (feature chunks) x (edge slots); the slots walk the row's pairs together and meet
// in LDS, so a long row (the padded tail of a static-shape training batch, a dense neighbourhood) costs its length / slots
template <int V>
__global__ __launch_bounds__(256) void k_cfconv_rows(const float* __restrict__ x, const float* __restrict__ W, const int32_t* __restrict__ rowptr,
                                                     const int64_t* __restrict__ src, int64_t n_out, int64_t n_src, int F, int chunks, int slots,
                                                     float* __restrict__ y) {
  __shared__ float red[256 * V];
  const int tid = threadIdx.x;
  const int slot = tid / chunks, ch = tid % chunks;
  for (int64_t row = blockIdx.x; row < n_out; row += gridDim.x) {
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    for (int c0 = 0; c0 < F; c0 += chunks * V) {
      const int c = c0 + ch * V;
      float acc[V];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = 0.f;
      if (slot < slots && c < F) {
        for (int e = e0 + slot; e < e1; e += slots) {
          const int64_t j = src ? src[e] : e;
          if ((uint64_t)j >= (uint64_t)n_src) continue;
          if (V == 4) {
            const f32x4 xv = *(const f32x4*)(x + j * F + c), wv = *(const f32x4*)(W + (int64_t)e * F + c);
            acc[0] = fmaf(xv.x, wv.x, acc[0]); acc[1 % V] = fmaf(xv.y, wv.y, acc[1 % V]);
            acc[2 % V] = fmaf(xv.z, wv.z, acc[2 % V]); acc[3 % V] = fmaf(xv.w, wv.w, acc[3 % V]);
          } else {
            acc[0] = fmaf(x[j * F + c], W[(int64_t)e * F + c], acc[0]);
          }
        }
      }
#pragma unroll
      for (int v = 0; v < V; ++v) red[tid * V + v] = acc[v];
      __syncthreads();
      if (slot == 0 && c < F) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float t = 0.f;
          for (int q = 0; q < slots; ++q) t += red[(q * chunks + ch) * V + v];      // fixed order: deterministic
          y[row * F + c + v] = t;
        }
      }
      __syncthreads();
    }
  }
}

__global__ void k_cfconv_atomic(const float* __restrict__ x, const float* __restrict__ W, const int64_t* __restrict__ out, const int64_t* __restrict__ src,
                                int64_t E, int64_t n_out, int64_t n_src, int F, float* __restrict__ y) {
  const int64_t total = E * F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = t / F;
    const int f = (int)(t % F);
    const int64_t i = out[e], j = src ? src[e] : e;
    if ((uint64_t)i >= (uint64_t)n_out || (uint64_t)j >= (uint64_t)n_src) continue;
    unsafeAtomicAdd(y + i * F + f, x[j * F + f] * W[t]);
  }
}

extern "C" int spk_cfconv_edge_f32(const float* x, const float* W, const int64_t* idx_out, const int64_t* idx_src, const int32_t* rowptr_out, int64_t E,
                                   int64_t n_out, int64_t n_src, int32_t F, float* y, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(E >= 0 && n_out >= 0 && n_src >= 0 && F > 0, "spk_cfconv_edge_f32: bad sizes");
  if (n_out == 0) return SPK_OK;
  SPK_CHECK_ARG(y != nullptr, "spk_cfconv_edge_f32: null output");
  SPK_CHECK_ARG(E == 0 || (x && W && idx_out), "spk_cfconv_edge_f32: null pointer");
  SPK_CHECK_ARG(E < (1ll << 31), "spk_cfconv_edge_f32: more than 2^31 pairs");
  SpkProfScope prof("cfconv_edge", stream);
  const int maxb = spk_num_cus() * 16;
  if (rowptr_out) {
    const bool v4 = (F % 4 == 0) && (((uintptr_t)x | (uintptr_t)W | (uintptr_t)y) % 16 == 0);
    const int per = v4 ? F / 4 : F;
    const int chunks = per < 256 ? per : 256;
    int slots = 256 / chunks;
    if (slots > 16) slots = 16;
    const int grid = spk_grid_for(n_out, 1, maxb * 4);
    if (v4) hipLaunchKernelGGL(k_cfconv_rows<4>, dim3(grid), dim3(256), 0, stream, x, W, rowptr_out, idx_src, n_out, n_src, F, chunks, slots, y);
    else hipLaunchKernelGGL(k_cfconv_rows<1>, dim3(grid), dim3(256), 0, stream, x, W, rowptr_out, idx_src, n_out, n_src, F, chunks, slots, y);
  } else {
    int rc = spk_zero_async(y, (size_t)n_out * F * sizeof(float), stream);
    if (rc) return rc;
    if (E > 0) hipLaunchKernelGGL(k_cfconv_atomic, dim3(spk_grid_for(E * F, 256, maxb)), dim3(256), 0, stream, x, W, idx_out, idx_src, E, n_out, n_src, F, y);
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// out[e, :] = a[ia_e, :] . b[ib_e, :]
template <int V>
__global__ void k_edge_mul(const float* __restrict__ a, const float* __restrict__ b, const int64_t* __restrict__ ia, const int64_t* __restrict__ ib,
                           int64_t E, int64_t na, int64_t nb, int F, float* __restrict__ out) {
  const int FV = F / V;
  const int64_t total = E * FV;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = t / FV;
    const int c = (int)(t % FV) * V;
    const int64_t i = ia ? ia[e] : e, j = ib ? ib[e] : e;
    const bool ok = (uint64_t)i < (uint64_t)na && (uint64_t)j < (uint64_t)nb;
    if (V == 4) {
      f32x4 o{0.f, 0.f, 0.f, 0.f};
      if (ok) o = *(const f32x4*)(a + i * F + c) * *(const f32x4*)(b + j * F + c);
      *(f32x4*)(out + e * F + c) = o;
    } else {
      out[e * F + c] = ok ? a[i * F + c] * b[j * F + c] : 0.f;
    }
  }
}

extern "C" int spk_edge_mul_f32(const float* a, const float* b, const int64_t* idx_a, const int64_t* idx_b, int64_t E, int64_t na, int64_t nb, int32_t F,
                                float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(E >= 0 && na >= 0 && nb >= 0 && F > 0, "spk_edge_mul_f32: bad sizes");
  if (E == 0) return SPK_OK;
  SPK_CHECK_ARG(a && b && out, "spk_edge_mul_f32: null pointer");
  SpkProfScope prof("edge_mul", stream);
  const int maxb = spk_num_cus() * 16;
  const bool v4 = (F % 4 == 0) && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16 == 0);
  if (v4) hipLaunchKernelGGL(k_edge_mul<4>, dim3(spk_grid_for(E * (F / 4), 256, maxb)), dim3(256), 0, stream, a, b, idx_a, idx_b, E, na, nb, F, out);
  else hipLaunchKernelGGL(k_edge_mul<1>, dim3(spk_grid_for(E * F, 256, maxb)), dim3(256), 0, stream, a, b, idx_a, idx_b, E, na, nb, F, out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------ radial functions, order k
// kind 0: Gaussian exp(c t^2), t = d - mu, c = -1/(2 w^2)   (nn/radial.py:11-15)
// kind 1: Bessel sin(f d) / d                                (nn/radial.py:105-110; value 0 at d = 0 like the reference's guard)
// kind 2: cosine cutoff 0.5 (cos(pi d / rc) + 1) [d < rc]    (nn/cutoff.py:14-33), one "basis function"
__device__ __forceinline__ float radial_order(const RadialDev& rb, int r, int order, float d) {
  if (rb.kind == SPK_RBF_GAUSSIAN) {
    const float w = rb.p1[r];
    const float c = -0.5f / (w * w);
    const float t = d - rb.p0[r];
    const float phi = expf(c * t * t);
    const float u = 2.f * c * t;                 // phi' / phi
    switch (order) {
      case 0: return phi;
      case 1: return u * phi;
      case 2: return (2.f * c + u * u) * phi;
      default: return (6.f * c * u + u * u * u) * phi;
    }
  }
  if (rb.kind == SPK_RBF_BESSEL) {
    if (d == 0.f) return 0.f;
    const float f = rb.p0[r];
    float s, co;
    sincosf(f * d, &s, &co);
    const float q = 1.f / d;
    switch (order) {
      case 0: return s * q;
      case 1: return (f * co - s * q) * q;
      case 2: return ((2.f * q * q - f * f) * s - 2.f * f * q * co) * q;
      default: return ((6.f * q * q - f * f) * f * co + (3.f * f * f - 6.f * q * q) * q * s) * q;
    }
  }
  if (!(d < rb.cutoff)) return 0.f;
  const float a = SPK_PI_F / rb.cutoff;
  float s, co;
  sincosf(a * d, &s, &co);
  switch (order) {
    case 0: return 0.5f * (co + 1.f);
    case 1: return -0.5f * a * s;
    case 2: return -0.5f * a * a * co;
    default: return 0.5f * a * a * a * s;
  }
}

__global__ void k_radial_d(const float* __restrict__ d, const float* __restrict__ a, int64_t n, RadialDev rb, int R, int order, float* __restrict__ out) {
  const int64_t total = n * R;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = t / R;
    float v = radial_order(rb, (int)(t % R), order, d[e]);
    if (a) v *= a[e];
    out[t] = v;
  }
}

__global__ void k_radial_c(const float* __restrict__ G, const float* __restrict__ d, const float* __restrict__ a, int64_t n, RadialDev rb, int R, int order,
                           float* __restrict__ out) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float dd = d[e];
    float acc = 0.f;
    for (int r = 0; r < R; ++r) acc = fmaf(G[e * R + r], radial_order(rb, r, order, dd), acc);
    out[e] = a ? acc * a[e] : acc;
  }
}

static int check_radial_k(const spk_radial_t* rb, int order, const char* who) {
  SPK_CHECK_ARG(rb != nullptr, "%s: null radial description", who);
  SPK_CHECK_ARG(rb->kind >= 0 && rb->kind <= 2, "%s: unknown radial kind %d", who, rb->kind);
  SPK_CHECK_ARG(order >= 0 && order <= 3, "%s: derivative order %d not provided", who, order);
  if (rb->kind == 2) {
    SPK_CHECK_ARG(rb->cutoff > 0.f, "%s: cutoff must be positive", who);
  } else {
    SPK_CHECK_ARG(rb->n_rbf >= 1 && rb->n_rbf <= 1024, "%s: n_rbf=%d unsupported", who, rb->n_rbf);
    SPK_CHECK_ARG(rb->p0 != nullptr && (rb->kind == SPK_RBF_BESSEL || rb->p1 != nullptr), "%s: null rbf parameters", who);
  }
  return SPK_OK;
}

extern "C" int spk_radial_d_f32(const float* d, const float* a, int64_t n, const spk_radial_t* rb, int32_t order, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_radial_k(rb, order, "spk_radial_d_f32");
  if (rc) return rc;
  if (n <= 0) return SPK_OK;
  SPK_CHECK_ARG(d && out, "spk_radial_d_f32: null pointer");
  SpkProfScope prof("radial_d", stream);
  const int R = rb->kind == 2 ? 1 : rb->n_rbf;
  hipLaunchKernelGGL(k_radial_d, dim3(spk_grid_for(n * R, 256, spk_num_cus() * 16)), dim3(256), 0, stream, d, a, n, spk_radial_dev(rb), R, order, out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_radial_c_f32(const float* G, const float* d, const float* a, int64_t n, const spk_radial_t* rb, int32_t order, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_radial_k(rb, order, "spk_radial_c_f32");
  if (rc) return rc;
  if (n <= 0) return SPK_OK;
  SPK_CHECK_ARG(G && d && out, "spk_radial_c_f32: null pointer");
  SpkProfScope prof("radial_c", stream);
  const int R = rb->kind == 2 ? 1 : rb->n_rbf;
  hipLaunchKernelGGL(k_radial_c, dim3(spk_grid_for(n, 256, spk_num_cus() * 16)), dim3(256), 0, stream, G, d, a, n, spk_radial_dev(rb), R, order, out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------ row scale / row dot
__global__ void k_rowscale(const float* __restrict__ W, const float* __restrict__ s, int64_t rows, int F, float* __restrict__ out) {
  const int64_t total = rows * F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) out[t] = W[t] * s[t / F];
}

// one wave per row
__global__ __launch_bounds__(256) void k_rowdot(const float* __restrict__ a, const float* __restrict__ b, int64_t rows, int F, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    float acc = 0.f;
    for (int f = lane; f < F; f += 64) acc = fmaf(a[row * F + f], b[row * F + f], acc);
    acc = spk_wave_sum(acc);
    if (lane == 0) out[row] = acc;
  }
}

extern "C" int spk_rowscale_f32(const float* W, const float* s, int64_t rows, int32_t F, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(rows >= 0 && F > 0, "spk_rowscale_f32: bad sizes");
  if (rows == 0) return SPK_OK;
  SPK_CHECK_ARG(W && s && out, "spk_rowscale_f32: null pointer");
  SpkProfScope prof("rowscale", stream);
  hipLaunchKernelGGL(k_rowscale, dim3(spk_grid_for(rows * F, 256, spk_num_cus() * 16)), dim3(256), 0, stream, W, s, rows, F, out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_rowdot_f32(const float* a, const float* b, int64_t rows, int32_t F, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(rows >= 0 && F > 0, "spk_rowdot_f32: bad sizes");
  if (rows == 0) return SPK_OK;
  SPK_CHECK_ARG(a && b && out, "spk_rowdot_f32: null pointer");
  SpkProfScope prof("rowdot", stream);
  hipLaunchKernelGGL(k_rowdot, dim3(spk_grid_for(rows, 4, spk_num_cus() * 16)), dim3(256), 0, stream, a, b, rows, F, out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------ force-matching loss
// loss = wE mean((E - E_t)^2) + wF mean((F - F_t)^2)  (ModelOutput.calculate_loss + AtomisticTask.loss_fn, task.py:59-66, 142-146, with two MSE outputs and the weights of
// the example configs) and its gradients w.r.t. E and F in ONE launch -- as framework arithmetic the two terms and their backward are
// 19 launches of a training step that is launch-latency bound.  One workgroup (the operands are N * 3 + M numbers).
__global__ __launch_bounds__(256) void k_fm_loss(const float* __restrict__ E, const float* __restrict__ Et, int64_t M, const float* __restrict__ F,
                                                  const float* __restrict__ Ft, int64_t n3, float wE, float wF, float* __restrict__ loss,
                                                  float* __restrict__ gE, float* __restrict__ gF) {
  __shared__ float red[256];
  const float cE = M > 0 ? wE / (float)M : 0.f, cF = n3 > 0 ? wF / (float)n3 : 0.f;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < M; i += 256) { const float d = E[i] - Et[i]; acc += cE * d * d; gE[i] = 2.f * cE * d; }
  for (int64_t i = threadIdx.x; i < n3; i += 256) { const float d = F[i] - Ft[i]; acc += cF * d * d; gF[i] = 2.f * cF * d; }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) loss[0] = red[0];
}
// (gE, gF) * g[0] -> (outE, outF): the backward of the node, one launch for both operands
__global__ void k_fm_loss_bwd(const float* __restrict__ g, const float* __restrict__ gE, int64_t M, const float* __restrict__ gF, int64_t n3,
                              float* __restrict__ outE, float* __restrict__ outF) {
  const float s = g[0];
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < M + n3; t += (int64_t)gridDim.x * blockDim.x) {
    if (t < M) outE[t] = s * gE[t];
    else outF[t - M] = s * gF[t - M];
  }
}
extern "C" int spk_fm_loss_f32(const float* E, const float* Et, int64_t M, const float* F, const float* Ft, int64_t n3, float wE, float wF,
                               float* loss, float* gE, float* gF, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(M >= 0 && n3 >= 0 && loss, "spk_fm_loss_f32: bad sizes / null loss");
  SPK_CHECK_ARG((M == 0 || (E && Et && gE)) && (n3 == 0 || (F && Ft && gF)), "spk_fm_loss_f32: null pointer");
  SpkProfScope prof("fm_loss", stream);
  hipLaunchKernelGGL(k_fm_loss, dim3(1), dim3(256), 0, stream, E, Et, M, F, Ft, n3, wE, wF, loss, gE, gF);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
extern "C" int spk_fm_loss_bwd_f32(const float* g, const float* gE, int64_t M, const float* gF, int64_t n3, float* outE, float* outF, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(M >= 0 && n3 >= 0, "spk_fm_loss_bwd_f32: bad sizes");
  if (M + n3 == 0) return SPK_OK;
  SPK_CHECK_ARG(g && (M == 0 || (gE && outE)) && (n3 == 0 || (gF && outF)), "spk_fm_loss_bwd_f32: null pointer");
  SpkProfScope prof("fm_loss_bwd", stream);
  hipLaunchKernelGGL(k_fm_loss_bwd, dim3(spk_grid_for(M + n3, 256, 64)), dim3(256), 0, stream, g, gE, M, gF, n3, outE, outF);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------ AdamW over one flat gradient bucket
// The optimizer of the reference's training configs (torch.optim.AdamW through AtomisticTask.configure_optimizers, task.py:187-199) as ONE
// launch for all parameters: the gradients already sit in one flat bucket (FlatGradAllReduce), the moments are flat buffers of the same
// layout, the parameters stay the model's own tensors and are reached through a chunk table.  The framework's multi-tensor kernel takes
// 13 us (SchNet, 30 tensors) / 31 us (PaiNN, 34 tensors) per step plus a launch that advances the step counters.
// Arithmetic of torch.optim.AdamW (decoupled weight decay, bias corrections from the step count, eps outside the corrected root).
// The step count lives on the device (graph replays): every workgroup reads it, the LAST one to finish writes count + 1.
__global__ __launch_bounds__(256) void k_adamw(const spk_adamw_chunk_t* __restrict__ chunks, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, float* step, unsigned* ticket, float lr, const float* __restrict__ lr_dev, float b1, float b2, float eps, float wd) {
  __shared__ float s_t;
  if (threadIdx.x == 0) s_t = *(volatile float*)step + 1.0f;
  __syncthreads();
  const float t = s_t;
  if (lr_dev) lr = *lr_dev;     // learning rate of a schedule: read from the device so that a captured step follows it
  const float bc1 = 1.0f - powf(b1, t), bc2s = sqrtf(1.0f - powf(b2, t));
  const float step_size = lr / bc1, decay = 1.0f - lr * wd;
  const spk_adamw_chunk_t c = chunks[blockIdx.x];
  float* __restrict__ p = (float*)c.param;
  for (int i = threadIdx.x; i < c.n; i += 256) {
    const int64_t f = c.offset + i;
    const float gi = g[f];
    const float mi = b1 * m[f] + (1.0f - b1) * gi;
    const float vi = b2 * v[f] + (1.0f - b2) * gi * gi;
    m[f] = mi;
    v[f] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[c.poffset + i] = p[c.poffset + i] * decay - step_size * (mi / denom);
  }
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
    *ticket = 0u;
    *step = t;
  }
}
static int adamw_launch(const spk_adamw_chunk_t* chunks, int64_t n_chunks, const float* grads, float* exp_avg, float* exp_avg_sq, float* step, uint32_t* ticket, float lr,
                        const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, hipStream_t stream) {
  SPK_CHECK_ARG(n_chunks >= 0 && n_chunks < (1ll << 31), "spk_adamw_f32: bad chunk count");
  if (n_chunks == 0) return SPK_OK;
  SPK_CHECK_ARG(chunks && grads && exp_avg && exp_avg_sq && step && ticket, "spk_adamw_f32: null pointer");
  SPK_CHECK_ARG(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && weight_decay >= 0.f, "spk_adamw_f32: bad hyper-parameters");
  SpkProfScope prof("adamw", stream);
  hipLaunchKernelGGL(k_adamw, dim3((unsigned)n_chunks), dim3(256), 0, stream, chunks, grads, exp_avg, exp_avg_sq, step, (unsigned*)ticket, lr, lr_dev, beta1, beta2, eps,
                     weight_decay);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
extern "C" int spk_adamw_f32(const spk_adamw_chunk_t* chunks, int64_t n_chunks, const float* grads, float* exp_avg, float* exp_avg_sq, float* step,
                             uint32_t* ticket, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream_) {
  return adamw_launch(chunks, n_chunks, grads, exp_avg, exp_avg_sq, step, ticket, lr, nullptr, beta1, beta2, eps, weight_decay, (hipStream_t)stream_);
}
extern "C" int spk_adamw_devlr_f32(const spk_adamw_chunk_t* chunks, int64_t n_chunks, const float* grads, float* exp_avg, float* exp_avg_sq, float* step,
                                   uint32_t* ticket, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, void* stream_) {
  SPK_CHECK_ARG(lr_dev != nullptr, "spk_adamw_devlr_f32: null learning-rate pointer");
  return adamw_launch(chunks, n_chunks, grads, exp_avg, exp_avg_sq, step, ticket, 0.f, lr_dev, beta1, beta2, eps, weight_decay, (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------------------------ 3-vector algebra (PaiNN, painn.py:55-66, 99-117)
// V-type operands are [M, 3, F] (Cartesian component in the middle), s-type [M, F], u-type [M, 3]; every V / s operand comes with a
// row stride (ld) so that the halves of a split tensor are read in place.  Five kernels, closed under differentiation:
//   vscale(V, s)[m,k,f] = V s        d/dV -> vscale(g, s)    d/ds -> vdot(g, V)
//   vdot(A, B)[m,f] = sum_k A B      d/dA -> vscale(B, g)    d/dB -> vscale(A, g)
//   vouter(s, u)[m,k,f] = s u_k      d/ds -> vcontract(g, u) d/du -> vrowdot(g, s)
//   vcontract(G, u)[m,f] = sum_k G u_k   d/dG -> vouter(g, u)    d/du -> vrowdot(G, g)
//   vrowdot(G, s)[m,k] = sum_f G s       d/dG -> vouter(s, g)    d/ds -> vcontract(G, g)
__global__ void k_vscale(const float* __restrict__ V, int64_t ldV, const float* __restrict__ s, int64_t lds, int64_t M, int F, float* __restrict__ out) {
  const int64_t total = M * 3 * F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(t % F);
    const int64_t row = t / F, m = row / 3;
    out[t] = V[row * ldV + f] * s[m * lds + f];
  }
}
__global__ void k_vdot(const float* __restrict__ A, int64_t ldA, const float* __restrict__ B, int64_t ldB, int64_t M, int F, float* __restrict__ out) {
  const int64_t total = M * F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(t % F);
    const int64_t m = t / F;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) acc = fmaf(A[(3 * m + k) * ldA + f], B[(3 * m + k) * ldB + f], acc);
    out[t] = acc;
  }
}
__global__ void k_vouter(const float* __restrict__ s, int64_t lds, const float* __restrict__ u, int64_t M, int F, float* __restrict__ out) {
  const int64_t total = M * 3 * F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(t % F);
    const int64_t row = t / F, m = row / 3;
    out[t] = s[m * lds + f] * u[row];
  }
}
__global__ void k_vcontract(const float* __restrict__ G, int64_t ldG, const float* __restrict__ u, int64_t M, int F, float* __restrict__ out) {
  const int64_t total = M * F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(t % F);
    const int64_t m = t / F;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) acc = fmaf(G[(3 * m + k) * ldG + f], u[3 * m + k], acc);
    out[t] = acc;
  }
}
// one wave per (m, k) row
__global__ __launch_bounds__(256) void k_vrowdot(const float* __restrict__ G, int64_t ldG, const float* __restrict__ s, int64_t lds, int64_t M, int F,
                                                 float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6); row < 3 * M; row += (int64_t)gridDim.x * 4) {
    const int64_t m = row / 3;
    float acc = 0.f;
    for (int f = lane; f < F; f += 64) acc = fmaf(G[row * ldG + f], s[m * lds + f], acc);
    acc = spk_wave_sum(acc);
    if (lane == 0) out[row] = acc;
  }
}

extern "C" int spk_vec3_f32(int32_t op, const float* A, int64_t ldA, const float* B, int64_t ldB, int64_t M, int32_t F, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(op >= 0 && op <= 4 && M >= 0 && F > 0, "spk_vec3_f32: bad arguments");
  if (M == 0) return SPK_OK;
  SPK_CHECK_ARG(A && B && out, "spk_vec3_f32: null pointer");
  SpkProfScope prof("vec3", stream);
  const int maxb = spk_num_cus() * 16;
  switch (op) {
    case SPK_VEC3_SCALE: hipLaunchKernelGGL(k_vscale, dim3(spk_grid_for(M * 3 * F, 256, maxb)), dim3(256), 0, stream, A, ldA, B, ldB, M, F, out); break;
    case SPK_VEC3_DOT: hipLaunchKernelGGL(k_vdot, dim3(spk_grid_for(M * F, 256, maxb)), dim3(256), 0, stream, A, ldA, B, ldB, M, F, out); break;
    case SPK_VEC3_OUTER: hipLaunchKernelGGL(k_vouter, dim3(spk_grid_for(M * 3 * F, 256, maxb)), dim3(256), 0, stream, A, ldA, B, M, F, out); break;
    case SPK_VEC3_CONTRACT: hipLaunchKernelGGL(k_vcontract, dim3(spk_grid_for(M * F, 256, maxb)), dim3(256), 0, stream, A, ldA, B, M, F, out); break;
    default: hipLaunchKernelGGL(k_vrowdot, dim3(spk_grid_for(3 * M, 4, maxb)), dim3(256), 0, stream, A, ldA, B, ldB, M, F, out); break;
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
