// G = U^T X with column sums of U: the workgroup body shared by k_gemm_tn (spk_train.hip) and k_gemm_pair (spk_dense.hip).
#pragma once
#include "spk_common.h"

#define TN_BATCH 16
#define TN_WAVES 8
#define TN_ROWS_PER_BLOCK 512
struct GemmTnArgs {
  const float* U; const float* X; int64_t n; int O, K, tiles_k, S, n_tiles; int64_t rows_per_slice, rows_per_wave;
  int64_t nb;   // the column sums of U (bias gradient) run over rows [0, nb) only: [value ; tangent]-stacked operands carry a bias on the value rows
  float* G; float* gb; float* ws; float* wsb; unsigned* tickets;
  int defer;    // S > 1: leave the partial tiles in ws / wsb and return -- a following launch adds them up (gemm_tn_reduce_block): no fence, no tickets
};
// one workgroup of 64 * TN_WAVES threads = (tile, slice s); n_tiles = number of output tiles (the grid width of the stand-alone launch)
__device__ __forceinline__ void gemm_tn_block(const GemmTnArgs& a, int tile, int s) {
  const float* __restrict__ U = a.U; const float* __restrict__ X = a.X;
  const int64_t n = a.n, rows_per_slice = a.rows_per_slice, rows_per_wave = a.rows_per_wave, nb = a.nb;
  const int O = a.O, K = a.K, tiles_k = a.tiles_k, S = a.S, NT_ = a.n_tiles;
  float* __restrict__ G = a.G; float* __restrict__ gb = a.gb; float* __restrict__ ws = a.ws; float* __restrict__ wsb = a.wsb;
  unsigned* __restrict__ tickets = a.tickets;
  __shared__ float red[TN_WAVES][32][33];
  __shared__ float redb[TN_WAVES][32];
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hi = lane >> 5, el = lane & 31;
  const int to = tile / tiles_k, tk = tile % tiles_k;
  const int o = 32 * to + el, k = 32 * tk + el;
  const bool o_ok = o < O, k_ok = k < K;
  const int64_t slice_end = ((s + 1) * rows_per_slice < n) ? (s + 1) * rows_per_slice : n;
  const int64_t r0 = s * rows_per_slice + wv * rows_per_wave;
  const int64_t r1 = (r0 + rows_per_wave < slice_end) ? r0 + rows_per_wave : slice_end;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float usum = 0.f;
  // The bias gradient (column sums of U) runs over the rows [0, nb) only.  The wave's row range is cut at nb and walked as two loops -- the
  // first adds to the column sums, the second does not: a per-element select inside one loop cost 30 registers and pushed the kernel into scratch.
  const int64_t rmid = nb < r0 ? r0 : (nb > r1 ? r1 : nb);
  // Addresses = wave-uniform base of the batch (scalar registers) + a 32-bit lane offset per row of the batch that does not change from batch
  // to batch (64-bit lane addresses: two registers per load in flight -- the double-buffered loop below went 940 bytes into scratch with them).
  // Two register sets: the rows of batch b + 1 are requested before the MFMAs of batch b run (a wave of the training step's pair-row problems
  // walks ~10 batches; one set = ten exposed round trips, 45 us for the batched launch of a step).
  unsigned uo[TN_BATCH], xo[TN_BATCH];
#pragma unroll
  for (int q = 0; q < TN_BATCH; ++q) {
    uo[q] = (unsigned)((2 * q + hi) * O + (o_ok ? o : 0));
    xo[q] = (unsigned)((2 * q + hi) * K + (k_ok ? k : 0));
  }
#define SPK_TN_LOAD(AV, BV, RB, REND)                                                       \
  {                                                                                         \
    const auto ub = spk_uniform_ptr(U + (RB) * O);                                          \
    const auto xb = spk_uniform_ptr(X + (RB) * K);                                          \
    _Pragma("unroll") for (int q = 0; q < TN_BATCH; ++q) {                                  \
      const bool ok = (RB) + 2 * q + hi < (REND);                                           \
      AV[q] = (ok && o_ok) ? ub[uo[q]] : 0.f;                                               \
      BV[q] = (ok && k_ok) ? xb[xo[q]] : 0.f;                                               \
    }                                                                                       \
  }
#define SPK_TN_MMA(AV, BV, WITH_BIAS)                                                       \
  _Pragma("unroll") for (int q = 0; q < TN_BATCH; ++q) {                                    \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[q], BV[q], acc, 0, 0, 0);                 \
    if (WITH_BIAS) usum += AV[q];                                                           \
  }
#define SPK_TN_LOOP(RBEG, REND, WITH_BIAS)                                                  \
  if ((RBEG) < (REND)) {                                                                    \
    float av0[TN_BATCH], bv0[TN_BATCH], av1[TN_BATCH], bv1[TN_BATCH];                       \
    SPK_TN_LOAD(av0, bv0, (RBEG), (REND))                                                   \
    for (int64_t rb = (RBEG); rb < (REND); rb += 4 * TN_BATCH) {                            \
      if (rb + 2 * TN_BATCH < (REND)) SPK_TN_LOAD(av1, bv1, rb + 2 * TN_BATCH, (REND))      \
      SPK_TN_MMA(av0, bv0, WITH_BIAS)                                                       \
      if (rb + 4 * TN_BATCH < (REND)) SPK_TN_LOAD(av0, bv0, rb + 4 * TN_BATCH, (REND))      \
      if (rb + 2 * TN_BATCH < (REND)) { SPK_TN_MMA(av1, bv1, WITH_BIAS) }                   \
    }                                                                                       \
  }
  SPK_TN_LOOP(r0, rmid, true)
  SPK_TN_LOOP(rmid, r1, false)
#undef SPK_TN_LOAD
#undef SPK_TN_MMA
#undef SPK_TN_LOOP
  usum += __shfl_xor(usum, 32, 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wv][(r & 3) + 8 * (r >> 2) + 4 * hi][el] = acc[r];
  if (hi == 0) redb[wv][el] = usum;
  __syncthreads();
  // thread t owns outputs (row t / 32 + 16 h, column t % 32), h = 0, 1
  const int orow = tid >> 5, ocol = tid & 31;
  float v0 = 0.f, v1 = 0.f, vb = 0.f;
#pragma unroll
  for (int w = 0; w < TN_WAVES; ++w) {
    v0 += red[w][orow][ocol];
    v1 += red[w][orow + 16][ocol];
  }
  if (tid < 32)
#pragma unroll
    for (int w = 0; w < TN_WAVES; ++w) vb += redb[w][tid];
  const int go0 = 32 * to + orow, go1 = go0 + 16, gk = 32 * tk + ocol;
  const bool want_b = gb != nullptr && tk == 0 && tid < 32;
  if (S > 1) {
    float* wt = ws + ((int64_t)s * NT_ + tile) * 1024;
    wt[tid] = v0;
    wt[tid + 512] = v1;
    if (want_b) wsb[((int64_t)s * NT_ + tile) * 32 + tid] = vb;
    if (a.defer) return;
    __threadfence();
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(&tickets[tile], 1u);
    __syncthreads();
    if (s_ticket != (unsigned)(S - 1)) return;
    __threadfence();
    if (tid == 0) tickets[tile] = 0u;
    v0 = 0.f; v1 = 0.f; vb = 0.f;
    for (int q = 0; q < S; ++q) {
      const float* wq = ws + ((int64_t)q * NT_ + tile) * 1024;
      v0 += wq[tid];
      v1 += wq[tid + 512];
      if (want_b) vb += wsb[((int64_t)q * NT_ + tile) * 32 + tid];
    }
  }
  if (gk < K) {
    if (go0 < O) G[(int64_t)go0 * K + gk] = v0;
    if (go1 < O) G[(int64_t)go1 * K + gk] = v1;
  }
  if (want_b && 32 * to + tid < O) gb[32 * to + tid] = vb;
}

// second launch of a deferred problem: G tile = sum of the S partial tiles in slice order (deterministic); 64 * TN_WAVES threads
__device__ __forceinline__ void gemm_tn_reduce_block(const GemmTnArgs& a, int tile) {
  const int tid = threadIdx.x;
  const int O = a.O, K = a.K, S = a.S, NT_ = a.n_tiles;
  const int to = tile / a.tiles_k, tk = tile % a.tiles_k;
  const int orow = tid >> 5, ocol = tid & 31;
  const int go0 = 32 * to + orow, go1 = go0 + 16, gk = 32 * tk + ocol;
  const bool want_b = a.gb != nullptr && tk == 0 && tid < 32;
  float v0 = 0.f, v1 = 0.f, vb = 0.f;
  for (int q = 0; q < S; ++q) {
    const float* wq = a.ws + ((int64_t)q * NT_ + tile) * 1024;
    v0 += wq[tid];
    v1 += wq[tid + 512];
    if (want_b) vb += a.wsb[((int64_t)q * NT_ + tile) * 32 + tid];
  }
  if (gk < K) {
    if (go0 < O) a.G[(int64_t)go0 * K + gk] = v0;
    if (go1 < O) a.G[(int64_t)go1 * K + gk] = v1;
  }
  if (want_b && 32 * to + tid < O) a.gb[32 * to + tid] = vb;
}

static inline GemmTnArgs spk_gemm_tn_args(const float* U, const float* X, int64_t n, int O, int K, int S, int tiles, float* G, float* gb, float* ws,
                                          uint32_t* tickets) {
  int64_t rpw = (n + (int64_t)S * TN_WAVES - 1) / ((int64_t)S * TN_WAVES);
  rpw += rpw & 1;                                                       // whole MFMA steps per wave
  if (rpw < 2) rpw = 2;
  GemmTnArgs a;
  a.U = U; a.X = X; a.n = n; a.nb = n; a.O = O; a.K = K; a.tiles_k = (K + 31) / 32; a.S = S; a.n_tiles = tiles;
  a.rows_per_slice = rpw * TN_WAVES; a.rows_per_wave = rpw;
  a.defer = 0;
  a.G = G; a.gb = gb; a.ws = ws; a.wsb = ws ? ws + (int64_t)S * tiles * 1024 : nullptr; a.tickets = (unsigned*)tickets;
  return a;
}
