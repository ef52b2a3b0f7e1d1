// PaiNN message (representation/painn.py:31-67) for LARGE lists (periodic boxes, configs[4] of BASELINE.json): BLOCK kernels --
// the "unique neighbours of a block of centre atoms staged in LDS, one 16-channel slice at a time, filter on the matrix core"
// design of HISTORY.md 4.3 / the round-3 review.  EXPERIMENT, OPT-IN (spk_painn_set_block(1) + a plan from spk_blocks_build;
// SPK_BLOCKS=1 for the torch operators): parity-green (tests/test_gpu_painn_blk.py), bit-reproducible (no atomics), and on the
// 32k-atom water box SLOWER than the row / tile kernels it was meant to replace -- forward 0.97 ms against 0.67 ms, backward
// 2.8 ms (passes T + G) against 1.46 ms, force call 11.3 ms against 7.26 ms (profiles/r04_block_kernels.md).  What was measured:
//
//   * what bounds the row / tile kernels (profiles/r04_msg_pmc.txt, r04_atom_order_experiment.json): the row kernels issue 143
//     (forward) / 330 (backward) VALU instructions per edge and wavefront, 60 % of them the filter recomputation, VALU 52 % / 66 %
//     busy at 2 waves per SIMD; the MFMA tile forward issues 55 but waits on dependent 4-byte gathers (VALU 24 % busy, L2 hit rate
//     33 %).  The ORDER of the atoms (lattice, cell-sorted, Morton, random) moves these times by 3-6 % (24 % for the tile forward at
//     random order): they are bound by instruction issue and latency, not by bytes -- staging for REUSE alone cannot help them;
//   * the block kernels do cut the instruction count (MFMA for the filter: 15 v_mfma_f32_16x16x4_f32 + ~130 VALU per 16 edges x 16
//     channels, MFMA pipe 21 % busy) and the gathered bytes (unique neighbours: 2.9 x fewer rows at 8 atoms per block), but the
//     staging itself is the bottleneck: 64 KB of 64-byte row pieces per block and slice arrive at ~10 B/clk per CU whichever way
//     they are requested -- register-staged wide loads (6 dependent round trips of 1.3-8 k cycles per workgroup before round-trip
//     batching, 3 after), asynchronous global_load_lds_dwordx4 from the compute waves (a wave's loads retire in order, so its next
//     tile waits for the whole next slice), or from two dedicated loader waves (the burst of 1 150 scattered 64-byte segments
//     occupies the CU's address path for ~7 k cycles, during which the compute waves' own tile loads do not issue) -- against
//     ~7 k cycles of tile work per slice.  Two buffers of 64 KB leave room for one workgroup per CU, so nothing else hides it.
//     Cycle stamps of every variant: profiles/r04_block_kernels.md.
//
// Structure (as measured last): a workgroup = one block of <= 8 consecutive centre atoms (one wavefront each) + two loader
// wavefronts, ALL 16-channel slices one after the other; the rows of the block's UNIQUE neighbours (ascending list + per-edge local
// index from the plan) are staged per slice into one of two LDS buffers by asynchronous loads while the previous slice is computed;
// the filter Phi f_c = A W^T runs on v_mfma_f32_16x16x4_f32 with a tile = 16 directed edges of ONE centre atom x 16 channels; A and
// A' = dA/dd come from a per-call "prep" launch that evaluates the radial basis and the cutoff once per edge and call in full
// precision and stores them in the lane order of the MFMA A operand; a lane owns (channel, 4 edges), a wave owns its atom, so the
// per-atom sums stay in registers and are stored once.  The backward is two passes over the same machinery: T (transposed sums ->
// gc, gmu; symmetric lists) and G (geometry gradient; per-edge sums over the slice's 16 channels by DPP row reductions, partials
// [slice][E][4] reduced in a fixed order by a finalize launch); the eval path's first-interaction backward is pass G alone.
// Blocks whose unique neighbours exceed the capacity are split by the plan (8 -> 4 -> 2 -> 1 atoms); lists on which single
// atoms do not fit are refused by the plan and keep the row kernels.
#include "spk_painn_msg.h"
#include "spk_painn_blk.h"

namespace {

constexpr int BA = SPK_BLK_ATOMS;   // atoms per group
constexpr int SL = 16;              // channels per slice
constexpr int NREC = 6;             // float4 record quantities per (tile, 16-lane row): jl, ux, uy, uz, fc, dfc
constexpr int SORT_MAX = 2048;

#define BLK_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

// ------------------------------------------------------------------------------------------------------------------ plan
__device__ __forceinline__ void blk_bitonic_sort(int* key, int P, int tid, int nthr) {
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += nthr) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const int a = key[i], b = key[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { key[i] = b; key[ixj] = a; }
        }
      }
      __syncthreads();
    }
}

// one workgroup (256 threads) per group of BA atoms: unique neighbours of the group, or of its halves / quarters / ... when they
// do not fit `cap`; local index of every edge's neighbour; number of 16-edge tiles (tiles are aligned to ATOMS)
__global__ __launch_bounds__(256) void k_blk_plan(const int64_t* __restrict__ idx_j, const int32_t* __restrict__ rowptr, int N, int cap,
                                                  int32_t* __restrict__ sub_n, int32_t* __restrict__ sub_u, int32_t* __restrict__ uniq,
                                                  uint16_t* __restrict__ jl, int32_t* __restrict__ stats) {
  __shared__ int sKey[SORT_MAX];
  __shared__ int sUq[SORT_MAX];
  __shared__ int sScan[256];
  __shared__ int sOk;
  const int tid = threadIdx.x;
  const int g = blockIdx.x;
  const int a0 = g * BA, a1 = min(a0 + BA, N);
  for (int w = BA; w >= 1; w >>= 1) {
    const int nsub = BA / w;
    if (tid == 0) sOk = 1;
    __syncthreads();
    int maxu = 0;
    for (int s = 0; s < nsub; ++s) {
      const int lo = min(a0 + s * w, a1), hi = min(lo + w, a1);
      const int e0 = rowptr[lo], n = rowptr[hi] - e0;
      if (n > SORT_MAX) { if (tid == 0) sOk = 0; break; }                   // (uniform: n is the same for every thread)
      int P = 2;
      while (P < n) P <<= 1;
      for (int i = tid; i < P; i += 256) sKey[i] = i < n ? (int)idx_j[e0 + i] : 0x7fffffff;
      __syncthreads();
      blk_bitonic_sort(sKey, P, tid, 256);
      // ordered compaction of the first occurrences: 8 consecutive keys per thread, block scan over the thread totals
      const int per = P / 256 > 0 ? P / 256 : 1;
      int cnt = 0;
      for (int q = 0; q < per; ++q) {
        const int i = tid * per + q;
        if (i < n && (i == 0 || sKey[i] != sKey[i - 1])) ++cnt;
      }
      sScan[tid] = cnt;
      __syncthreads();
      for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? sScan[tid - off] : 0;
        __syncthreads();
        sScan[tid] += v;
        __syncthreads();
      }
      const int U = sScan[255];
      int pos = sScan[tid] - cnt;
      for (int q = 0; q < per; ++q) {
        const int i = tid * per + q;
        if (i < n && (i == 0 || sKey[i] != sKey[i - 1])) sUq[pos++] = sKey[i];
      }
      __syncthreads();
      if (U > cap) { if (tid == 0) sOk = 0; break; }                        // (uniform)
      maxu = max(maxu, U);
      for (int i = tid; i < U; i += 256) uniq[e0 + i] = sUq[i];
      if (tid == 0) sub_u[g * BA + s] = U;
      for (int i = tid; i < n; i += 256) {
        const int key = (int)idx_j[e0 + i];
        int lo_ = 0, hi_ = U;                                               // lower bound in the ascending unique keys
        while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (sUq[mid] < key) lo_ = mid + 1; else hi_ = mid; }
        jl[e0 + i] = (uint16_t)lo_;
      }
      __syncthreads();
    }
    __syncthreads();
    if (sOk) {
      if (tid == 0) { sub_n[g] = nsub; atomicMax(&stats[0], maxu); }
      return;
    }
    __syncthreads();
  }
  if (tid == 0) { sub_n[g] = 0; atomicOr(&stats[1], 1); }
}

// atom_tile0 = exclusive scan of ceil(degree / 16) (single workgroup: runs once per list), stats[2] = number of tiles
__global__ __launch_bounds__(1024) void k_blk_tile_scan(const int32_t* __restrict__ rowptr, int N, int32_t* __restrict__ atom_tile0, int32_t* __restrict__ stats) {
  __shared__ int sS[1024];
  __shared__ int sBase;
  const int tid = threadIdx.x;
  if (tid == 0) sBase = 0;
  __syncthreads();
  for (int c0 = 0; c0 < N; c0 += 1024) {
    const int a = c0 + tid;
    const int v = a < N ? (rowptr[a + 1] - rowptr[a] + 15) / 16 : 0;
    sS[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = tid >= off ? sS[tid - off] : 0;
      __syncthreads();
      sS[tid] += t;
      __syncthreads();
    }
    const int base = sBase;
    if (a < N) atom_tile0[a] = base + sS[tid] - v;
    __syncthreads();
    if (tid == 1023) sBase = base + sS[1023];
    __syncthreads();
  }
  if (tid == 0) { atom_tile0[N] = sBase; stats[2] = sBase; }
}

// blk0 = exclusive scan of sub_n (single workgroup), stats[3] = number of blocks
__global__ __launch_bounds__(1024) void k_blk_block_scan(const int32_t* __restrict__ sub_n, int ng, int32_t* __restrict__ blk0, int32_t* __restrict__ stats) {
  __shared__ int sS[1024];
  __shared__ int sBase;
  const int tid = threadIdx.x;
  if (tid == 0) sBase = 0;
  __syncthreads();
  for (int c0 = 0; c0 < ng; c0 += 1024) {
    const int g = c0 + tid;
    const int v = g < ng ? sub_n[g] : 0;
    sS[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = tid >= off ? sS[tid - off] : 0;
      __syncthreads();
      sS[tid] += t;
      __syncthreads();
    }
    const int base = sBase;
    if (g < ng) blk0[g] = base + sS[tid] - v;
    __syncthreads();
    if (tid == 1023) sBase = base + sS[1023];
    __syncthreads();
  }
  if (tid == 0) stats[3] = sBase;
}
__global__ void k_blk_desc_fill(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_u, const int32_t* __restrict__ blk0,
                                const int32_t* __restrict__ rowptr, int ng, int N, int32_t* __restrict__ desc) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const int nsub = sub_n[g];
  if (nsub <= 0) return;
  const int w = BA / nsub, a0 = g * BA, a1 = min(a0 + BA, N);
  for (int s = 0; s < nsub; ++s) {
    const int lo = min(a0 + s * w, a1), hi = min(lo + w, a1);
    int32_t* d = desc + 4 * (int64_t)(blk0[g] + s);
    d[0] = lo; d[1] = hi - lo; d[2] = rowptr[lo]; d[3] = sub_u[g * BA + s];
  }
}

__global__ void k_blk_tile_fill(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ atom_tile0, int N, int32_t* __restrict__ tile_info) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= N) return;
  const int e0 = rowptr[a], e1 = rowptr[a + 1];
  int t = atom_tile0[a];
  for (int e = e0; e < e1; e += 16, ++t) { tile_info[2 * t] = e; tile_info[2 * t + 1] = min(16, e1 - e); }
}

// ------------------------------------------------------------------------------------------------------------------ prep
// One wavefront per tile: A / A' in the lane order of the MFMA A operand (lane = 16 h + row: A[row][4 u + h]), the per-edge record
// quantities as float4 over the 4 rows of a 16-lane row group.  Padded rows (beyond the atom's last edge) get A = A' = 0, f_c = 0.
template <int KS>
__global__ __launch_bounds__(256) void k_blk_prep(const float* __restrict__ rij, const int32_t* __restrict__ tile_info, const uint16_t* __restrict__ jl,
                                                  int n_tiles, RadialDev rb, float* __restrict__ apack, float* __restrict__ adpack, float* __restrict__ rec) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wv;
  if (t >= n_tiles) return;
  const int e0 = tile_info[2 * t], ne = tile_info[2 * t + 1];
  const int row = lane & 15, h = lane >> 4;
  const bool valid = row < ne;
  const int64_t e = e0 + (valid ? row : 0);
  const float rx = rij[3 * e], ry = rij[3 * e + 1], rz = rij[3 * e + 2];
  const float d = sqrtf(rx * rx + ry * ry + rz * rz);
  const float inv = 1.0f / d;
  float fc, dfc;
  spk_cutoff_eval(rb.cutoff, d, fc, dfc);
  if (!valid) { fc = 0.f; dfc = 0.f; }
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    float p, dp;
    spk_rbf_eval(rb, 4 * u + h, d, p, dp);
    apack[((int64_t)t * KS + u) * 64 + lane] = fc * p;
    adpack[((int64_t)t * KS + u) * 64 + lane] = fc * dp + dfc * p;
  }
  if (h == 0) {
    // rec[((t * NREC + q) * 4 + row / 4) * 4 + row % 4]
    float* r0 = rec + (int64_t)t * NREC * 16 + (row >> 2) * 4 + (row & 3);
    r0[0 * 16] = __int_as_float(valid ? (int)jl[e] : 0);
    r0[1 * 16] = valid ? rx * inv : 0.f;
    r0[2 * 16] = valid ? ry * inv : 0.f;
    r0[3 * 16] = valid ? rz * inv : 0.f;
    r0[4 * 16] = fc;
    r0[5 * 16] = dfc;
  }
}

// ------------------------------------------------------------------------------------------------------------------ kernels
#define BLK_STAMP(n) do { if (a.dbg && blockIdx.x == a.dbg_block && threadIdx.x == 0) a.dbg[n] = (long long)__builtin_readcyclecounter(); } while (0)
struct BlkArgs {
  MsgArgs m;
  spk_blocks_t b;
  long long* dbg;    // tuning aid: cycle stamps of thread 0 of workgroup dbg_block (spk_painn_blk_set_debug_buffer; null in production)
  int dbg_block;
  int nsl;           // F / 16 slices
  int bpx;           // blocks per XCD (blockIdx swizzle)
  int stg;           // 16-byte staging units per thread and slice = ceil(max_unique * NP * 4 / NTHR)
};

__device__ __forceinline__ float blk_row16_sum(float v) {     // every lane: the sum over its row of 16 lanes (DPP, no LDS)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}
__device__ __forceinline__ float blk_rows_sum(float v) {      // sum over the four 16-lane rows (same column), every lane
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

constexpr int NCOMP = 64 * BA;     // compute threads: one wavefront per centre atom of a block
constexpr int NLOAD = 128;         // + two LOADER wavefronts that issue the asynchronous global -> LDS loads of the next slice: a wave's loads
                                   // complete in order, so a compute wave that issued them would wait for the whole next slice before its own
                                   // next tile (measured: 10-13 k cycles per slice instead of ~5 k)
constexpr int NTHR = NCOMP + NLOAD;
constexpr int STGMAX = SPK_BLK_STG_MAX;   // 16-byte units per COMPUTE-thread-equivalent and slice: two LDS buffers of STGMAX * NCOMP * 16 B = 144 KB
constexpr int LSTG = STGMAX * NCOMP / NLOAD;   // units per loader thread and slice

typedef __attribute__((address_space(3))) void* blk_lds_ptr;
typedef const __attribute__((address_space(1))) void* blk_glb_ptr;
// asynchronous 16 bytes per lane, global -> LDS without registers: the LDS destination is wave-uniform base + 16 lane
__device__ __forceinline__ void blk_glds16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((blk_glb_ptr)g, (blk_lds_ptr)lds_wave_base, 16, 0, 0);
}

// blockIdx -> block: workgroups go round-robin over the 8 XCDs; XCD x takes the blocks [x bpx, (x + 1) bpx), so neighbouring
// blocks (whose halos overlap) share an L2
__device__ __forceinline__ int blk_decode(const BlkArgs& a) {
  const int w = blockIdx.x;
  const int j = w >> 3;
  const int b = (w & 7) * a.bpx + j;
  return (j < a.bpx && b < a.b.n_blocks) ? b : -1;
}

// per-tile inputs of a lane: A operand values (and A'), record quantities of the lane's 4 rows.  pa / pd / pr point at the
// lane's entries of tile 0 of the atom; the tiles of an atom are consecutive.
template <int KS, bool DER>
struct BlkTileIn {
  float A[KS];
  float Ad[DER ? KS : 1];
  f32x4 rj, rux, ruy, ruz, rfc, rdfc;
  __device__ __forceinline__ void load(const float* __restrict__ pa, const float* __restrict__ pd, const f32x4* __restrict__ pr, int k) {
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      A[u] = pa[(k * KS + u) * 64];
      if (DER) Ad[u] = pd[(k * KS + u) * 64];
    }
    const f32x4* rp = pr + k * (NREC * 4);
    rj = rp[0]; rux = rp[4]; ruy = rp[8]; ruz = rp[12]; rfc = rp[16];
    if (DER) rdfc = rp[20];
  }
};

// filter weights of the lane's channel in the MFMA B-operand order (k = 4 u + h), bias of the three parts
template <int KS>
struct BlkWeights {
  float w[3][KS], bias[3];
  __device__ __forceinline__ void load(const MsgArgs& m, int ch, int h) {
    const int F = m.F, K = m.rb.n_rbf;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      bias[p] = m.bf[p * F + ch];
#pragma unroll
      for (int u = 0; u < KS; ++u) { const int k = min(4 * u + h, K - 1); const float v = m.wf[(int64_t)(p * F + ch) * K + k]; w[p][u] = 4 * u + h < K ? v : 0.f; }
    }
  }
};

// One workgroup = one block of <= BA consecutive centre atoms, ALL 16-channel slices one after the other; one wavefront per centre
// atom.  The rows of the block's unique neighbours are staged per slice by asynchronous global -> LDS loads into one of two buffers:
// slice s + 1 lands while slice s is computed (the addresses need the neighbours' atom indices only, which are loaded once per
// block), and so do the next slice's filter weights and centre values.  Measured before this structure (one slice per workgroup,
// register staging): 6 dependent memory round trips of 1.3-8 k cycles in front of 9 k cycles of tile work.
// MODE 0: forward (stages c | mu of the neighbours; MU0: c only)    -> q_out, mu_out
// MODE 1: backward pass T (stages gq | gmu of the neighbours)        -> gc, gmu
template <int KS, int MODE, bool MU0>
__global__ __launch_bounds__(NTHR) void k_painn_blk_sum(BlkArgs a) {
  constexpr int NP = MODE == 0 ? (MU0 ? 3 : 6) : 4;            // 64-byte row pieces per staged neighbour
  extern __shared__ __attribute__((aligned(16))) float sBuf[];   // 2 x [stg * NTHR * 4] floats, rows as [U][NP][16]
  const int b = blk_decode(a);
  if (b < 0) return;
  BLK_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int el = lane & 15, h = lane >> 4;
  const int F = a.m.F;
  const int64_t F3 = 3 * (int64_t)F;
  const int lo = a.b.blk_desc[4 * b], na = a.b.blk_desc[4 * b + 1], e0 = a.b.blk_desc[4 * b + 2], U = a.b.blk_desc[4 * b + 3];
  if (na <= 0) return;
  const int total = U * NP * 4;
  const int bufsz = a.stg * NCOMP * 4;   // floats per buffer
  if (wv >= BA) {
    // ================= loader wavefronts: the atom index behind each of this thread's staging units once per block, then per slice
    // one batch of asynchronous 16-byte loads into the buffer the compute waves are NOT reading
    const int lt = tid - NCOMP, lw = wv - BA;
    const int lstg = (total + NLOAD - 1) / NLOAD;
    int idx[LSTG];
#pragma unroll
    for (int k = 0; k < LSTG; ++k) {
      const int i = min(k * NLOAD + lt, total > 0 ? total - 1 : 0);
      idx[k] = (k < lstg) ? a.b.uniq[e0 + i / (NP * 4)] : 0;
    }
    auto stage = [&](int slice, float* buf) {
#pragma unroll
      for (int k = 0; k < LSTG; ++k)
        if (k < lstg) {
          const int i = min(k * NLOAD + lt, total - 1);
          const int rem = i % (NP * 4), pc = rem >> 2, q4 = rem & 3;
          const int64_t an = idx[k];
          const float* src;
          if (MODE == 0) src = (pc < 3 ? a.m.c + an * F3 + pc * F : a.m.mu + an * F3 + (pc - 3) * F) + slice * SL + 4 * q4;
          else src = (pc == 0 ? a.m.gq_out + an * F : a.m.gmu_out + an * F3 + (pc - 1) * F) + slice * SL + 4 * q4;
          blk_glds16(src, buf + (k * NLOAD + lw * 64) * 4);
        }
    };
    if (total > 0) stage(0, sBuf);
    for (int slice = 0; slice < a.nsl; ++slice) {
      __syncthreads();
      if (slice + 1 < a.nsl && total > 0) stage(slice + 1, sBuf + ((slice + 1) & 1) * bufsz);
    }
    return;
  }
  // ================= compute wavefronts: this wave's centre atom and its tiles
  const bool has = wv < na;
  const int atom = lo + (has ? wv : 0);
  const int t0 = a.b.atom_tile0[atom];
  const int n = has ? a.b.atom_tile0[atom + 1] - t0 : 0;
  const float* pa = a.b.apack + (int64_t)t0 * KS * 64 + lane;
  const f32x4* pr = (const f32x4*)(a.b.rec + (int64_t)t0 * NREC * 16) + h;
  auto load_centre = [&](float (&cv)[7], int ch) {
    const int64_t o1 = (int64_t)atom * F + ch, o3 = (int64_t)atom * F3 + ch;
    if (MODE == 0) {
      cv[0] = a.m.q[o1]; cv[1] = a.m.mu[o3]; cv[2] = a.m.mu[o3 + F]; cv[3] = a.m.mu[o3 + 2 * F]; cv[4] = 0.f; cv[5] = 0.f; cv[6] = 0.f;
    } else {
      cv[0] = a.m.mu[o3]; cv[1] = a.m.mu[o3 + F]; cv[2] = a.m.mu[o3 + 2 * F]; cv[3] = a.m.c[o3 + 2 * F];
      cv[4] = a.m.gmu_out[o3]; cv[5] = a.m.gmu_out[o3 + F]; cv[6] = a.m.gmu_out[o3 + 2 * F];
    }
  };
  BlkWeights<KS> W;
  float cen[7];
  W.load(a.m, el, h);
  load_centre(cen, el);
  BlkTileIn<KS, false> bufA, bufB;
  bufA.load(pa, nullptr, pr, 0);                 // (unconditional: a wave without tiles reads a valid tile)
  BLK_STAMP(1);
  for (int slice = 0; slice < a.nsl; ++slice) {
    float* sRow = sBuf + (slice & 1) * bufsz;
    __syncthreads();                             // rows of this slice have landed (the barrier drains vmcnt); every wave is done with the other buffer
    BLK_STAMP(2 + 3 * slice);
    const bool more = slice + 1 < a.nsl;
    BlkWeights<KS> Wn;
    float cn[7];
    if (more) { Wn.load(a.m, (slice + 1) * SL + el, h); load_centre(cn, (slice + 1) * SL + el); }
    const int ch = slice * SL + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    auto process = [&](const BlkTileIn<KS, false>& t) {
      f32x4 acc[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) acc[p] = t.rfc * W.bias[p];
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        acc[0] = BLK_MFMA(t.A[u], W.w[0][u], acc[0]);
        acc[1] = BLK_MFMA(t.A[u], W.w[1][u], acc[1]);
        if (!(MODE == 0 && MU0)) acc[2] = BLK_MFMA(t.A[u], W.w[2][u], acc[2]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = __float_as_int(t.rj[r]);
        const float* row = sRow + j * (NP * 16) + el;
        const float ux = t.rux[r], uy = t.ruy[r], uz = t.ruz[r];
        if (MODE == 0) {
          const float mq = acc[0][r] * row[0];
          const float mR = acc[1][r] * row[16];
          s0 += mq;
          s1 = fmaf(mR, ux, s1); s2 = fmaf(mR, uy, s2); s3 = fmaf(mR, uz, s3);
          if (!MU0) {
            const float mm = acc[2][r] * row[32];
            s1 = fmaf(mm, row[48], s1); s2 = fmaf(mm, row[64], s2); s3 = fmaf(mm, row[80], s3);
          }
        } else {
          const float gb0 = row[16], gb1 = row[32], gb2 = row[48];
          s0 = fmaf(acc[0][r], row[0], s0);                                     // gc_q  += F_q gq_j
          s4 = fmaf(-acc[1][r], gb0 * ux + gb1 * uy + gb2 * uz, s4);          // gc_R  -= F_R (gmu_j . u)
          s1 = fmaf(acc[2][r], gb0, s1); s2 = fmaf(acc[2][r], gb1, s2); s3 = fmaf(acc[2][r], gb2, s3);   // S += F_mu gmu_j
        }
      }
    };
    // two tile buffers, one tile ahead, no register rotation (a copy of a buffer would wait for its loads)
#define BLK_TSTAMP(i_) do { if (a.dbg && slice == 1 && blockIdx.x == a.dbg_block && threadIdx.x == 0) a.dbg[i_] = (long long)__builtin_readcyclecounter(); } while (0)
#define BLK_TWAIT() do { if (a.dbg && slice == 1 && blockIdx.x == a.dbg_block) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
    for (int k = 0; k < n; k += 2) {
      bufB.load(pa, nullptr, pr, k + 1 < n ? k + 1 : k);
      BLK_TSTAMP(32 + 3 * k);
      if (a.dbg && slice == 1 && blockIdx.x == a.dbg_block) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      BLK_TSTAMP(33 + 3 * k);
      process(bufA);
      BLK_TSTAMP(34 + 3 * k);
      if (k + 1 >= n) { if (more) bufA.load(pa, nullptr, pr, 0); break; }
      bufA.load(pa, nullptr, pr, k + 2 < n ? k + 2 : 0);          // past the last tile: tile 0 again, for the next slice
      BLK_TSTAMP(35 + 3 * k);
      if (a.dbg && slice == 1 && blockIdx.x == a.dbg_block) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      BLK_TSTAMP(36 + 3 * k);
      process(bufB);
      BLK_TSTAMP(37 + 3 * k);
    }
    BLK_STAMP(3 + 3 * slice);
    // ---- per-atom epilogue: reduce over the four 16-lane rows, add the residual / form gc, gmu, store 64 bytes per output piece
    s0 = blk_rows_sum(s0); s1 = blk_rows_sum(s1); s2 = blk_rows_sum(s2); s3 = blk_rows_sum(s3);
    if (MODE == 1) s4 = blk_rows_sum(s4);
    if (has && h == 0) {
      const int64_t o1 = (int64_t)atom * F + ch, o3 = (int64_t)atom * F3 + ch;
      if (MODE == 0) {
        a.m.q_out[o1] = cen[0] + s0;
        a.m.mu_out[o3] = cen[1] + s1; a.m.mu_out[o3 + F] = cen[2] + s2; a.m.mu_out[o3 + 2 * F] = cen[3] + s3;
      } else {
        a.m.gc[o3] = s0; a.m.gc[o3 + F] = s4; a.m.gc[o3 + 2 * F] = cen[0] * s1 + cen[1] * s2 + cen[2] * s3;
        a.m.gmu[o3] = cen[4] + cen[3] * s1; a.m.gmu[o3 + F] = cen[5] + cen[3] * s2; a.m.gmu[o3 + 2 * F] = cen[6] + cen[3] * s3;
      }
    }
    if (more) {
      W = Wn;
#pragma unroll
      for (int v = 0; v < 7; ++v) cen[v] = cn[v];
    }
    BLK_STAMP(4 + 3 * slice);
  }
}

// backward pass G: per-edge sums over the 16 channels of the slice -> part[slice][e] = (dd, t_x, t_y, t_z)
template <int KS, bool MU0>
__global__ __launch_bounds__(NTHR) void k_painn_blk_geom(BlkArgs a) {
  constexpr int NP = MU0 ? 2 : 6;                                // staged: c_q, c_R (MU0) | c_q, c_R, c_mu, mu_x, mu_y, mu_z
  extern __shared__ __attribute__((aligned(16))) float sBuf[];
  const int b = blk_decode(a);
  if (b < 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int el = lane & 15, h = lane >> 4;
  const int F = a.m.F;
  const int64_t F3 = 3 * (int64_t)F;
  const int lo = a.b.blk_desc[4 * b], na = a.b.blk_desc[4 * b + 1], e0 = a.b.blk_desc[4 * b + 2], U = a.b.blk_desc[4 * b + 3];
  if (na <= 0) return;
  const int total = U * NP * 4;
  const int bufsz = a.stg * NCOMP * 4;
  if (wv >= BA) {      // loader wavefronts (see k_painn_blk_sum)
    const int lt = tid - NCOMP, lw = wv - BA;
    const int lstg = (total + NLOAD - 1) / NLOAD;
    int idx[LSTG];
#pragma unroll
    for (int k = 0; k < LSTG; ++k) {
      const int i = min(k * NLOAD + lt, total > 0 ? total - 1 : 0);
      idx[k] = (k < lstg) ? a.b.uniq[e0 + i / (NP * 4)] : 0;
    }
    auto stage = [&](int slice, float* buf) {
#pragma unroll
      for (int k = 0; k < LSTG; ++k)
        if (k < lstg) {
          const int i = min(k * NLOAD + lt, total - 1);
          const int rem = i % (NP * 4), pc = rem >> 2, q4 = rem & 3;
          const int64_t an = idx[k];
          const float* src = (pc < 3 ? a.m.c + an * F3 + pc * F : a.m.mu + an * F3 + (pc - 3) * F) + slice * SL + 4 * q4;
          blk_glds16(src, buf + (k * NLOAD + lw * 64) * 4);
        }
    };
    if (total > 0) stage(0, sBuf);
    for (int slice = 0; slice < a.nsl; ++slice) {
      __syncthreads();
      if (slice + 1 < a.nsl && total > 0) stage(slice + 1, sBuf + ((slice + 1) & 1) * bufsz);
    }
    return;
  }
  const bool has = wv < na;
  const int atom = lo + (has ? wv : 0);
  const int t0 = a.b.atom_tile0[atom];
  const int n = has ? a.b.atom_tile0[atom + 1] - t0 : 0;
  const int ea0 = a.m.rowptr[atom], ea1 = a.m.rowptr[atom + 1];
  const float* pa = a.b.apack + (int64_t)t0 * KS * 64 + lane;
  const float* pd = a.b.adpack + (int64_t)t0 * KS * 64 + lane;
  const f32x4* pr = (const f32x4*)(a.b.rec + (int64_t)t0 * NREC * 16) + h;
  auto load_centre = [&](float (&cv)[4], int ch) {          // gradients arriving at the wave's centre atom
    cv[0] = a.m.gq_out[(int64_t)atom * F + ch];
    cv[1] = a.m.gmu_out[(int64_t)atom * F3 + ch]; cv[2] = a.m.gmu_out[(int64_t)atom * F3 + F + ch]; cv[3] = a.m.gmu_out[(int64_t)atom * F3 + 2 * F + ch];
  };
  BlkWeights<KS> W;
  float cen[4];
  W.load(a.m, el, h);
  load_centre(cen, el);
  BlkTileIn<KS, true> bufA, bufB;
  bufA.load(pa, pd, pr, 0);
  for (int slice = 0; slice < a.nsl; ++slice) {
    float* sRow = sBuf + (slice & 1) * bufsz;
    f32x4* part = (f32x4*)a.b.part + (int64_t)slice * a.m.E;
    __syncthreads();
    const bool more = slice + 1 < a.nsl;
    BlkWeights<KS> Wn;
    float cn[4];
    if (more) { Wn.load(a.m, (slice + 1) * SL + el, h); load_centre(cn, (slice + 1) * SL + el); }
    const float gqa = cen[0], ga0 = cen[1], ga1 = cen[2], ga2 = cen[3];
    auto process = [&](const BlkTileIn<KS, true>& t, int k) {
      f32x4 FR = t.rfc * W.bias[1], dFq = t.rdfc * W.bias[0], dFR = t.rdfc * W.bias[1], dFm = t.rdfc * W.bias[2];
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        FR = BLK_MFMA(t.A[u], W.w[1][u], FR);
        dFq = BLK_MFMA(t.Ad[u], W.w[0][u], dFq);
        dFR = BLK_MFMA(t.Ad[u], W.w[1][u], dFR);
        if (!MU0) dFm = BLK_MFMA(t.Ad[u], W.w[2][u], dFm);
      }
      const int e_first = ea0 + 16 * k;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = __float_as_int(t.rj[r]);
        const float* row = sRow + j * (NP * 16) + el;
        const float ux = t.rux[r], uy = t.ruy[r], uz = t.ruz[r];
        const float cq = row[0], cR = row[16];
        const float gu = ga0 * ux + ga1 * uy + ga2 * uz;
        float dd = cq * gqa * dFq[r] + cR * gu * dFR[r];
        if (!MU0) {
          const float gm = ga0 * row[48] + ga1 * row[64] + ga2 * row[80];
          dd = fmaf(row[32] * gm, dFm[r], dd);
        }
        const float mR = FR[r] * cR;
        float tx = ga0 * mR, ty = ga1 * mR, tz = ga2 * mR;
        dd = blk_row16_sum(dd); tx = blk_row16_sum(tx); ty = blk_row16_sum(ty); tz = blk_row16_sum(tz);
        const int e = e_first + 4 * h + r;
        if (el == 0 && e < ea1) part[e] = f32x4{dd, tx, ty, tz};
      }
    };
    for (int k = 0; k < n; k += 2) {
      bufB.load(pa, pd, pr, k + 1 < n ? k + 1 : k);
      process(bufA, k);
      if (k + 1 >= n) { if (more) bufA.load(pa, pd, pr, 0); break; }
      bufA.load(pa, pd, pr, k + 2 < n ? k + 2 : 0);
      process(bufB, k + 1);
    }
    if (more) {
      W = Wn;
#pragma unroll
      for (int v = 0; v < 4; ++v) cen[v] = cn[v];
    }
  }
}

// gr[e] += dd u + (t - (t . u) u) / d with (dd, t) summed over the slices (fixed order)
__global__ void k_painn_blk_geom_final(const float* __restrict__ part, const float* __restrict__ rij, int64_t E, int nsl, float* __restrict__ gr) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const f32x4* p = (const f32x4*)part + e;
  f32x4 s = p[0];
  for (int k = 1; k < nsl; ++k) s += p[(int64_t)k * E];
  const float rx = rij[3 * e], ry = rij[3 * e + 1], rz = rij[3 * e + 2];
  const float d = sqrtf(rx * rx + ry * ry + rz * rz);
  if (!(d > 0.f)) return;
  const float inv = 1.0f / d;
  const float ux = rx * inv, uy = ry * inv, uz = rz * inv;
  const float dot = s.y * ux + s.z * uy + s.w * uz;
  gr[3 * e] += s.x * ux + (s.y - dot * ux) * inv;
  gr[3 * e + 1] += s.x * uy + (s.z - dot * uy) * inv;
  gr[3 * e + 2] += s.x * uz + (s.w - dot * uz) * inv;
}

int g_blk_mode = 0;   // 1: whenever a usable plan hangs on the graph; 0 / -1: never (the measured state of these kernels: slower than the row /
                      // tile kernels on the 32k-atom water box, see the file comment -- an opt-in experiment)
long long* g_blk_dbg = nullptr;
int g_blk_dbg_block = 0;

template <class KernT>
int blk_set_lds(KernT kern, size_t lds) {
  SPK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // per device; cheap, not a stream operation
  return SPK_OK;
}

void blk_fill_args(const MsgArgs& m, BlkArgs& a, int np) {
  a.m = m;
  a.b = *m.blocks;
  a.nsl = m.F / SL;
  a.bpx = (a.b.n_blocks + 7) / 8;
  a.stg = (a.b.max_unique * np * 4 + NCOMP - 1) / NCOMP;
  if (a.stg < 1) a.stg = 1;
  a.dbg = g_blk_dbg; a.dbg_block = g_blk_dbg_block;
}
int blk_grid(const BlkArgs& a) { return 8 * a.bpx; }
size_t blk_lds(const BlkArgs& a) { return (size_t)2 * a.stg * NCOMP * 16; }

}  // namespace

extern "C" void spk_painn_set_block(int mode) { g_blk_mode = mode; }
extern "C" int spk_blocks_group_atoms(void) { return BA; }
// tuning aid: device buffer (>= 32 int64) receiving the shader-clock stamps of thread 0 of workgroup `block` of the sum kernels; NULL = off
extern "C" void spk_painn_blk_set_debug_buffer(void* buf, int block) { g_blk_dbg = (long long*)buf; g_blk_dbg_block = block; }

bool spk_painn_blk_ok(const MsgArgs& a, bool bwd) {
  if (g_blk_mode < 0 || !a.blocks) return false;
  const spk_blocks_t& b = *a.blocks;
  if (!b.ok || b.n_groups <= 0 || b.n_blocks <= 0 || !b.blk_desc || !b.apack || !b.adpack || !b.rec || (bwd && !b.part)) return false;
  if (a.F % SL != 0 || a.F < SL || a.rb.n_rbf > 4 * b.ks || !(b.ks == 5 || b.ks == 8)) return false;
  if ((b.max_unique * 6 * 4 + NCOMP - 1) / NCOMP > STGMAX) return false;
  if (a.N * 3 * (int64_t)a.F >= (1LL << 31)) return false;
  return g_blk_mode > 0;
}

int spk_painn_blk_prep(const MsgArgs& a, hipStream_t stream) {
  const spk_blocks_t& b = *a.blocks;
  if (b.n_tiles <= 0) return SPK_OK;
  SpkProfScope prof("painn_blk_prep", stream);
  const int grid = (b.n_tiles + 3) / 4;
  if (b.ks == 5) hipLaunchKernelGGL(k_blk_prep<5>, dim3(grid), dim3(256), 0, stream, a.rij, b.tile_info, b.jl, b.n_tiles, a.rb, b.apack, b.adpack, b.rec);
  else hipLaunchKernelGGL(k_blk_prep<8>, dim3(grid), dim3(256), 0, stream, a.rij, b.tile_info, b.jl, b.n_tiles, a.rb, b.apack, b.adpack, b.rec);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

template <int KS, int MODE, bool MU0>
static int blk_launch_sum(const MsgArgs& m, hipStream_t stream) {
  constexpr int NP = MODE == 0 ? (MU0 ? 3 : 6) : 4;
  BlkArgs a;
  blk_fill_args(m, a, NP);
  auto kern = k_painn_blk_sum<KS, MODE, MU0>;
  SPK_TRY(blk_set_lds(kern, blk_lds(a)));
  hipLaunchKernelGGL(kern, dim3(blk_grid(a)), dim3(NTHR), blk_lds(a), stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
template <int KS, bool MU0>
static int blk_launch_geom(const MsgArgs& m, hipStream_t stream) {
  constexpr int NP = MU0 ? 2 : 6;
  BlkArgs a;
  blk_fill_args(m, a, NP);
  auto kern = k_painn_blk_geom<KS, MU0>;
  SPK_TRY(blk_set_lds(kern, blk_lds(a)));
  hipLaunchKernelGGL(kern, dim3(blk_grid(a)), dim3(NTHR), blk_lds(a), stream, a);
  SPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_painn_blk_geom_final, dim3((unsigned)((a.m.E + 255) / 256)), dim3(256), 0, stream, a.b.part, a.m.rij, a.m.E, a.nsl, a.m.gr);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

int spk_painn_blk_fwd(const MsgArgs& m, hipStream_t stream) {
  if (!m.blocks_prepared) SPK_TRY(spk_painn_blk_prep(m, stream));
  SpkProfScope prof(m.mu_zero ? "painn_msg_fwd_blk_mu0" : "painn_msg_fwd_blk", stream);
  if (m.blocks->ks == 5) return m.mu_zero ? blk_launch_sum<5, 0, true>(m, stream) : blk_launch_sum<5, 0, false>(m, stream);
  return m.mu_zero ? blk_launch_sum<8, 0, true>(m, stream) : blk_launch_sum<8, 0, false>(m, stream);
}

int spk_painn_blk_bwd(const MsgArgs& m, hipStream_t stream) {
  if (!m.blocks_prepared) SPK_TRY(spk_painn_blk_prep(m, stream));
  if (!m.geom_only) {
    SpkProfScope prof("painn_msg_bwd_blk_T", stream);
    if (m.blocks->ks == 5) SPK_TRY((blk_launch_sum<5, 1, false>(m, stream))); else SPK_TRY((blk_launch_sum<8, 1, false>(m, stream)));
  }
  if (m.E == 0) return SPK_OK;
  SpkProfScope prof(m.mu_zero ? "painn_msg_bwd_blk_G_mu0" : "painn_msg_bwd_blk_G", stream);
  if (m.blocks->ks == 5) return m.mu_zero ? blk_launch_geom<5, true>(m, stream) : blk_launch_geom<5, false>(m, stream);
  return m.mu_zero ? blk_launch_geom<8, true>(m, stream) : blk_launch_geom<8, false>(m, stream);
}

// ------------------------------------------------------------------------------------------------------------------ C ABI of the plan
extern "C" int spk_blocks_sizes(int64_t n_atoms, int64_t n_edges, int32_t n_rbf, int32_t n_atom_basis, int64_t* sizes) {
  SPK_CHECK_ARG(sizes && n_atoms >= 0 && n_edges >= 0 && n_rbf >= 1 && n_atom_basis >= SL, "spk_blocks_sizes: bad arguments");
  const int64_t ng = (n_atoms + BA - 1) / BA;
  const int64_t max_tiles = n_edges / 16 + n_atoms + 1;
  const int ks = n_rbf <= 20 ? 5 : 8;
  sizes[0] = ng;                      // sub_n      int32
  sizes[1] = ng * BA;                 // sub_u      int32
  sizes[2] = n_edges > 0 ? n_edges : 1;   // uniq   int32
  sizes[3] = n_edges > 0 ? n_edges : 1;   // jl     uint16
  sizes[4] = n_atoms + 1;             // atom_tile0 int32
  sizes[5] = 2 * max_tiles;           // tile_info  int32
  sizes[6] = max_tiles * ks * 64;     // apack, adpack (each) float -- the exact tile count is known after the build (host_stats[2])
  sizes[7] = max_tiles * NREC * 16;   // rec        float
  sizes[8] = (n_atom_basis / SL) * 4 * (n_edges > 0 ? n_edges : 1);   // part float
  sizes[9] = ks;
  sizes[10] = 4 * ng * BA + ng;       // blk_desc   int32: [<= ng BA][4], then ng ints of scratch
  return SPK_OK;
}

// Builds the plan into the caller's buffers (sizes from spk_blocks_sizes).  Synchronises the stream once (12-byte D2H).
// host_stats[0] = largest unique-neighbour count, [1] = 1 if the list does not fit (plan unusable), [2] = number of tiles.
extern "C" int spk_blocks_build(const spk_graph_t* g, int32_t n_rbf, int32_t cap, spk_blocks_t* out, int32_t* dev_stats, int32_t* host_stats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_blocks_build";
  SPK_CHECK_ARG(g && out && dev_stats && host_stats, "%s: null argument", who);
  SPK_CHECK_ARG(g->sorted && g->rowptr && g->idx_j, "%s: needs a list sorted by idx_i with its row pointers", who);
  SPK_CHECK_ARG(g->n_atoms < (1LL << 31) && g->n_edges < (1LL << 31), "%s: list too large", who);
  SPK_CHECK_ARG(out->sub_n && out->sub_u && out->uniq && out->jl && out->atom_tile0 && out->tile_info, "%s: null plan buffer", who);
  SPK_CHECK_ARG(out->blk_desc != nullptr, "%s: null plan buffer", who);
  if (cap <= 0) cap = SPK_BLK_CAP;
  SPK_CHECK_ARG(cap >= 1 && cap <= SPK_BLK_CAP, "%s: capacity %d out of range (1..%d)", who, cap, (int)SPK_BLK_CAP);
  const int N = (int)g->n_atoms;
  const int ng = (N + BA - 1) / BA;
  out->n_groups = ng; out->cap = cap; out->ks = n_rbf <= 20 ? 5 : 8; out->ok = 0; out->max_unique = 0; out->n_tiles = 0; out->n_blocks = 0;
  host_stats[0] = host_stats[1] = host_stats[2] = host_stats[3] = 0;
  if (N == 0) { out->ok = 1; return SPK_OK; }
  { int zr = spk_zero_async(dev_stats, 4 * sizeof(int32_t), stream); if (zr) return zr; }
  hipLaunchKernelGGL(k_blk_plan, dim3(ng), dim3(256), 0, stream, g->idx_j, g->rowptr, N, cap, (int32_t*)out->sub_n, (int32_t*)out->sub_u, (int32_t*)out->uniq,
                     (uint16_t*)out->jl, dev_stats);
  SPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_blk_tile_scan, dim3(1), dim3(1024), 0, stream, g->rowptr, N, (int32_t*)out->atom_tile0, dev_stats);
  SPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_blk_tile_fill, dim3((N + 255) / 256), dim3(256), 0, stream, g->rowptr, out->atom_tile0, N, (int32_t*)out->tile_info);
  SPK_LAUNCH_CHECK();
  // blocks: the sub-blocks of all groups in atom order (their first-block prefix is scratch behind the descriptors)
  int32_t* blk0 = (int32_t*)out->blk_desc + 4 * (int64_t)ng * BA;
  hipLaunchKernelGGL(k_blk_block_scan, dim3(1), dim3(1024), 0, stream, out->sub_n, ng, blk0, dev_stats);
  SPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_blk_desc_fill, dim3((ng + 255) / 256), dim3(256), 0, stream, out->sub_n, out->sub_u, blk0, g->rowptr, ng, N, (int32_t*)out->blk_desc);
  SPK_LAUNCH_CHECK();
  SPK_HIP_TRY(hipMemcpyAsync(host_stats, dev_stats, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  SPK_HIP_TRY(hipStreamSynchronize(stream));
  out->max_unique = host_stats[0];
  out->n_tiles = host_stats[2];
  out->n_blocks = host_stats[3];
  out->ok = host_stats[1] ? 0 : 1;
  return SPK_OK;
}

// Stand-alone prep (the whole-representation drivers call it once per force call; the message entry points run it themselves)
extern "C" int spk_blocks_prepare_f32(const spk_graph_t* g, const spk_radial_t* rb, const float* r_ij, void* stream) {
  SPK_CHECK_ARG(g && rb && g->blocks && g->blocks->ok, "spk_blocks_prepare_f32: no usable block plan");
  if (g->n_edges == 0) return SPK_OK;
  SPK_CHECK_ARG(r_ij, "spk_blocks_prepare_f32: null r_ij");
  MsgArgs a = {};
  a.rij = r_ij; a.rb = spk_radial_dev(rb); a.blocks = g->blocks; a.E = g->n_edges; a.N = g->n_atoms;
  return spk_painn_blk_prep(a, (hipStream_t)stream);
}
