// EXPERIMENT (round 5; default OFF -- SPK_FM_CHAIN=1 / spk_fm_set_chain(1) records them): row chains of the force-matching engine --
// consecutive ATOM-LOCAL launches of a pass as ONE launch.  Parity-green (tests/test_gpu_fm.py::test_row_chains_equal_the_launch_by_launch_step)
// and SLOWER than the launches it replaces; measurements, cycle stamps and the reading of them: profiles/r05_row_chains.md.
//
// Premise.  At configs[3] sizes (8 frames per GPU: 168 atoms) a training step is ~140 launches of 2-5 us of work each, and a launch of a replayed
// HIP graph costs ~4.5 us whatever it does (profiles/r04b_train_*_kernel_stats.csv: even k_fm_gr's 2.4 k items take 4 us).  Between two
// neighbour-coupled kernels (message / convolution and their transposes) every launch of the engine is atom-local: a Dense layer maps row i
// to row i, the mixing kernels of PaiNN touch rows of atom i only.  Such a run
//     Dense -> element-wise -> Dense -> Dense -> element-wise -> Dense -> Dense        (painn.py:99-116 + :31-48 of the next interaction)
// is here ONE launch: workgroup b owns FM_CHAIN_ATOMS consecutive atoms and walks the stages for its own rows, the results of a stage go
// to the same global buffers as before (the later passes read them); no grid-wide synchronisation, no atomics, the same arithmetic per
// output element.  PaiNN: 137 -> 62 launches per step, SchNet 100 -> 73.
//
// GEMM of a stage: P [R x NW] = X [R x K] op(W) with R = the <= 24 rows of the block (value / tangent stacks x 1 or 3 rows per atom x 4
// atoms).  Four rows are far below any 16- or 32-row MFMA tile, so the product runs on v_mfma_f32_4x4x1_16B_f32: sixteen independent
// 4 x 4 outer products per instruction -- here the SAME four rows of X against 64 different columns of W, one k per instruction; a wave
// covers 4 rows x 64 columns per accumulator, no padding rows, full fp32 rate (512 FLOP / 8 cycles).  X sits in LDS (replicated reads:
// lane l needs row l % 4), the columns of W stream from L2 (16-byte loads along k for y = x W^T, 4-byte loads for the transposed form).
//
// Finding.  A stage costs ~5 us however it is arranged -- 4x4x1 products are a few per cent of that.  Tried, in this order, each measured on the
// device (scripts/gpu_r5_e.sh, gpu_r5_f.sh, gpu_r5_g.sh): one weight load in flight (75 us per seven-stage chain) -> chunks of 16-32 k with the
// next chunk requested ahead, K split over the waves, Y handed to the next stage through LDS (46 us) -> LDS-only barriers + owner-consistent
// thread mapping so that no stage waits for global visibility (50 us) -> a warm-up pass that requests every weight line and external operand at
// kernel start (52 us; the pass itself costs 3 us per chain) -> two-phase epilogues with the next stage's weights requested before the stores
// (53 us, 19 k lines of code) -> rolled epilogues, out-of-line activation (50 us) -> descriptors copied to LDS once (50 us).  Then the stages
// were HOLLOWED OUT (SPK_FM_CHAIN_DRY bits; wrong results, timing only): without epilogue loads / stores 0.896 -> 0.889 ms per PaiNN step,
// without weight loads 0.844, without X loads 0.794, without the element-wise stages 0.756, without the warm-up 0.679 -- i.e. with NOTHING of
// a stage touching global memory the chained step takes what the launch-by-launch step takes (0.672 ms).  The memory round trips are ~25 % of a
// stage; the rest is the fixed cost of a generic stage interpreter on one workgroup: ~25 descriptor fields read and made uniform, three
// barriers with eight waves of different length, a work-item switch, an out-of-line activation, the LDS round trips of X, the partial tiles
// and Y -- ~3.5 us per stage, about what the launch it replaces costs.  What would win is not this kernel made faster but a hand-specialised
// kernel per chain (descriptor in SGPRs, one barrier per stage, operands resident in LDS, a store wave): profiles/r05_row_chains.md.
#pragma once
#include "spk_fm_kernels.h"

#define FM_CHAIN_ATOMS 4          // atoms per workgroup
#define FM_CHAIN_MAX_STAGES 12
#define FM_CHAIN_MAX_W 512        // K and NW of a stage
#define FM_CHAIN_MAX_ROWS 24      // 2 stacks x 3 rows per atom x 4 atoms

// epilogues (rows of the block = `ns` stacks of `rpa * FM_CHAIN_ATOMS` rows; stack s of the global tensors starts at row s * N * rpa)
enum {
  FM_G_DENSE = 0,      // every row: p += b; pre_out = p; y = act(p) + res                                   (be.dense)
  FM_G_BWD_INPUT,      // every row: x' = x act'(pre_in) (prologue, pre_in may be NULL); y = p + res           (be.dense_bwd_input; W used transposed)
  FM_G_DUAL_FWD,       // stacks (v, t): p_v += b; pre_out = [p_v ; p_t]; y_v = act(p_v), y_t = act'(p_v) p_t    (be.dense_dual without cutoff factors)
  FM_G_TANGENT,        // one stack: pre_out = p; y = act'(pre_in) p                                           (be.dense_tangent; either orientation of W)
  FM_G_DUAL_BWD        // stacks (g, h): a = pre_in_v, a_t = pre_in_t: y_v = p_g act'(a) + p_h act''(a) a_t, y_t = p_h act'(a)   (be.dense_dual_bwd)
};

struct FmGemmStage {
  const float* X;        // [ns * N * rpa, K]
  const float* W;        // trans == 0: [NW, K] (p = x W^T);  trans == 1: [K, NW] (p = x W)
  const float* b;        // [NW] or NULL
  const float* res;      // like Y or NULL
  const float* pre_in;   // BWD_INPUT: like X or NULL;  TANGENT: [N rpa, NW];  DUAL_BWD: [2 N rpa, NW]
  float* Y;              // [ns_out * N * rpa, NW]
  float* pre_out;        // like Y or NULL
  int K, NW, act, mode, trans, ns, rpa;
  int x_from_lds;        // the previous stage left this stage's X in LDS (its Y, same rows): no reload from memory
  int keep_y;            // leave Y in LDS for the next stage
  int ks;                // split of K over the waves (1, 2, 4, 8): (column group, k slice) work items, partial tiles summed by the epilogue
  int ext;               // bit 0 / 1 / 2: X / res / pre_in come from OUTSIDE the chain (an earlier launch): their lines are requested at kernel start
};
struct FmChainStage {
  int is_ew;
  FmGemmStage g;
  FmEwArgs<float> e;
};
struct FmChainDesc {
  int n_stages;
  int buf_floats;        // the dynamic LDS is two buffers of this many floats: X of a stage in one, its partial products (then Y) in the other
  int64_t N;             // atoms
  unsigned long long* dbg;   // cycle stamps of workgroup 0 (debugging aid, SPK_FM_CHAIN_STAMPS=1; NULL otherwise)
  int dry;                   // TIMING EXPERIMENT ONLY (SPK_FM_CHAIN_DRY=1, wrong results): no global operand loads and no global stores in the epilogues
  FmChainStage st[FM_CHAIN_MAX_STAGES];
};

#ifndef SPK_FM_EMU
typedef float fm_f32x4 __attribute__((ext_vector_type(4)));
#define FM_CHAIN_THREADS 512
#define FM_CHAIN_CHUNK 16          // k per weight chunk

// One work item: the R rows of the block x 64 columns [c0, c0 + 64) x the k range [k_lo, k_hi) -> partial tile Pp[lr][col].
// X in LDS (row stride ldx), W in global memory.  `first` holds the item's first weight chunk, requested by the caller before the X tile
// was complete.
template <int RG>
struct FmChainItem {
  // weight chunk of this lane's column (the k range of an item is a whole number of chunks: K % (ks * 32) == 0, checked when the stage is recorded):
  // trans == 0: W[col][k .. k + 32) as 8 x 16 bytes; trans == 1: W[k + q][col], q < 32, as 32 x 4 bytes
  static __device__ __forceinline__ void load(const float* __restrict__ W, int K, int NW, int trans, int colc, int k, float (&w)[FM_CHAIN_CHUNK]) {
    if (trans & 2) {          // timing experiment (SPK_FM_CHAIN_DRY & 2): no weight traffic
#pragma unroll
      for (int q = 0; q < FM_CHAIN_CHUNK; ++q) w[q] = 1.0f;
      return;
    }
    if (!trans) {
      const float* p = W + (int64_t)colc * K + k;
#pragma unroll
      for (int q = 0; q < FM_CHAIN_CHUNK / 4; ++q) {
        const fm_f32x4 v = *(const fm_f32x4*)(p + 4 * q);
        w[4 * q] = v[0]; w[4 * q + 1] = v[1]; w[4 * q + 2] = v[2]; w[4 * q + 3] = v[3];
      }
    } else {
      const float* p = W + (int64_t)k * NW + colc;
#pragma unroll
      for (int q = 0; q < FM_CHAIN_CHUNK; ++q) w[q] = p[(int64_t)q * NW];
    }
  }
  static __device__ __forceinline__ void run(const float* __restrict__ Xs, int ldx, const float* __restrict__ W, int K, int NW, int trans, int c0, int k_lo, int k_hi,
                                             float* __restrict__ Pp, int ldp, float (&wc)[FM_CHAIN_CHUNK]) {
    const int lane = threadIdx.x & 63;
    const int col = c0 + lane;
    const bool cv = col < NW;
    const int colc = cv ? col : NW - 1;
    fm_f32x4 acc[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) acc[g] = fm_f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xrow = Xs + (lane & 3) * ldx;
    for (int k = k_lo; k < k_hi; k += FM_CHAIN_CHUNK) {
      float wn[FM_CHAIN_CHUNK];
      const bool more = k + FM_CHAIN_CHUNK < k_hi;
      if (more) load(W, K, NW, trans, colc, k + FM_CHAIN_CHUNK, wn);       // the next chunk leaves before the products of this one
#pragma unroll
      for (int q4 = 0; q4 < FM_CHAIN_CHUNK / 4; ++q4) {
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          const fm_f32x4 av = *(const fm_f32x4*)(xrow + g * 4 * ldx + k + 4 * q4);
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[m], wc[4 * q4 + m], acc[g], 0, 0, 0);
        }
      }
      if (more) {
#pragma unroll
        for (int q = 0; q < FM_CHAIN_CHUNK; ++q) wc[q] = wn[q];
      }
    }
    if (cv) {
#pragma unroll
      for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int v = 0; v < 4; ++v) Pp[(g * 4 + v) * ldp + col] = acc[g][v];
    }
  }
};

// Who touches what (the reason a chain needs no wait for global memory between its stages): thread (a, f) = (tid / 128, tid % 128) of the
// 512 OWNS the rows of atom a0 + a and, of every tensor, the columns c with c % 128 == f.  Every global value a stage writes is written by its
// owner, and every global value a later stage of the SAME chain reads (X of a Dense stage, residuals, saved pre-activations, the operands of the
// element-wise kernels -- whose item (i, f) touches columns f, F + f, 2 F + f of rows of atom i, with F a multiple of 128) is read by its
// owner: a store followed by a load of the same address in ONE thread needs no fence.  What waves exchange goes through LDS (the X tile, the
// partial products), behind LDS-only barriers -- `__syncthreads()` would drain the vector-memory counter, i.e. wait ~2 us per stage for the
// stores of the saved tensors to be acknowledged (first version of this kernel: 3.7 us per stage; spk_painn_mol.hip found the same).
#define FM_CHAIN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define FM_CHAIN_STAMP(dbgp, slot)                                                                    \
  do {                                                                                                \
    if ((dbgp) && blockIdx.x == 0 && threadIdx.x == 0) (dbgp)[slot] = __builtin_readcyclecounter();    \
  } while (0)
#define FM_CHAIN_CW 128

// first weight chunk of the wave's first work item of a stage (W, K, NW, trans, ks of that stage)
__device__ __forceinline__ void fm_chain_first_chunk(const float* W, int K, int NW, int trans, int ks_n, float (&wc)[FM_CHAIN_CHUNK]) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_cg = (NW + 63) >> 6;
  if (wv < n_cg * ks_n) {
    const int cg = wv % n_cg, ks = wv / n_cg;
    const int col = cg * 64 + lane, colc = col < NW ? col : NW - 1;
    FmChainItem<1>::load(W, K, NW, trans, colc, ks * (K / ks_n), wc);
  }
}

#define FM_CHAIN_MC 3       // columns a thread owns in one row (NW <= 3 x 128)
// one copy of the activation code (value, first and second derivative) instead of one per use
__device__ __attribute__((noinline)) float fm_chain_act(int act, int order, float z) { return fm_act<float>(act, order, z); }

// X of the stage in bufX (row stride ldx; loaded here unless inherited), partial products -> bufP, then the epilogue.  `wc` arrives holding the
// first weight chunk of this wave's first work item and leaves holding that of the NEXT Dense stage (nW .. nks; nW == NULL: none): those loads
// -- like every operand of the epilogue -- are requested BEFORE the first store of the stage (a load behind a store waits for the store).
// Returns the row stride of Y in bufP (for a stage that inherits it).
__device__ __forceinline__ int fm_chain_gemm_stage(const FmGemmStage& g, int64_t N, int64_t a0, float* bufX, int ldx_in, float* bufP, float (&wc)[FM_CHAIN_CHUNK],
                                                   const float* nW, int nK, int nNW, int ntrans, int nks, unsigned long long* dbg, int dry) {
  const int tid = threadIdx.x;
  const int K = g.K, NW = g.NW;
  const int ns_in = g.mode == FM_G_TANGENT ? 1 : g.ns;
  const int per = g.rpa * FM_CHAIN_ATOMS;           // rows of one stack in the block
  const int R = ns_in * per;
  const int RG = R >> 2;
  const int ldx = g.x_from_lds ? ldx_in : K + 4;     // (+4: the rows of a 4-row group land in different banks for the replicated 16-byte reads)
  const int ldp = NW + 4;
  const int wv = tid >> 6, nw = FM_CHAIN_THREADS >> 6, lane = tid & 63;
  const int n_cg = (NW + 63) >> 6, KS = g.ks;
  const int kslice = K / KS;
  const int n_items = n_cg * KS;
  // ---- the owner's share of X -> LDS (BWD_INPUT: times act'(pre_in))
  const int oa = tid / FM_CHAIN_CW, of = tid - oa * FM_CHAIN_CW;       // owned atom of the block, owned column class
  const bool atom_ok = a0 + oa < N;
  const int64_t grow0 = (a0 + oa) * g.rpa;                                // first global row of the atom inside a stack
  const int64_t stack = N * g.rpa;                                        // global rows per stack
  if (!g.x_from_lds) {
    for (int sr = 0; sr < ns_in * g.rpa; ++sr) {
      const int st = sr / g.rpa, r = sr - st * g.rpa;
      const int lr = st * per + oa * g.rpa + r;
      const int64_t gr = st * stack + grow0 + r;
      for (int c = of; c < K; c += FM_CHAIN_CW) {
        float v = 0.f;
        if (atom_ok) {
          v = g.X[gr * K + c];
          if (g.mode == FM_G_BWD_INPUT && g.pre_in) v *= fm_chain_act(g.act, 1, g.pre_in[gr * K + c]);
        }
        bufX[lr * ldx + c] = v;
      }
    }
  }
  FM_CHAIN_STAMP(dbg, 0);
  FM_CHAIN_LDS_BARRIER();
  FM_CHAIN_STAMP(dbg, 1);
  // ---- partial products
  for (int it = wv; it < n_items; it += nw) {
    const int cg = it % n_cg, ks = it / n_cg;
    const int k_lo = ks * kslice, k_hi = k_lo + kslice;
    float* Pp = bufP + ks * (R * ldp);
    if (it != wv) {
      const int col = cg * 64 + lane, colc = col < NW ? col : NW - 1;
      FmChainItem<1>::load(g.W, K, NW, g.trans, colc, k_lo, wc);
    }
#define FM_CHAIN_RUN(RGv) FmChainItem<RGv>::run(bufX, ldx, g.W, K, NW, g.trans, cg * 64, k_lo, k_hi, Pp, ldp, wc)
    switch (RG) {
      case 1: FM_CHAIN_RUN(1); break;
      case 2: FM_CHAIN_RUN(2); break;
      case 3: FM_CHAIN_RUN(3); break;
      default: FM_CHAIN_RUN(6); break;
    }
#undef FM_CHAIN_RUN
  }
  FM_CHAIN_STAMP(dbg, 2);
  FM_CHAIN_LDS_BARRIER();
  FM_CHAIN_STAMP(dbg, 3);
  // ---- epilogue, by owner: sum of the partial tiles -> global (and back into the first partial tile when the next stage reads Y from LDS).
  // ROLLED loops and an out-of-line activation on purpose: the unrolled two-phase form of this epilogue (all operand loads, the next stage's
  // weights, then all stores) made the kernel 19 k lines of straight-line code for no gain (0.93 -> 0.88 ms per PaiNN step when rolled back).
  const int pstride = R * ldp;
  const bool dual = g.mode == FM_G_DUAL_FWD || g.mode == FM_G_DUAL_BWD;
  const int nrow = dual ? g.rpa : ns_in * g.rpa;
  FM_CHAIN_STAMP(dbg, 4);
  if (nW) fm_chain_first_chunk(nW, nK, nNW, ntrans, nks, wc);
  for (int r = 0; r < nrow; ++r) {
    const int st = dual ? 0 : r / g.rpa, rr = dual ? r : r - st * g.rpa;
    const int lr = st * per + oa * g.rpa + rr;
    const int64_t gr = st * stack + grow0 + rr;
    for (int c = of; c < NW; c += FM_CHAIN_CW) {
      float p = bufP[lr * ldp + c];
      for (int q = 1; q < KS; ++q) p += bufP[q * pstride + lr * ldp + c];
      if (!dual) {
        if (atom_ok) {
          if (g.mode == FM_G_DENSE) {
            if (g.b) p += g.b[c];
            if (g.pre_out && !dry) g.pre_out[gr * NW + c] = p;
            p = fm_chain_act(g.act, 0, p);
            if (g.res && !dry) p += g.res[gr * NW + c];
          } else if (g.mode == FM_G_BWD_INPUT) {
            if (g.res && !dry) p += g.res[gr * NW + c];
          } else {
            if (g.pre_out && !dry) g.pre_out[gr * NW + c] = p;
            p *= fm_chain_act(g.act, 1, dry ? p : g.pre_in[gr * NW + c]);
          }
          if (!dry) g.Y[gr * NW + c] = p;
        } else p = 0.f;
        if (g.keep_y) bufP[lr * ldp + c] = p;
      } else {
        const int lt = per + lr;
        const int64_t gt = gr + stack;
        float pt = bufP[lt * ldp + c];
        for (int q = 1; q < KS; ++q) pt += bufP[q * pstride + lt * ldp + c];
        float yv = 0.f, yt = 0.f;
        if (atom_ok) {
          if (g.mode == FM_G_DUAL_FWD) {
            if (g.b) p += g.b[c];
            if (g.pre_out && !dry) { g.pre_out[gr * NW + c] = p; g.pre_out[gt * NW + c] = pt; }
            yv = fm_chain_act(g.act, 0, p);
            yt = fm_chain_act(g.act, 1, p) * pt;
          } else {      // FM_G_DUAL_BWD
            const float a = dry ? p : g.pre_in[gr * NW + c], at = dry ? pt : g.pre_in[gt * NW + c], a1 = fm_chain_act(g.act, 1, a);
            yv = p * a1 + pt * fm_chain_act(g.act, 2, a) * at;
            yt = pt * a1;
          }
          if (!dry) { g.Y[gr * NW + c] = yv; g.Y[gt * NW + c] = yt; }
        }
        if (g.keep_y) { bufP[lr * ldp + c] = yv; bufP[lt * ldp + c] = yt; }
      }
    }
  }
  FM_CHAIN_STAMP(dbg, 5);
  FM_CHAIN_LDS_BARRIER();       // Y in LDS (when kept) is complete; the partial tiles may be overwritten by the stage after the next
  FM_CHAIN_STAMP(dbg, 6);
  return ldp;
}

// a value every lane holds (read from LDS) moved to scalar registers
__device__ __forceinline__ int fm_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class P>
__device__ __forceinline__ P* fm_uni(P* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (P*)(((uint64_t)hi << 32) | lo);
}

__global__ __launch_bounds__(FM_CHAIN_THREADS) void k_fm_chain(FmChainDesc d_) {
  // The stage descriptors are kernel arguments (2.7 KB).  They are copied into LDS once (all lines of the kernel-argument segment requested
  // together) and read from there, made wave-uniform with v_readfirstlane.  (Tried because the per-stage scalar loads from the argument segment
  // were a suspect for the fixed cost of a stage; the step time did not move: profiles/r05_row_chains.md.)
  __shared__ FmChainDesc sdesc;
  {
    const int* src = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
    int* dst = (int*)&sdesc;
    for (int i = threadIdx.x; i < (int)(sizeof(FmChainDesc) / 4); i += FM_CHAIN_THREADS) dst[i] = src[i];
  }
  (void)d_;
  __syncthreads();
  const FmChainDesc* d = &sdesc;
  extern __shared__ __attribute__((aligned(16))) float fm_chain_lds[];
  const int64_t N = ((int64_t)fm_uni((int)(d->N >> 32)) << 32) | (uint32_t)fm_uni((int)(d->N & 0xffffffff));
  const int64_t a0 = (int64_t)blockIdx.x * FM_CHAIN_ATOMS;
  const int n_st = fm_uni(d->n_stages);
  const int bf = fm_uni(d->buf_floats);
  const int dryf = fm_uni(d->dry);
  unsigned long long* const dbgp = fm_uni(d->dbg);
  FM_CHAIN_STAMP(dbgp, 0);
  // ---- warm-up: one request per 128-byte line of every weight matrix and bias of the chain, and of the block's rows of every operand that an
  // EARLIER launch wrote -- all independent, all in flight at once.  Every stage of a chain meets different weights, and what earlier launches
  // wrote sits behind the memory-side cache: cold, each stage paid two or three dependent misses of ~2 us (measured: ~5 us per stage with the
  // barriers already LDS-only); warmed, the stages find their operands in this XCD's L2.  (Operands written INSIDE the chain are not touched
  // here: their owner has not stored them yet.)
  {
    float dummy = 0.f;
    const int oa_ = threadIdx.x / FM_CHAIN_CW, of_ = threadIdx.x - oa_ * FM_CHAIN_CW;
    const bool aok = a0 + oa_ < N;
    for (int si = 0; si < n_st && !(dryf & 16); ++si) {
      if (fm_uni(d->st[si].is_ew)) continue;
      const float* W = fm_uni(d->st[si].g.W);
      const float* b = fm_uni(d->st[si].g.b);
      const int K = fm_uni(d->st[si].g.K), NW = fm_uni(d->st[si].g.NW), ext = fm_uni(d->st[si].g.ext), rpa = fm_uni(d->st[si].g.rpa), mode = fm_uni(d->st[si].g.mode);
      const int ns_in = mode == FM_G_TANGENT ? 1 : fm_uni(d->st[si].g.ns);
      for (int i = threadIdx.x * 32; i < K * NW; i += FM_CHAIN_THREADS * 32) dummy += W[i];
      if (b && threadIdx.x * 32 < NW) dummy += b[threadIdx.x * 32];
      if (aok && ext) {
        const float* X = fm_uni(d->st[si].g.X);
        const float* res = fm_uni(d->st[si].g.res);
        const float* pin = fm_uni(d->st[si].g.pre_in);
        const int wpin = mode == FM_G_BWD_INPUT ? K : NW;                      // width of pre_in
        const int ns_pin = mode == FM_G_TANGENT ? 1 : fm_uni(d->st[si].g.ns);
        const int ns_res = ns_in;
        for (int sr = 0; sr < 2 * rpa; ++sr) {
          const int st = sr / rpa, r = sr - st * rpa;
          const int64_t gr = st * N * rpa + (a0 + oa_) * rpa + r;
          if ((ext & 1) && st < ns_in && of_ * 32 < K) dummy += X[gr * K + of_ * 32];
          if ((ext & 2) && res && st < ns_res && of_ * 32 < NW) dummy += res[gr * NW + of_ * 32];
          if ((ext & 4) && pin && st < ns_pin && of_ * 32 < wpin) dummy += pin[gr * wpin + of_ * 32];
        }
      }
    }
    asm volatile("" ::"v"(dummy));
  }
  int cur = 0, ldx = 0;
  FM_CHAIN_STAMP(dbgp, 1);
  float wc[FM_CHAIN_CHUNK];
  // index of the first Dense stage at or behind `from` (n_st: none)
  auto next_gemm = [&](int from) { int j = from; while (j < n_st && fm_uni(d->st[j].is_ew)) ++j; return j; };
  {
    const int j = next_gemm(0);
    if (j < n_st) fm_chain_first_chunk(fm_uni(d->st[j].g.W), fm_uni(d->st[j].g.K), fm_uni(d->st[j].g.NW), fm_uni(d->st[j].g.trans) | (dryf & 2), fm_uni(d->st[j].g.ks), wc);
  }
  for (int si = 0; si < n_st; ++si) {
    if (fm_uni(d->st[si].is_ew)) {
      if (dryf & 8) continue;
      FmEwArgs<float> e;
      e.kind = fm_uni(d->st[si].e.kind); e.F = fm_uni(d->st[si].e.F); e.N = N; e.eps = d->st[si].e.eps;
#pragma unroll
      for (int q = 0; q < 8; ++q) e.in[q] = fm_uni(d->st[si].e.in[q]);
#pragma unroll
      for (int q = 0; q < 3; ++q) e.out[q] = fm_uni(d->st[si].e.out[q]);
      // item (atom, channel) by its owner: with F a multiple of 128 every operand is the owner's own (no wait); any other width takes the
      // conservative route -- all earlier stores of the workgroup visible before, all of this stage's after
      const bool owned = (e.F % FM_CHAIN_CW) == 0;
      if (!owned) __syncthreads();
      const int oa = threadIdx.x / FM_CHAIN_CW, of = threadIdx.x - oa * FM_CHAIN_CW;
      if (a0 + oa < N)
        for (int ch = of; ch < e.F; ch += FM_CHAIN_CW) fm_ew_at<float>(e, (a0 + oa) * e.F + ch);
      if (!owned) __syncthreads();
      FM_CHAIN_STAMP(dbgp, 2 + 8 * si + 6);
    } else {
      FmGemmStage g;
      g.X = fm_uni(d->st[si].g.X); g.W = fm_uni(d->st[si].g.W); g.b = fm_uni(d->st[si].g.b); g.res = fm_uni(d->st[si].g.res); g.pre_in = fm_uni(d->st[si].g.pre_in);
      g.Y = fm_uni(d->st[si].g.Y); g.pre_out = fm_uni(d->st[si].g.pre_out); g.K = fm_uni(d->st[si].g.K); g.NW = fm_uni(d->st[si].g.NW); g.act = fm_uni(d->st[si].g.act);
      g.mode = fm_uni(d->st[si].g.mode); g.trans = fm_uni(d->st[si].g.trans); g.ns = fm_uni(d->st[si].g.ns); g.rpa = fm_uni(d->st[si].g.rpa);
      g.x_from_lds = fm_uni(d->st[si].g.x_from_lds); g.keep_y = fm_uni(d->st[si].g.keep_y); g.ks = fm_uni(d->st[si].g.ks);
      if (dryf & 2) g.trans |= 2;
      if (dryf & 4) g.x_from_lds = 1;          // (timing experiment: X is whatever the buffer holds)
      const int j = next_gemm(si + 1);
      const float* nW = j < n_st ? fm_uni(d->st[j].g.W) : nullptr;
      const int nK = j < n_st ? fm_uni(d->st[j].g.K) : 0, nNW = j < n_st ? fm_uni(d->st[j].g.NW) : 0, ntr = (j < n_st ? fm_uni(d->st[j].g.trans) : 0) | (dryf & 2),
                nks = j < n_st ? fm_uni(d->st[j].g.ks) : 1;
      ldx = fm_chain_gemm_stage(g, N, a0, fm_chain_lds + cur * bf, ldx, fm_chain_lds + (cur ^ 1) * bf, wc, nW, nK, nNW, ntr, nks, dbgp ? dbgp + 2 + 8 * si : nullptr, dryf & 1);
      cur ^= 1;             // Y (when kept) is in the buffer the next stage reads its X from
    }
  }
}
#endif
