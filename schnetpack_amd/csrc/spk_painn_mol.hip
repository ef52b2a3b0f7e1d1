// Molecule-resident PaiNN representation (representation/painn.py:207-256) for BATCHES OF SMALL MOLECULES.
//
// A batch produced by the reference's collate function (data/loader.py:35-46) is block diagonal: no edge leaves a molecule.
// With <= 32 atoms per block the whole representation of a block -- every interaction: inter-atomic context net, equivariant
// message, channel mix + intra-atomic context net + update (painn.py:31-67, :92-117) -- is local to ONE workgroup: q [32 x F],
// mu [3 x 32 x F] and the context rows c [3 x 32 x F] live in LDS for the whole kernel, the 2 x 3 x 512-byte neighbour gathers of
// the message never leave the compute unit, and nothing but the saved-for-backward tensors goes to memory.  One launch replaces
// the 3 L launches of the general driver (spk_painn.hip: context chain, message, mixing per interaction).
//
// Workgroup = 8 wavefronts, one group of atoms (a block, or several small blocks, <= 32 atoms, <= 768 directed edges).
// Per interaction (B = workgroup barrier):
//   P1  pre_a = W_a1 q + b          (T-GEMM, 4 feature tiles)             -> saved, silu -> sH                     B
//   P2  c = W_a2 silu(pre_a) + b    (12 feature tiles over the 8 waves)   -> saved, LDS planes (q | R | mu part)   B
//   P3  message: one wavefront per centre atom walks its CSR row; a lane owns two channels; the filter slice
//       Phi_e = (phi(d_e) W_f^T + b_f) f_c(d_e) is recomputed per edge from register-resident weights; c_j / mu_j come from LDS;
//       the row sums stay in registers                                                                             B, write, B
//   P4  (V | W) = mu W_mix^T for the three components (wave t: feature tile t of V and of W, so |V|, sum_x V W and the
//       update are register-local), |V| -> LDS, mix -> saved                                                       B
//   P5  pre_b = W_b1 [q | |V|] + b  (K = 256)                              -> saved, silu -> sH                     B
//   P6  a = W_b2 silu(pre_b) + b -> saved;  q += a_q + a_qmu sum_x V W;  mu += a_mu W                               B
// fp32 MFMA (v_mfma_f32_32x32x2_f32) throughout: 1e-5 parity with the reference rules out bf16.  Deterministic: no atomics.
// The saved tensors have exactly the layout of the general driver (spk_painn_saved_floats), so either backward can follow.
#include "spk_common.h"
#include "spk_pack.h"
#include "spk_painn_mol.h"
#include "spk_split.h"

#define PM_MAXL 6
#define PM_LD 132                 // row stride (floats) of the [32][128] tiles in LDS: conflict-free 16-byte accesses
#define PM_TILE (32 * PM_LD)
#define PM_MAXEDGES 768           // directed edges per group (= 2 x 384 pairs, the plan's bound)
#define PM_NRBF 20                // register-resident filter weights: n_rbf <= 20
#define PM_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)

typedef float pm_f2 __attribute__((ext_vector_type(2)));

struct PmLayerDev {
  const float *ctx1_p, *ctx1_b;   // packed forward image of interatomic_context_net.0.weight [F, F], bias
  const float *ctx2_p, *ctx2_b;   // ... .1.weight [3F, F]
  const float* mix_p;             // mixing.mu_channel_mix.weight [2F, F]
  const float *ic1_p, *ic1_b;     // intraatomic_context_net.0.weight [F, 2F]
  const float *ic2_p, *ic2_b;     // ... .1.weight [3F, F]
  const float *wf, *bf;           // filter_net rows of this interaction, raw [3F, n_rbf], [3F]
};

struct PmFwdArgs {
  PmLayerDev L[PM_MAXL];
  int n_layers;
  const float* q0;          // [N, 128]
  float* q_out;             // [N, 128]
  float* mu_out;            // [N, 3, 128]
  const float* rij;         // [E, 3]
  const float *R, *offsets; // POT: positions [N, 3] (+ offsets [E, 3] or null) instead of rij
  float* rij_out;           // POT: [E, 3] the pair vectors, for the backward launch
  float* gq_out;            // POT: [N, 128] dE/dq_L = the gradient of the summed energy through the head, for the backward launch
  const int64_t* idx_i;     // POT
  const float* emb;         // POT with q0 == null: rows of the nuclear embedding table [n_types, 128] ...
  const int64_t* Z;         // ... by atomic number [N]
  int n_types;              // rows of emb: a Z outside [0, n_types) reads nothing and poisons its molecule with NaN (nn.Embedding raises)
  PmHeadDev head;           // POT
  const int64_t* idx_j;
  const int32_t* rowptr;    // CSR of idx_i
  const int32_t* grp_atom0; // [G+1]
  int n_groups;
  float* saved;             // per interaction: preA [N,F] | c [N,3F] | mu_in [N,3F] | mix [N,6F] | preB [N,F] | a [N,3F]
  int64_t N;
  float eps;
  RadialDev rb;
  long long* dbg;           // tuning aid: cycle stamps of thread 0 of workgroup 0 (spk_painn_mol_set_debug_buffer; null in production)
  int assign;               // tuning: 0 = dynamic (default), 1 = snake over the waves, >= 20: static greedy with this cost per edge of the younger wave
  int split;                // host side: 1 = a.L[] holds the SPLIT weight images, launch the SP instances (spk_split.h)
  int tiled;                // host side: 1 = launch the instance with the message on the matrix core (Gaussian bases; SPK_PM_TILED=0: row form)
};
#define PM_STAMP(n) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[n] = (long long)__builtin_readcyclecounter(); } while (0)

// per directed edge of a group: local neighbour, unit vector, distance, cutoff value -- computed once per group
struct __attribute__((aligned(8))) PmEdge { int jl; float ux, uy, uz, d, fc; };

template <class T>
__device__ __forceinline__ T pm_ld(const void* sbase, unsigned voff) { return *(const T*)((const char*)sbase + voff); }
template <class T>
__device__ __forceinline__ void pm_st(void* sbase, unsigned voff, T v) { *(T*)((char*)sbase + voff) = v; }
// register r of the half hi of a 32x32 accumulator holds row (r & 3) + 8 (r >> 2) + 4 hi
__device__ __forceinline__ int pm_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---- T-GEMM pieces (convention of spk_dense.hip): A = packed weights straight from L2, B = activations [32][PM_LD] in LDS;
// accumulator rows = output features 32 t + pm_row(r, hi), columns = atoms (lane & 31).
// Packed image: P[((t * KB + ug) * 64 + lane) * 4 + v] = W[32 t + (lane & 31)][8 ug + 4 (lane >> 5) + v]; one k-block = 1024 bytes.
__device__ __forceinline__ void pm_load8(f32x4 (&av)[8], const char* sb /* wave-uniform */, int lane) {
#pragma unroll
  for (int u = 0; u < 8; ++u) av[u] = pm_ld<f32x4>(sb, (unsigned)(lane * 16 + u * 1024));
}
// (B operands are requested four groups ahead: left alone the compiler issues each ds_read right in front of the four MFMAs
//  that need it -- zero prefetch distance, the matrix pipe waits ~100 cycles per group of four)
// SP (spk_split.h): `av` holds eight chunks of the SPLIT weight image (chunk 2 s = the high parts of k-step s, 2 s + 1 its 2^11-scaled low
// parts; same bytes and offsets as the fp32 image, k_pack_weight_split), the activations are read as eight consecutive fp32 values
// per k-step (brow is the fp32 form's address, lane row + 4 hi: 4 hi more make it row + 8 hi) and split in registers; three f16
// instructions per 16 k instead of eight f32 ones, the cross terms in their own accumulator, folded in at the end of the block.
template <bool SP = false>
__device__ __forceinline__ f32x16 pm_mma8(const f32x4 (&av)[8], const float* __restrict__ brow, f32x16 acc) {
  if constexpr (SP) {
    const float* b = brow + 4 * ((threadIdx.x >> 5) & 1);
    f32x16 cx;
#pragma unroll
    for (int r = 0; r < 16; ++r) cx[r] = 0.f;
    f32x4 b0[4], b1[4];
    b0[0] = *(const f32x4*)b; b1[0] = *(const f32x4*)(b + 4);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4) { b0[s + 1] = *(const f32x4*)(b + 16 * (s + 1)); b1[s + 1] = *(const f32x4*)(b + 16 * (s + 1) + 4); asm volatile("" ::: "memory"); }
      h16x4 h0, l0, h1, l1;
      sp_split4(b0[s], h0, l0);
      sp_split4(b1[s], h1, l1);
      SP_STEP(__builtin_bit_cast(h16x8, av[2 * s]), __builtin_bit_cast(h16x8, av[2 * s + 1]), sp_cat(h0, h1), sp_cat(l0, l1), acc, cx);
    }
    SP_FOLD(acc, cx);
    return acc;
  }
  f32x4 bv[8];
#pragma unroll
  for (int u = 0; u < 4; ++u) bv[u] = *(const f32x4*)(brow + 8 * u);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    acc = PM_MFMA(av[u].x, bv[u].x, acc);
    acc = PM_MFMA(av[u].y, bv[u].y, acc);
    acc = PM_MFMA(av[u].z, bv[u].z, acc);
    acc = PM_MFMA(av[u].w, bv[u].w, acc);
    if (u + 4 < 8) {
      bv[u + 4] = *(const f32x4*)(brow + 8 * (u + 4));
      asm volatile("" ::: "memory");
    }
  }
  return acc;
}
// 16 k-blocks (K = 128 of the image's row, starting at k-block kb0) against ONE LDS tile
template <bool SP = false>
__device__ __forceinline__ f32x16 pm_tile16(const float* __restrict__ wp, int KB, int t /* wave-uniform */, int kb0, const float* __restrict__ sB, int lane,
                                            f32x16 acc) {
  const char* sb = (const char*)wp + ((size_t)t * KB + kb0) * 1024;
  const float* brow = sB + (lane & 31) * PM_LD + 4 * (lane >> 5);
  f32x4 a0[8], a1[8];
  pm_load8(a0, sb, lane);
  pm_load8(a1, sb + 8 * 1024, lane);
  acc = pm_mma8<SP>(a0, brow, acc);
  acc = pm_mma8<SP>(a1, brow + 64, acc);
  return acc;
}
// the same weights against the THREE component planes of mu (A operand shared: 96 MFMAs per 8 k-blocks)
template <bool SP = false>
__device__ __forceinline__ void pm_tile16x3(const float* __restrict__ wp, int t, const float* __restrict__ sB, int lane, f32x16& c0, f32x16& c1, f32x16& c2) {
  const char* sb = (const char*)wp + (size_t)t * 16 * 1024;
  const float* brow = sB + (lane & 31) * PM_LD + 4 * (lane >> 5);
  f32x4 a0[8], a1[8];
  pm_load8(a0, sb, lane);
  pm_load8(a1, sb + 8 * 1024, lane);
  c0 = pm_mma8<SP>(a0, brow, c0);
  c1 = pm_mma8<SP>(a0, brow + PM_TILE, c1);
  c2 = pm_mma8<SP>(a0, brow + 2 * PM_TILE, c2);
  c0 = pm_mma8<SP>(a1, brow + 64, c0);
  c1 = pm_mma8<SP>(a1, brow + PM_TILE + 64, c1);
  c2 = pm_mma8<SP>(a1, brow + 2 * PM_TILE + 64, c2);
}
__device__ __forceinline__ f32x16 pm_bias_acc(const float* __restrict__ b, int t, int hi) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = b[32 * t + pm_row(r, hi)];
  return acc;
}
__device__ __forceinline__ float pm_silu(float x) { return x * spk_sigmoid(x); }
__device__ __forceinline__ float pm_silu_grad(float x);

// phi_k(d) for the lane's own k (nn/radial.py:11-15 gaussian, :105-110 bessel), parameters preloaded
__device__ __forceinline__ float pm_phi(int kind, float p0k, float p1k, float d) {
  if (kind == SPK_RBF_GAUSSIAN) {
    const float c = -0.5f / (p1k * p1k);
    const float t = d - p0k;
    return expf(c * t * t);
  }
  const float s = sinf(p0k * d);
  return d == 0.0f ? s : s / d;
}

// Workgroup barrier that waits for the LDS traffic of the wave only (like ck's block_sync_lds): __syncthreads() also drains
// vmcnt, i.e. waits until the saved-for-backward STORES of the phase have been acknowledged by L2 (~5 k cycles each time) and
// until the weight tiles prefetched for the next phase have arrived -- neither is needed at the barrier.
#define PM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// A feature tile of packed weights on its way from L2.  Only the FIRST eight k-blocks travel across a phase boundary (32
// registers); the following chunks are requested when the tile is used, each one chunk (32 MFMAs = 2 k cycles) ahead of its use.
// (Whole tiles held across a barrier -- 64 to 96 registers next to the accumulators -- were spilled by the register allocator
// RIGHT BEHIND their loads: load, wait, store to scratch, one L2 round trip after the other.)
struct PmW { f32x4 a0[8]; const char* sb; };
__device__ __forceinline__ void pm_wload(PmW& W, const float* __restrict__ wp, int KB, int t /* wave-uniform */, int kb0, int lane) {
  W.sb = (const char*)wp + ((size_t)t * KB + kb0) * 1024;
  pm_load8(W.a0, W.sb, lane);
}
// 16 k-blocks against one LDS tile
template <bool SP = false>
__device__ __forceinline__ f32x16 pm_wmma(const PmW& W, const float* __restrict__ sB, int lane, f32x16 acc) {
  const float* brow = sB + (lane & 31) * PM_LD + 4 * (lane >> 5);
  f32x4 a1[8];
  pm_load8(a1, W.sb + 8 * 1024, lane);
  acc = pm_mma8<SP>(W.a0, brow, acc);
  acc = pm_mma8<SP>(a1, brow + 64, acc);
  return acc;
}
// 16 k-blocks, the same weights against the three component planes of an LDS tensor
template <bool SP = false>
__device__ __forceinline__ void pm_wmma3(const PmW& W, const float* __restrict__ sB, int lane, f32x16& c0, f32x16& c1, f32x16& c2) {
  const float* brow = sB + (lane & 31) * PM_LD + 4 * (lane >> 5);
  f32x4 a1[8];
  pm_load8(a1, W.sb + 8 * 1024, lane);
  c0 = pm_mma8<SP>(W.a0, brow, c0);
  c1 = pm_mma8<SP>(W.a0, brow + PM_TILE, c1);
  c2 = pm_mma8<SP>(W.a0, brow + 2 * PM_TILE, c2);
  c0 = pm_mma8<SP>(a1, brow + 64, c0);
  c1 = pm_mma8<SP>(a1, brow + PM_TILE + 64, c1);
  c2 = pm_mma8<SP>(a1, brow + 2 * PM_TILE + 64, c2);
}
// 24 k-blocks against three half tiles (b0, b1, b2: LDS addresses of the lane's row, k offset included)
template <bool SP = false>
__device__ __forceinline__ f32x16 pm_wmma_3chunks(const PmW& W, const float* __restrict__ b0, const float* __restrict__ b1, const float* __restrict__ b2, int lane,
                                                  f32x16 acc) {
  f32x4 a1[8], a2[8];
  pm_load8(a1, W.sb + 8 * 1024, lane);
  acc = pm_mma8<SP>(W.a0, b0, acc);
  pm_load8(a2, W.sb + 16 * 1024, lane);
  acc = pm_mma8<SP>(a1, b1, acc);
  acc = pm_mma8<SP>(a2, b2, acc);
  return acc;
}
// a whole tile in registers (the W_mix^T half tile that serves the three components of M4)
struct PmWF { f32x4 a0[8], a1[8]; };
__device__ __forceinline__ void pm_wfload(PmWF& W, const float* __restrict__ wp, int KB, int t, int kb0, int lane) {
  const char* sb = (const char*)wp + ((size_t)t * KB + kb0) * 1024;
  pm_load8(W.a0, sb, lane);
  pm_load8(W.a1, sb + 8 * 1024, lane);
}
template <bool SP = false>
__device__ __forceinline__ f32x16 pm_wfmma(const PmWF& W, const float* __restrict__ sB, int lane, f32x16 acc) {
  const float* brow = sB + (lane & 31) * PM_LD + 4 * (lane >> 5);
  acc = pm_mma8<SP>(W.a0, brow, acc);
  acc = pm_mma8<SP>(W.a1, brow + 64, acc);
  return acc;
}

// filter rows 2 lane, 2 lane + 1 of the three parts (q | R | mu) of the filter net as they lie in memory: K consecutive floats per
// row, [part][row][chunk of 4 k]; one more chunk holds (bias, 0, 0, 0): the bias is the weight of an extra basis function whose
// value is f_c(d), so that  Phi f_c = sum_k w_k (f_c phi_k) + b f_c  is ONE contraction (no separate multiplications by f_c)
template <int K>
struct PmFilt { f32x4 w[3][2][K / 4 + 1]; };
template <int K>
__device__ __forceinline__ void pm_filt_load(PmFilt<K>& Wf, const float* __restrict__ wf, const float* __restrict__ bf, int lane, bool mu0) {
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const float* r0 = wf + (size_t)(p * 128 + 2 * lane) * K;
    const bool off = mu0 && p == 2;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
      for (int c = 0; c < K / 4; ++c) Wf.w[p][ch][c] = !off ? *(const f32x4*)(r0 + ch * K + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
      Wf.w[p][ch][K / 4] = f32x4{off ? 0.f : bf[p * 128 + 2 * lane + ch], 0.f, 0.f, 0.f};
    }
  }
}

// F = sum over the K + 4 slots of w * basis for the lane's two channels of one part: products over PAIRS of slots (the pair
// (w[k], w[k+1]) is adjacent in the loaded row chunk, the pair of basis values in the broadcast LDS read: v_pk_fma_f32 without
// any repacking), the two partial sums of a channel meet at the end
template <int NC>
__device__ __forceinline__ pm_f2 pm_filter2(const f32x4 (&w0)[NC], const f32x4 (&w1)[NC], const f32x4 (&ph)[NC]) {
  pm_f2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    s0 = pm_f2{w0[c].x, w0[c].y} * pm_f2{ph[c].x, ph[c].y} + s0;
    s1 = pm_f2{w1[c].x, w1[c].y} * pm_f2{ph[c].x, ph[c].y} + s1;
    if (c + 1 < NC) {          // (the last chunk is (bias, 0, 0, 0))
      s0 = pm_f2{w0[c].z, w0[c].w} * pm_f2{ph[c].z, ph[c].w} + s0;
      s1 = pm_f2{w1[c].z, w1[c].w} * pm_f2{ph[c].z, ph[c].w} + s1;
    }
  }
  return pm_f2{s0.x + s0.y, s1.x + s1.y};
}

// basis slot `k` (= lane & 31) of an edge at distance d with cutoff value fc / slope dfc:
//   k <  K : fc phi_k(d)                     |  d/dd: fc phi_k' + dfc phi_k
//   k == K : fc (the bias slot)              |  dfc
//   else 0                                                                       (nn/radial.py:11-15 gaussian, :105-110 bessel)
// Gaussian: exp through v_exp_f32 (exp2 of a non-positive argument, 1 ulp); Bessel: the accurate sincosf.
template <int K>
__device__ __forceinline__ void pm_basis(int kind, int k, float p0k, float p1k, float d, float fc, float dfc, float& b, float& db) {
  float phi, dphi;
  if (kind == SPK_RBF_GAUSSIAN) {
    const float c = -0.5f / (p1k * p1k);
    const float t = d - p0k;
    phi = __builtin_amdgcn_exp2f(1.4426950408889634f * c * t * t);
    dphi = 2.0f * c * t * phi;
  } else {
    float sn, co;
    sincosf(p0k * d, &sn, &co);
    if (d == 0.0f) { phi = sn; dphi = 0.f; }
    else { const float inv = 1.0f / d; phi = sn * inv; dphi = (p0k * co - phi) * inv; }
  }
  b = k < K ? fc * phi : (k == K ? fc : 0.f);
  db = k < K ? fc * dphi + dfc * phi : (k == K ? dfc : 0.f);
}

// P3: message (painn.py:43-66): a wavefront per centre atom (the atoms of the wave: sAsg), lane = channels 2 lane, 2 lane + 1; two
// edges of the row per step -- lanes 0..31 evaluate the radial basis of the first, lanes 32..63 of the second, the values reach
// all lanes through a 256-byte LDS slot of the wave (broadcast reads).  q is updated in place (no other atom's message reads
// it); the new mu rows stay in registers (rm) until every wave has read its neighbours' rows.
// Atoms of the group -> waves of the message phases (sAsg[wave][slot], -1 = none; run by ONE wave, lane = atom, deg = row length or
// -1 beyond the group).  Waves w and w + 4 share a SIMD and the phases are VALU-bound; per-wave stamps show the OLDER wave of a SIMD
// (w < 4) getting the larger share of the issue slots -- 2.2 k cycles per edge of the backward against 3.0 k for its partner -- and a
// wave left alone on its SIMD running at 1.6 k per edge, i.e. less than twice as fast: the phase is shortest when all eight waves
// finish together.  Rows are dealt longest first to the wave that would finish it earliest, (edges so far + row) x the cost per edge
// of that wave (22 : 30), at most 4 atoms per wave.  mode 1 = the snake over the waves used until round 3 (for A/B runs).
__device__ __forceinline__ void pm_assign_atoms(int deg, int na, int lane, int* __restrict__ sAsg, int mode) {
  int rank = 0;
  for (int b = 0; b < 32; ++b) {
    const int db = __builtin_amdgcn_readlane(deg, b);
    rank += (db > deg || (db == deg && b < lane)) ? 1 : 0;
  }
  if (lane < 32) sAsg[lane] = -1;
  if (mode == 0) {          // dynamic (default): sAsg = the atoms by falling row length; the waves take the next one when they are free
    if (lane < na) sAsg[rank] = lane;
    return;
  }
  if (mode == 1) {
    if (lane < na) {
      const int rnd = rank >> 3, pos = rank & 7;
      sAsg[((rnd & 1) ? 7 - pos : pos) * 4 + rnd] = lane;
    }
    return;
  }
  int myload = 0, mycnt = 0;          // lanes 0..7: edges (+ 1 per atom) and atoms given to wave `lane`
  const int cost = mode >= 20 ? mode : (lane < 4 ? 22 : 30);
  for (int r = 0; r < na; ++r) {
    const unsigned long long m = __ballot(rank == r && lane < na);
    const int at = (int)__ffsll((long long)m) - 1;
    const int d = __builtin_amdgcn_readlane(deg, at) + 1;
    const int key = (lane < 8 && mycnt < 4) ? (myload + d) * (lane < 4 ? 22 : cost) * 8 + lane : 0x7fffffff;
    int best = 0x7fffffff;
    for (int w = 0; w < 8; ++w) { const int kw = __builtin_amdgcn_readlane(key, w); best = kw < best ? kw : best; }
    const int w = best & 7;
    if (lane == w) { sAsg[w * 4 + mycnt] = at; mycnt += 1; myload += d; }
  }
}

// the next atom of a wave in a message phase: dynamic (asg = the 32-entry order, *ctr = atoms taken so far in this phase) or static
// (asg = the four slots of the wave)
__device__ __forceinline__ int pm_next_atom(const int* __restrict__ asg, int* __restrict__ ctr, int na, int it, int lane, bool dyn) {
  if (!dyn) return it < 4 ? asg[it] : -1;
  int r = 0;
  if (lane == 0) r = atomicAdd(ctr, 1);
  r = __builtin_amdgcn_readfirstlane(r);
  return r < na ? asg[r] : -1;
}

template <int K, bool MU0>
__device__ __forceinline__ void pm_message(const PmFilt<K>& Wf, const float* __restrict__ bf, float* __restrict__ sQ, const float* __restrict__ sMu,
                                           const float* __restrict__ sC, const PmEdge* __restrict__ sE, const int* __restrict__ sRow,
                                           const int* __restrict__ myAsg, float* __restrict__ myPhi, int rbf_kind, float p0k, float p1k, float cutoff,
                                           int lane, pm_f2 (&rm)[4][3], int (&ats)[4], int* __restrict__ ctr, int na, bool dyn) {
  constexpr bool mu0 = MU0;
  const int hi = lane >> 5;
  (void)bf;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int at = pm_next_atom(myAsg, ctr, na, it, lane, dyn);
    ats[it] = at;
    pm_f2 accq = {0.f, 0.f}, av0 = {0.f, 0.f}, av1 = {0.f, 0.f}, av2 = {0.f, 0.f};
    if (at >= 0) {
      const int rs = sRow[at], re = sRow[at + 1];
      for (int le = rs; le < re; le += 2) {
        const bool two = le + 1 < re;
        const PmEdge eA = sE[le], eB = sE[two ? le + 1 : le];
        const float fA = eA.d < cutoff ? eA.fc : 0.f, fB = (two && eB.d < cutoff) ? eB.fc : 0.f;
        if (fA == 0.f && fB == 0.f) continue;          // skin pairs contribute exactly zero
        {
          float bs, dbs;
          pm_basis<K>(rbf_kind, lane & 31, p0k, p1k, hi ? eB.d : eA.d, hi ? fB : fA, 0.f, bs, dbs);
          myPhi[lane] = bs;
        }
        const int jA = eA.jl * PM_LD + 2 * lane, jB = eB.jl * PM_LD + 2 * lane;
        f32x4 pa[K / 4 + 1], pb[K / 4 + 1];
#pragma unroll
        for (int c = 0; c < K / 4 + 1; ++c) { pa[c] = *(const f32x4*)(myPhi + 4 * c); pb[c] = *(const f32x4*)(myPhi + 32 + 4 * c); }
        {
          const pm_f2 FqA = pm_filter2<K / 4 + 1>(Wf.w[0][0], Wf.w[0][1], pa), FqB = pm_filter2<K / 4 + 1>(Wf.w[0][0], Wf.w[0][1], pb);
          accq += FqA * *(const pm_f2*)(sC + jA) + FqB * *(const pm_f2*)(sC + jB);
        }
        {
          const pm_f2 FRA = pm_filter2<K / 4 + 1>(Wf.w[1][0], Wf.w[1][1], pa), FRB = pm_filter2<K / 4 + 1>(Wf.w[1][0], Wf.w[1][1], pb);
          const pm_f2 mRA = FRA * *(const pm_f2*)(sC + PM_TILE + jA), mRB = FRB * *(const pm_f2*)(sC + PM_TILE + jB);
          av0 += mRA * eA.ux + mRB * eB.ux; av1 += mRA * eA.uy + mRB * eB.uy; av2 += mRA * eA.uz + mRB * eB.uz;
        }
        if (!mu0) {
          const pm_f2 FmA = pm_filter2<K / 4 + 1>(Wf.w[2][0], Wf.w[2][1], pa), FmB = pm_filter2<K / 4 + 1>(Wf.w[2][0], Wf.w[2][1], pb);
          const pm_f2 mmA = FmA * *(const pm_f2*)(sC + 2 * PM_TILE + jA), mmB = FmB * *(const pm_f2*)(sC + 2 * PM_TILE + jB);
          av0 += mmA * *(const pm_f2*)(sMu + jA) + mmB * *(const pm_f2*)(sMu + jB);
          av1 += mmA * *(const pm_f2*)(sMu + PM_TILE + jA) + mmB * *(const pm_f2*)(sMu + PM_TILE + jB);
          av2 += mmA * *(const pm_f2*)(sMu + 2 * PM_TILE + jA) + mmB * *(const pm_f2*)(sMu + 2 * PM_TILE + jB);
        }
      }
      const int io = at * PM_LD + 2 * lane;
      *(pm_f2*)(sQ + io) += accq;
      av0 += *(const pm_f2*)(sMu + io); av1 += *(const pm_f2*)(sMu + PM_TILE + io); av2 += *(const pm_f2*)(sMu + 2 * PM_TILE + io);
    }
    rm[it][0] = av0; rm[it][1] = av1; rm[it][2] = av2;
  }
}
__device__ __forceinline__ void pm_message_write(float* __restrict__ sMu, const int (&ats)[4], int lane, const pm_f2 (&rm)[4][3]) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int at = ats[it];
    if (at >= 0) {
      const int io = at * PM_LD + 2 * lane;
      *(pm_f2*)(sMu + io) = rm[it][0]; *(pm_f2*)(sMu + PM_TILE + io) = rm[it][1]; *(pm_f2*)(sMu + 2 * PM_TILE + io) = rm[it][2];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// P3 on the matrix core ("tiled" message; Gaussian bases; a group has at most 32 edge tiles = 16 per wave).  The filter of an edge is a contraction of n_rbf + 1 basis
// values with the filter rows -- per edge and channel 21 FMAs that the row form above issues as packed VALU math, 62 % of its
// instructions.  Here a TILE of 32 edge slots x 32 channels x 3 parts is three small GEMMs on v_mfma_f32_32x32x2_f32:
//   F_p[slot, ch] = sum_k  A[slot, k] B_p[k, ch],   A = f_c(d) phi_k(d) (k < n_rbf), f_c(d) (k = n_rbf: the bias slot), 0 beyond,
// A evaluated in registers by the lane that owns (slot = lane & 31, k = 2 kb + (lane >> 5)), B_p = the filter rows of the wave's
// 32-channel group (33 registers per interaction).  In the accumulator a lane holds ONE channel (lane & 31) of 16 slots, and the 16
// slots of a lane half belong to ONE centre atom (a tile = the rows of two atoms with <= 16 edges each, or one longer row split
// over the halves): the products with the neighbour rows (LDS, conflict-free: 32 consecutive channels) are summed over the edges
// IN THE LANE -- no lane-crossing reduction, no atomics, fixed order.  Wave (team, cg) works on channel group cg of the tiles
// team, team + 2, ...: the MFMAs of one wave of a SIMD were meant to run beside the VALU / LDS work of the other.
// MEASURED (round 3, cfg 3): correct -- and slower than the row form.  A tile costs 33 MFMAs (2.1 k cycles) + ~470 VALU / LDS
// instructions per wave; per SIMD 13 tiles take 52 k cycles = the SUM of the two, not their maximum (row form: 45 k): with two
// waves per SIMD in the same phase of the same loop the matrix pipe idles while both waves wait on the LDS round trips of the
// products (six per pair of slots before the reads were pipelined by hand, 60 k) and both queue for it afterwards.  Kept behind
// SPK_PM_TILED=1 (default off) with its parity tests; what it would need is a third and fourth wave per SIMD (128 registers
// each: the accumulators alone are 48) or the products themselves as a second GEMM.
#define PM_KB(K) (((K) + 2) / 2)
#define PM_TILE_INTS 256          // sPhi as ints: [0, 256) the tile list (32 tiles x 8), then as floats [256, 280) centres, [280, 304) exponents
typedef float pm_f8 __attribute__((ext_vector_type(8)));
template <int K>
struct PmTW { float b[3][PM_KB(K)]; };
template <int K>
__device__ __forceinline__ void pm_tw_load(PmTW<K>& W, const float* __restrict__ wf, const float* __restrict__ bf, int cg, int lane, bool mu0) {
  const int hi = lane >> 5, ch = cg * 32 + (lane & 31);
#pragma unroll
  for (int kb = 0; kb < PM_KB(K); ++kb) {
    const int k = 2 * kb + hi;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      float v = 0.f;
      if (!(mu0 && p == 2)) {
        if (k < K) v = wf[(size_t)(p * 128 + ch) * K + k];
        else if (k == K) v = bf[p * 128 + ch];
      }
      W.b[p][kb] = v;
    }
  }
}
// tiles of a group (run by one wave, lane = atom): rows of more than 16 edges own a tile (first 16 edges in half A, the rest in half
// B), the others are paired in atom order.  sTile[8 u + ...] = atom A, first edge A, edges A, atom B, first edge B, edges B.
// Also the table of the Gaussian basis: centres and -log2(e) / (2 width^2) per k (0 beyond n_rbf).  Returns the number of tiles.
template <int K>
__device__ __forceinline__ int pm_build_tiles(const int* __restrict__ sRow, int na, int lane, const RadialDev& rb, int* __restrict__ sTile) {
  const int rs = lane < na ? sRow[lane] : 0, dg = lane < na ? sRow[lane + 1] - rs : 0;
  const bool lng = dg > 16, sht = dg > 0 && !lng;
  const unsigned long long ml = __ballot(lng), ms = __ballot(sht);
  const unsigned long long below = (1ull << lane) - 1ull;
  const int n_long = __popcll(ml), n_short = __popcll(ms);
  if (lng) {
    int* T = sTile + 8 * __popcll(ml & below);
    T[0] = lane; T[1] = rs; T[2] = 16; T[3] = lane; T[4] = rs + 16; T[5] = dg - 16;
  } else if (sht) {
    const int r = __popcll(ms & below);
    int* T = sTile + 8 * (n_long + (r >> 1));
    if (r & 1) { T[3] = lane; T[4] = rs; T[5] = dg; }
    else {
      T[0] = lane; T[1] = rs; T[2] = dg;
      if (r == n_short - 1) { T[3] = lane; T[4] = rs; T[5] = 0; }          // (no partner: an empty half B)
    }
  }
  if (lane < 24) {
    float* tb = (float*)(sTile + PM_TILE_INTS);
    const float wd = (lane < K && rb.p1) ? rb.p1[lane] : 1.f;
    tb[lane] = (lane < K && rb.p0) ? rb.p0[lane] : 0.f;
    tb[24 + lane] = lane < K ? -0.5f * 1.4426950408889634f / (wd * wd) : 0.f;
  }
  return n_long + ((n_short + 1) >> 1);
}
// edge records of the tiled form: arrays of PM_ME entries each (structure of arrays: the values of two consecutive slots land in
// adjacent registers = operands of packed math), 16 zero records behind the last edge (slots beyond a row's end read the following
// rows' records -- finite values times a filter that is exactly 0)
#define PM_ME (PM_MAXEDGES + 16)
struct PmEdgeT { const int* jo; const float *ux, *uy, *uz, *d, *fc; };
__device__ __forceinline__ PmEdgeT pm_edge_arrays(const void* base) {
  const float* f = (const float*)base;
  return PmEdgeT{(const int*)f, f + PM_ME, f + 2 * PM_ME, f + 3 * PM_ME, f + 4 * PM_ME, f + 5 * PM_ME};
}
template <int K, bool MU0>
__device__ __forceinline__ void pm_message_tiled(const PmTW<K>& W, float* __restrict__ sQ, const float* __restrict__ sMu, const float* __restrict__ sC,
                                                 const PmEdgeT E, const int* __restrict__ sTile, int nT, int team, int cg,
                                                 float cutoff, int lane, pm_f8 (&dm)[3], float* __restrict__ ovf /* [na, 3, 128] global: slots beyond 8 */) {
  constexpr int KB = PM_KB(K);
  const int hi = lane >> 5, m = lane & 31;
  const int hA = (m >> 2) & 1, rA = (m & 3) + 4 * (m >> 3);      // the slot this lane evaluates the basis of: (half, index in the half)
  const int ch = cg * 32 + m;
  const float* tb = (const float*)(sTile + PM_TILE_INTS) + hi;
  const pm_f2 z2 = {0.f, 0.f};
  const unsigned chb = 4u * (unsigned)ch;
  const unsigned c_base = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)sC;
  const unsigned mu_base = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)sMu;
  int s = 0;
#pragma nounroll
  for (int u = team; u < nT; u += 2, ++s) {
    const int* T = sTile + 8 * u;
    const int iA = T[0], fA = T[1], nA = T[2], iB = T[3], fB = T[4], nB = T[5];
    // ---- A operand (Gaussian bases; the Bessel models keep the row form: eleven inlined sines per instance)
    float A[KB];
    {
      const int cnt = hA ? nB : nA;
      const bool valid = rA < cnt;
      const int e = (hA ? fB : fA) + rA;
      const float d = E.d[e];
      const float fc = (valid && d < cutoff) ? E.fc[e] : 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int k = 2 * kb + hi;
        const float tt = d - tb[2 * kb];
        const float g = fc * __builtin_amdgcn_exp2f(tb[24 + 2 * kb] * tt * tt);
        A[kb] = (2 * kb + 1 < K || k < K) ? g : (k == K ? fc : 0.f);
      }
    }
    // ---- the three filters of the tile for this wave's channel group
    f32x16 F0, F1, F2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { F0[r] = 0.f; F1[r] = 0.f; F2[r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      F0 = PM_MFMA(A[kb], W.b[0][kb], F0);
      F1 = PM_MFMA(A[kb], W.b[1][kb], F1);
      if (!MU0) F2 = PM_MFMA(A[kb], W.b[2][kb], F2);
    }
    // ---- products with the neighbour rows, summed over the 16 slots of the lane's half (= one centre atom), two slots per step.
    // Software-pipelined by hand: the LDS reads of step p + 1 are issued before the arithmetic of step p and PINNED there (left
    // alone the compiler puts every read right in front of its use: six exposed LDS round trips per step, 6 k cycles per tile --
    // three times the MFMA time of the tile)
    const int fh = hi ? fB : fA;
    // (explicit 32-bit LDS addresses: row offset of the neighbour [bytes, from the edge record] + channel + plane base, the three
    //  planes of mu resp. c as immediate offsets of the reads -- as generic pointer arithmetic every read cost two or three adds)
    typedef const __attribute__((address_space(3))) float* lds_cf;
    unsigned jo[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) jo[r] = (unsigned)E.jo[fh + r] * 4u + chb;
    pm_f2 q2 = z2, m0 = z2, m1 = z2, m2 = z2;
    struct Stage { pm_f2 ux, uy, uz, cq, cR, cM, ma, mb, mc; };
    auto issue = [&](int r, Stage& S) {
      const int e = fh + r;
      S.ux = pm_f2{E.ux[e], E.ux[e + 1]}; S.uy = pm_f2{E.uy[e], E.uy[e + 1]}; S.uz = pm_f2{E.uz[e], E.uz[e + 1]};
      lds_cf c0 = (lds_cf)(size_t)(jo[r] + c_base), c1 = (lds_cf)(size_t)(jo[r + 1] + c_base);
      S.cq = pm_f2{c0[0], c1[0]}; S.cR = pm_f2{c0[PM_TILE], c1[PM_TILE]};
      if (!MU0) {
        lds_cf u0 = (lds_cf)(size_t)(jo[r] + mu_base), u1 = (lds_cf)(size_t)(jo[r + 1] + mu_base);
        S.cM = pm_f2{c0[2 * PM_TILE], c1[2 * PM_TILE]};
        S.ma = pm_f2{u0[0], u1[0]}; S.mb = pm_f2{u0[PM_TILE], u1[PM_TILE]}; S.mc = pm_f2{u0[2 * PM_TILE], u1[2 * PM_TILE]};
      }
    };
    auto compute = [&](int r, const Stage& S) {
      q2 += pm_f2{F0[r], F0[r + 1]} * S.cq;
      const pm_f2 tR = pm_f2{F1[r], F1[r + 1]} * S.cR;
      m0 += tR * S.ux; m1 += tR * S.uy; m2 += tR * S.uz;
      if (!MU0) {
        const pm_f2 tM = pm_f2{F2[r], F2[r + 1]} * S.cM;
        m0 += tM * S.ma; m1 += tM * S.mb; m2 += tM * S.mc;
      }
    };
    Stage S0, S1;
    issue(0, S0);
#pragma unroll
    for (int r = 0; r < 16; r += 4) {
      issue(r + 2, S1);
      asm volatile("" ::: "memory");
      compute(r, S0);
      if (r + 4 < 16) issue(r + 4, S0);
      asm volatile("" ::: "memory");
      compute(r + 2, S1);
    }
    float dq = q2.x + q2.y, d0 = m0.x + m0.y, d1 = m1.x + m1.y, d2 = m2.x + m2.y;
    const bool longrow = (iA == iB) && nB > 0;
    if (longrow) { dq += __shfl_xor(dq, 32, 64); d0 += __shfl_xor(d0, 32, 64); d1 += __shfl_xor(d1, 32, 64); d2 += __shfl_xor(d2, 32, 64); }
    const bool owner = hi ? (nB > 0 && !longrow) : (nA > 0);
    const int ic = hi ? iB : iA;
    if (owner) sQ[ic * PM_LD + ch] += dq;         // q in place: nobody's message reads q
    // the new mu rows wait for the barrier (the old ones are still being read): slot s of the wave, selected without indexed
    // registers; groups of more than 16 tiles park the slots beyond 8 in global memory
    if (s < 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const bool me = (s == j); dm[0][j] = me ? d0 : dm[0][j]; dm[1][j] = me ? d1 : dm[1][j]; dm[2][j] = me ? d2 : dm[2][j]; }
    } else if (owner) {
      float* o = ovf + (size_t)ic * 384 + ch;
      o[0] = d0; o[128] = d1; o[256] = d2;
    }
  }
}
__device__ __forceinline__ void pm_message_tiled_write(float* __restrict__ sMu, const int* __restrict__ sTile, int nT, int team, int cg, int lane, const pm_f8 (&dm)[3],
                                                       const float* __restrict__ ovf) {
  const int hi = lane >> 5, ch = cg * 32 + (lane & 31);
  int s = 0;
#pragma nounroll
  for (int u = team; u < nT; u += 2, ++s) {
    const int* T = sTile + 8 * u;
    const int iA = T[0], nA = T[2], iB = T[3], nB = T[5];
    const bool longrow = (iA == iB) && nB > 0;
    const bool owner = hi ? (nB > 0 && !longrow) : (nA > 0);
    if (owner) {
      const int ic = hi ? iB : iA;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      if (s < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const bool me = (s == j); v0 = me ? dm[0][j] : v0; v1 = me ? dm[1][j] : v1; v2 = me ? dm[2][j] : v2; }
      } else {
        const float* o = ovf + (size_t)ic * 384 + ch;
        v0 = o[0]; v1 = o[128]; v2 = o[256];
      }
      const int io = ic * PM_LD + ch;
      sMu[io] += v0; sMu[PM_TILE + io] += v1; sMu[2 * PM_TILE + io] += v2;
    }
  }
}

// Every weight tile is requested one step AHEAD of its use -- right after the MFMAs of the previous tile and BEFORE that tile's
// epilogue: loads and stores share one in-order counter on this architecture, so a load issued behind the saved-tensor stores of
// an epilogue could not be waited for before those stores were acknowledged.
// The two teams of four waves (wave t of a team = SIMD t) run DIFFERENT code paths with the same sequence of barriers: the register
// allocation of a path then only sees what that team keeps alive (team 0: V / W / sum V W across P4-P6; team 1: the K = 256 tile).
template <int K, bool TILED, bool POT, bool SP>      // K = n_rbf (a multiple of 4, <= PM_NRBF): the register-resident filter weights are indexed statically; TILED: message on the matrix core (Gaussian bases); POT: the standard potential; SP: Dense phases on the split-precision matrix path (split weight images)
__global__ __launch_bounds__(512) void k_painn_mol_fwd(PmFwdArgs a) {
  constexpr int F = 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sQ = smem;                          // [32][LD]     q
  float* sMu = sQ + PM_TILE;                 // [3][32][LD]  mu, component planes
  float* sC = sMu + 3 * PM_TILE;             // [3][32][LD]  c planes (q | R | mu part) during P2-P3; plane 0 = |V| during P4-P5
  float* sH = sC + 3 * PM_TILE;              // [32][LD]     hidden layer of the two context nets
  PmEdge* sE = (PmEdge*)(sH + PM_TILE);      // [PM_MAXEDGES]
  float* sPhi = (float*)(sE + PM_MAXEDGES + 16);  // [8 waves][64] radial basis of the two edges a wave is working on (tiled form: tile list + basis table)
  int* sRow = (int*)(sPhi + 8 * 64);         // [33] local CSR
  int* sAsg = sRow + 36;                     // [8 waves][4] atoms of a wave in the message phase (balanced by row length), -1 = none
  float* sN = sC;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, el = lane & 31;
  const int team = wv >> 2, t = wv & 3;
  const float p0k = (el < K && a.rb.p0) ? a.rb.p0[el] : 0.f;      // lanes 0..31 and 32..63: the basis of edge A resp. B
  const float p1k = (el < K && a.rb.p1) ? a.rb.p1[el] : 1.f;
  const float cutoff = a.rb.cutoff;
  const int64_t nf = a.N * (int64_t)F;
  const int64_t per = 17 * nf;
  float* myPhi = sPhi + wv * 64;
  const int* myAsg = sAsg + wv * 4;
  const bool dyn = (a.assign == 0);

  for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
    const int a0 = a.grp_atom0[grp], na = a.grp_atom0[grp + 1] - a0;
    const int e0 = a.rowptr[a0], ne = a.rowptr[a0 + na] - e0;
    PM_BARRIER();      // the previous group is done with every LDS buffer
    PM_STAMP(0);

    // ---- group set-up: q rows, mu = 0 (painn.py:246), local CSR, edge geometry (shared by all interactions)
    for (int s = tid; s < 32 * 32; s += 512) {
      const int row = s >> 5, c4 = s & 31;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < na) {
        if (POT && !a.q0) {
          const int64_t z = a.Z[a0 + row];
          if (z >= 0 && z < a.n_types) v = *(const f32x4*)(a.emb + (size_t)z * F + 4 * c4);
          else { const float qn = __builtin_nanf(""); v = f32x4{qn, qn, qn, qn}; }       // no row to read: loud, never out of bounds
        } else v = pm_ld<f32x4>(a.q0 + (size_t)a0 * F, (unsigned)(s * 16));
      }
      *(f32x4*)(sQ + row * PM_LD + 4 * c4) = v;
    }
    for (int s = tid; s < 3 * PM_TILE / 4; s += 512) *(f32x4*)(sMu + 4 * s) = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int le = tid; le < ne + (TILED ? 16 : 0); le += 512) {
      PmEdge ed = {0, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (le < ne) {
        const int64_t e = (int64_t)e0 + le;
        float rx, ry, rz;
        if (POT) {          // r_ij = R_j - R_i (+ offset): atomistic/distances.py:14-26
          const int64_t i3 = 3 * a.idx_i[e], j3 = 3 * a.idx_j[e];
          rx = a.R[j3] - a.R[i3]; ry = a.R[j3 + 1] - a.R[i3 + 1]; rz = a.R[j3 + 2] - a.R[i3 + 2];
          if (a.offsets) { rx += a.offsets[3 * e]; ry += a.offsets[3 * e + 1]; rz += a.offsets[3 * e + 2]; }
          a.rij_out[3 * e] = rx; a.rij_out[3 * e + 1] = ry; a.rij_out[3 * e + 2] = rz;
        } else { rx = a.rij[3 * e]; ry = a.rij[3 * e + 1]; rz = a.rij[3 * e + 2]; }
        ed.d = sqrtf(rx * rx + ry * ry + rz * rz);
        const float inv = 1.0f / ed.d;
        ed.ux = rx * inv; ed.uy = ry * inv; ed.uz = rz * inv;
        float dfc;
        spk_cutoff_eval(cutoff, ed.d, ed.fc, dfc);
        ed.jl = (int)(a.idx_j[e] - a0);
      }
      if (TILED) {          // structure of arrays, the neighbour as its row offset (pm_message_tiled)
        float* f = (float*)sE;
        ((int*)f)[le] = ed.jl * PM_LD; f[PM_ME + le] = ed.ux; f[2 * PM_ME + le] = ed.uy; f[3 * PM_ME + le] = ed.uz; f[4 * PM_ME + le] = ed.d; f[5 * PM_ME + le] = ed.fc;
      } else sE[le] = ed;
    }
    if (wv == 7) {
      // local CSR + the atoms of every wave in the message phase (pm_assign_atoms: edges balanced per SIMD)
      int r0 = 0;
      if (lane <= 32) r0 = a.rowptr[a0 + (lane < na ? lane : na)] - e0;
      const int r1 = __shfl_down(r0, 1, 64);
      if (lane <= 32) sRow[lane] = r0;
      pm_assign_atoms(lane < na ? r1 - r0 : -1, na, lane, sAsg, a.assign);
      if (lane == 0) { sRow[34] = 0; sRow[35] = 0; }
      if (TILED) { const int n = pm_build_tiles<K>(sRow, na, lane, a.rb, (int*)sPhi); if (lane == 0) sRow[33] = n; }      // (the row form uses sPhi per edge; the two are exclusive)
    }
    // mu entering the first interaction is zero: the backward reads it from the saved block
    for (int s = tid; s < na * 96; s += 512) pm_st<f32x4>(a.saved + 4 * nf + (size_t)a0 * 3 * F, (unsigned)(s * 16), f32x4{0.f, 0.f, 0.f, 0.f});

    if (team == 0) {
      // ======================================================================== team 0
      PmW Wt;            // the weight tile this wave needs next
      f32x16 bz;         // ... and its bias
      pm_wload(Wt, a.L[0].ctx1_p, 16, t, 0, lane); bz = pm_bias_acc(a.L[0].ctx1_b, t, hi);
      for (int l = 0; l < a.n_layers; ++l) {
        // (lane-derived LDS / global offsets are re-derived inside every interaction: hoisted out of the loop by the compiler they
        //  fill the prologue with dozens of live registers -- and their spills)
        int lane_o_ = lane, tid_o_ = tid;
        asm volatile("" : "+v"(lane_o_), "+v"(tid_o_));
        const int lane = lane_o_, tid = tid_o_, hi = lane >> 5, el = lane & 31;
        (void)tid; (void)hi; (void)el;
        const PmLayerDev& P = a.L[l];
        float* S = a.saved + (int64_t)l * per;
        const bool last = (l + 1 == a.n_layers);
        float* mu_next_g = last ? a.mu_out : (a.saved + (int64_t)(l + 1) * per + 4 * nf);
        PM_BARRIER();
        PM_STAMP(1 + 8 * l);
        const int nT = sRow[33];
        constexpr bool tiled = TILED;
        const int* sTile = (const int*)sPhi;
        // ---- P1: pre_a = W_a1 q + b (saved), silu -> sH
        {
          f32x16 acc = pm_wmma<SP>(Wt, sQ, lane, bz);
          pm_wload(Wt, P.ctx2_p, 16, t, 0, lane); bz = pm_bias_acc(P.ctx2_b, t, hi);
          float* preA_g = S;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 pv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *(f32x4*)(sH + el * PM_LD + 32 * t + 8 * q + 4 * hi) = f32x4{pm_silu(pv.x), pm_silu(pv.y), pm_silu(pv.z), pm_silu(pv.w)};
            if (el < na) pm_st<f32x4>(preA_g + (size_t)a0 * F + 32 * t, (unsigned)((el * F + 8 * q + 4 * hi) * 4), pv);
          }
        }
        PM_BARRIER();
        PM_STAMP(2 + 8 * l);
        // ---- P2: c = W_a2 silu(pre_a) + b, tiles t and 8 + t (saved, LDS planes)
        PmFilt<K> Wf;
        PmTW<K> TW;
        {
          float* c_g = S + nf;
          f32x16 acc = pm_wmma<SP>(Wt, sH, lane, bz);
          pm_wload(Wt, P.ctx2_p, 16, 8 + t, 0, lane); bz = pm_bias_acc(P.ctx2_b, 8 + t, hi);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 cv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *(f32x4*)(sC + el * PM_LD + 32 * t + 8 * q + 4 * hi) = cv;
            if (el < na) pm_st<f32x4>(c_g + (size_t)a0 * 3 * F + 32 * t, (unsigned)((el * 3 * F + 8 * q + 4 * hi) * 4), cv);
          }
          if (tiled) pm_tw_load<K>(TW, P.wf, P.bf, t, lane, l == 0);
          else pm_filt_load<K>(Wf, P.wf, P.bf, lane, l == 0);
          acc = pm_wmma<SP>(Wt, sH, lane, bz);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 cv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *(f32x4*)(sC + 2 * PM_TILE + el * PM_LD + 32 * t + 8 * q + 4 * hi) = cv;
            if (el < na) pm_st<f32x4>(c_g + (size_t)a0 * 3 * F + 32 * (8 + t), (unsigned)((el * 3 * F + 8 * q + 4 * hi) * 4), cv);
          }
        }
        PM_BARRIER();
        PM_STAMP(3 + 8 * l);
        // ---- P3: message
        if (tiled) {
          pm_f8 dm[3];
          float* ovf = a.mu_out + (size_t)a0 * 3 * F;          // (written for good in P6 of the last interaction: free until then)
          if (l == 0) pm_message_tiled<K, true>(TW, sQ, sMu, sC, pm_edge_arrays(sE), sTile, nT, 0, t, cutoff, lane, dm, ovf);
          else pm_message_tiled<K, false>(TW, sQ, sMu, sC, pm_edge_arrays(sE), sTile, nT, 0, t, cutoff, lane, dm, ovf);
          pm_wload(Wt, P.mix_p, 16, t, 0, lane);
          PM_STAMP(4 + 8 * l);
          PM_BARRIER();       // every wave has read its neighbours' rows: mu can be replaced
          pm_message_tiled_write(sMu, sTile, nT, 0, t, lane, dm, ovf);
        } else {
          pm_f2 rm[4][3];
          int ats[4];
          int* ctr = sRow + 34 + (l & 1);
          if (tid == 0) sRow[34 + ((l + 1) & 1)] = 0;          // (the counter of the next interaction; everybody left it a barrier ago)
          if (l == 0) pm_message<K, true>(Wf, P.bf, sQ, sMu, sC, sE, sRow, dyn ? sAsg : myAsg, myPhi, a.rb.kind, p0k, p1k, cutoff, lane, rm, ats, ctr, na, dyn);
          else pm_message<K, false>(Wf, P.bf, sQ, sMu, sC, sE, sRow, dyn ? sAsg : myAsg, myPhi, a.rb.kind, p0k, p1k, cutoff, lane, rm, ats, ctr, na, dyn);
          pm_wload(Wt, P.mix_p, 16, t, 0, lane);
          PM_STAMP(4 + 8 * l);
          PM_BARRIER();       // every wave has read its neighbours' rows: mu can be replaced
          pm_message_write(sMu, ats, lane, rm);
        }
        PM_BARRIER();
        PM_STAMP(5 + 8 * l);
        // ---- P4: V = mu W_mix^T[:F] for the three components (feature tile t); |V| -> LDS (read by team 1 in P5); V saved
        f32x16 W0, W1, W2, sVW;
        f32x16 V0, V1, V2;
        {
#pragma unroll
          for (int r = 0; r < 16; ++r) { V0[r] = 0.f; V1[r] = 0.f; V2[r] = 0.f; W0[r] = 0.f; W1[r] = 0.f; W2[r] = 0.f; }
          pm_wmma3<SP>(Wt, sMu, lane, V0, V1, V2);
          pm_wload(Wt, P.mix_p, 16, 4 + t, 0, lane);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 nv;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int r = 4 * q + v;
              nv[v] = sqrtf(V0[r] * V0[r] + V1[r] * V1[r] + V2[r] * V2[r] + a.eps);
            }
            *(f32x4*)(sN + el * PM_LD + 32 * t + 8 * q + 4 * hi) = nv;
          }
        }
        PM_BARRIER();
        PM_STAMP(6 + 8 * l);
        // ---- P5: W = mu W_mix^T[F:] (beside team 1's pre_b on the same SIMD); sum_x V W; mix saved
        {
          pm_wmma3<SP>(Wt, sMu, lane, W0, W1, W2);
          pm_wload(Wt, P.ic2_p, 16, 8 + t, 0, lane); bz = pm_bias_acc(P.ic2_b, 8 + t, hi);
#pragma unroll
          for (int r = 0; r < 16; ++r) sVW[r] = V0[r] * W0[r] + V1[r] * W1[r] + V2[r] * W2[r];
          if (el < na) {
            float* mg = S + 7 * nf + (size_t)a0 * 6 * F + 32 * t;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned o = (unsigned)((el * 6 * F + 8 * q + 4 * hi) * 4);
              pm_st<f32x4>(mg, o, f32x4{V0[4 * q], V0[4 * q + 1], V0[4 * q + 2], V0[4 * q + 3]});
              pm_st<f32x4>(mg, o + 2 * F * 4, f32x4{V1[4 * q], V1[4 * q + 1], V1[4 * q + 2], V1[4 * q + 3]});
              pm_st<f32x4>(mg, o + 4 * F * 4, f32x4{V2[4 * q], V2[4 * q + 1], V2[4 * q + 2], V2[4 * q + 3]});
              pm_st<f32x4>(mg, o + F * 4, f32x4{W0[4 * q], W0[4 * q + 1], W0[4 * q + 2], W0[4 * q + 3]});
              pm_st<f32x4>(mg, o + 3 * F * 4, f32x4{W1[4 * q], W1[4 * q + 1], W1[4 * q + 2], W1[4 * q + 3]});
              pm_st<f32x4>(mg, o + 5 * F * 4, f32x4{W2[4 * q], W2[4 * q + 1], W2[4 * q + 2], W2[4 * q + 3]});
            }
          }
        }
        PM_BARRIER();
        PM_STAMP(7 + 8 * l);
        // ---- P6: a = W_b2 silu(pre_b) + b (saved);  q += a_q + a_qmu sum_x V W;  mu += a_mu W      (painn.py:110-116)
        {
          float* ag = S + 14 * nf + (size_t)a0 * 3 * F + 32 * t;
          f32x16 tq = pm_wmma<SP>(Wt, sH, lane, bz);                       // a_qmu
          pm_wload(Wt, P.ic2_p, 16, t, 0, lane); bz = pm_bias_acc(P.ic2_b, t, hi);
          if (el < na) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              pm_st<f32x4>(ag, (unsigned)((el * 3 * F + 2 * F + 8 * q + 4 * hi) * 4), f32x4{tq[4 * q], tq[4 * q + 1], tq[4 * q + 2], tq[4 * q + 3]});
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) tq[r] *= sVW[r];
          f32x16 aq = pm_wmma<SP>(Wt, sH, lane, bz);                       // a_q
          pm_wload(Wt, P.ic2_p, 16, 4 + t, 0, lane); bz = pm_bias_acc(P.ic2_b, 4 + t, hi);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float* qp = sQ + el * PM_LD + 32 * t + 8 * q + 4 * hi;
            f32x4 qv = *(const f32x4*)qp;
            const f32x4 av = {aq[4 * q], aq[4 * q + 1], aq[4 * q + 2], aq[4 * q + 3]};
            qv.x += av.x + tq[4 * q]; qv.y += av.y + tq[4 * q + 1]; qv.z += av.z + tq[4 * q + 2]; qv.w += av.w + tq[4 * q + 3];
            if (el >= na) qv = f32x4{0.f, 0.f, 0.f, 0.f};
            *(f32x4*)qp = qv;
            if (el < na) {
              pm_st<f32x4>(ag, (unsigned)((el * 3 * F + 8 * q + 4 * hi) * 4), av);
              if (last) pm_st<f32x4>(a.q_out + (size_t)a0 * F + 32 * t, (unsigned)((el * F + 8 * q + 4 * hi) * 4), qv);
            }
          }
          f32x16 am = pm_wmma<SP>(Wt, sH, lane, bz);                       // a_mu
          if (!last) { pm_wload(Wt, a.L[l + 1].ctx1_p, 16, t, 0, lane); bz = pm_bias_acc(a.L[l + 1].ctx1_b, t, hi); }
          float* mn = mu_next_g + (size_t)a0 * 3 * F + 32 * t;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 av = {am[4 * q], am[4 * q + 1], am[4 * q + 2], am[4 * q + 3]};
            float* mp = sMu + el * PM_LD + 32 * t + 8 * q + 4 * hi;
            f32x4 m0 = *(const f32x4*)mp, m1 = *(const f32x4*)(mp + PM_TILE), m2 = *(const f32x4*)(mp + 2 * PM_TILE);
            m0.x += av.x * W0[4 * q]; m0.y += av.y * W0[4 * q + 1]; m0.z += av.z * W0[4 * q + 2]; m0.w += av.w * W0[4 * q + 3];
            m1.x += av.x * W1[4 * q]; m1.y += av.y * W1[4 * q + 1]; m1.z += av.z * W1[4 * q + 2]; m1.w += av.w * W1[4 * q + 3];
            m2.x += av.x * W2[4 * q]; m2.y += av.y * W2[4 * q + 1]; m2.z += av.z * W2[4 * q + 2]; m2.w += av.w * W2[4 * q + 3];
            if (el >= na) { m0 = f32x4{0.f, 0.f, 0.f, 0.f}; m1 = m0; m2 = m0; }
            *(f32x4*)mp = m0; *(f32x4*)(mp + PM_TILE) = m1; *(f32x4*)(mp + 2 * PM_TILE) = m2;
            if (el < na) {
              const unsigned o = (unsigned)((el * 3 * F + 8 * q + 4 * hi) * 4);
              pm_st<f32x4>(ag, (unsigned)((el * 3 * F + F + 8 * q + 4 * hi) * 4), av);
              pm_st<f32x4>(mn, o, m0);
              pm_st<f32x4>(mn, o + F * 4, m1);
              pm_st<f32x4>(mn, o + 2 * F * 4, m2);
            }
          }
        }
        PM_STAMP(8 + 8 * l);
        // (the barrier at the top of the next interaction / group closes this phase)
      }
    } else {
      // ======================================================================== team 1
      for (int l = 0; l < a.n_layers; ++l) {
        // (lane-derived LDS / global offsets are re-derived inside every interaction: hoisted out of the loop by the compiler they
        //  fill the prologue with dozens of live registers -- and their spills)
        int lane_o_ = lane, tid_o_ = tid;
        asm volatile("" : "+v"(lane_o_), "+v"(tid_o_));
        const int lane = lane_o_, tid = tid_o_, hi = lane >> 5, el = lane & 31;
        (void)tid; (void)hi; (void)el;
        const PmLayerDev& P = a.L[l];
        float* S = a.saved + (int64_t)l * per;
        PmW Wt;
        f32x16 bz;
        PM_BARRIER();
        const int nT = sRow[33];
        constexpr bool tiled = TILED;
        const int* sTile = (const int*)sPhi;
        // ---- P1: (team 0: pre_a) -- request the tile of P2
        pm_wload(Wt, P.ctx2_p, 16, 4 + t, 0, lane); bz = pm_bias_acc(P.ctx2_b, 4 + t, hi);
        PM_BARRIER();
        // ---- P2: c tile 4 + t (the R part)
        PmFilt<K> Wf;
        PmTW<K> TW;
        {
          float* c_g = S + nf;
          if (tiled) pm_tw_load<K>(TW, P.wf, P.bf, t, lane, l == 0);
          else pm_filt_load<K>(Wf, P.wf, P.bf, lane, l == 0);          // (requested before the MFMAs of the tile: they arrive meanwhile)
          f32x16 acc = pm_wmma<SP>(Wt, sH, lane, bz);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 cv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *(f32x4*)(sC + PM_TILE + el * PM_LD + 32 * t + 8 * q + 4 * hi) = cv;
            if (el < na) pm_st<f32x4>(c_g + (size_t)a0 * 3 * F + 32 * (4 + t), (unsigned)((el * 3 * F + 8 * q + 4 * hi) * 4), cv);
          }
        }
        PM_BARRIER();
        // ---- P3: message
        if (tiled) {
          pm_f8 dm[3];
          float* ovf = a.mu_out + (size_t)a0 * 3 * F;          // (written for good in P6 of the last interaction: free until then)
          if (l == 0) pm_message_tiled<K, true>(TW, sQ, sMu, sC, pm_edge_arrays(sE), sTile, nT, 1, t, cutoff, lane, dm, ovf);
          else pm_message_tiled<K, false>(TW, sQ, sMu, sC, pm_edge_arrays(sE), sTile, nT, 1, t, cutoff, lane, dm, ovf);
          PM_BARRIER();
          pm_message_tiled_write(sMu, sTile, nT, 1, t, lane, dm, ovf);
        } else {
          pm_f2 rm[4][3];
          int ats[4];
          int* ctr = sRow + 34 + (l & 1);
          if (tid == 0) sRow[34 + ((l + 1) & 1)] = 0;          // (the counter of the next interaction; everybody left it a barrier ago)
          if (l == 0) pm_message<K, true>(Wf, P.bf, sQ, sMu, sC, sE, sRow, dyn ? sAsg : myAsg, myPhi, a.rb.kind, p0k, p1k, cutoff, lane, rm, ats, ctr, na, dyn);
          else pm_message<K, false>(Wf, P.bf, sQ, sMu, sC, sE, sRow, dyn ? sAsg : myAsg, myPhi, a.rb.kind, p0k, p1k, cutoff, lane, rm, ats, ctr, na, dyn);
          PM_BARRIER();
          pm_message_write(sMu, ats, lane, rm);
        }
        PM_BARRIER();
        // ---- P4: (team 0: channel mix) -- request the K = 256 tile of P5
        PmW Wu;
        pm_wload(Wt, P.ic1_p, 32, t, 0, lane);
        pm_wload(Wu, P.ic1_p, 32, t, 16, lane);
        bz = pm_bias_acc(P.ic1_b, t, hi);
        PM_BARRIER();
        // ---- P5: pre_b = W_b1 [q | |V|] + b (K = 256; saved), silu -> sH
        {
          float* preB_g = S + 13 * nf;
          f32x16 acc = pm_wmma<SP>(Wt, sQ, lane, bz);
          acc = pm_wmma<SP>(Wu, sN, lane, acc);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 pv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *(f32x4*)(sH + el * PM_LD + 32 * t + 8 * q + 4 * hi) = f32x4{pm_silu(pv.x), pm_silu(pv.y), pm_silu(pv.z), pm_silu(pv.w)};
            if (el < na) pm_st<f32x4>(preB_g + (size_t)a0 * F + 32 * t, (unsigned)((el * F + 8 * q + 4 * hi) * 4), pv);
          }
        }
        PM_BARRIER();
        // ---- P6: (team 0: update)
      }
    }
    if (POT) {
      // ================= energy head on the atom tile that is still in LDS: y = w2 . act(W1 q + b1) + b2, E[mol] += sum_atoms y
      const PmHeadDev& Hd = a.head;
      long long my_mol = -1;
      if (wv == 1 && lane < 32) my_mol = lane < na ? Hd.idx_m[a0 + lane] : -2;      // wave 1: molecule id of atom `lane`
      PM_BARRIER();                    // q_L is complete
      if (wv < 2) {                    // H = 64: two hidden tiles
        f32x4 av[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) av[u] = pm_ld<f32x4>(Hd.w1 + (size_t)(32 * wv) * F, (unsigned)((el * F + 8 * u + 4 * hi) * 4));
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = Hd.b1[32 * wv + pm_row(r, hi)];
        const float* brow = sQ + el * PM_LD + 4 * hi;
        {
          f32x4 a0v[8], a1v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { a0v[u] = av[u]; a1v[u] = av[8 + u]; }
          acc = pm_mma8(a0v, brow, acc);
          acc = pm_mma8(a1v, brow + 64, acc);
        }
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 pv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          if (el < na) pm_st<f32x4>(Hd.pre_h + (size_t)a0 * 64 + 32 * wv, (unsigned)((el * 64 + 8 * q + 4 * hi) * 4), pv);
          const f32x4 wv2 = *(const f32x4*)(Hd.w2 + 32 * wv + 8 * q + 4 * hi);
          if (Hd.act == SPK_ACT_SILU) part += pm_silu(pv.x) * wv2.x + pm_silu(pv.y) * wv2.y + pm_silu(pv.z) * wv2.z + pm_silu(pv.w) * wv2.w;
          else part += spk_ssp(pv.x) * wv2.x + spk_ssp(pv.y) * wv2.y + spk_ssp(pv.z) * wv2.z + spk_ssp(pv.w) * wv2.w;
        }
        part += __shfl_xor(part, 32, 64);
        if (hi == 0) sH[wv * 32 + el] = part;          // (sH: the hidden tile of the last context net is no longer needed)
        // hidden gradient of the summed energy, w2 . act'(pre) -> sC plane 1 [atom][64] (B operand of the GEMM below)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wv2 = *(const f32x4*)(Hd.w2 + 32 * wv + 8 * q + 4 * hi);
          f32x4 gh;
          if (Hd.act == SPK_ACT_SILU) gh = f32x4{wv2.x * pm_silu_grad(acc[4 * q]), wv2.y * pm_silu_grad(acc[4 * q + 1]), wv2.z * pm_silu_grad(acc[4 * q + 2]), wv2.w * pm_silu_grad(acc[4 * q + 3])};
          else gh = f32x4{wv2.x * spk_sigmoid(acc[4 * q]), wv2.y * spk_sigmoid(acc[4 * q + 1]), wv2.z * spk_sigmoid(acc[4 * q + 2]), wv2.w * spk_sigmoid(acc[4 * q + 3])};      // ssp' = sigmoid
          if (el >= na) gh = f32x4{0.f, 0.f, 0.f, 0.f};
          *(f32x4*)(sC + PM_TILE + el * PM_LD + 32 * wv + 8 * q + 4 * hi) = gh;
        }
      }
      PM_BARRIER();
      // dE/dq_L = gh W1 (feature tile wv - 4 on waves 4..7, K = 64) -> global: the backward launch starts from it
      if (wv >= 4) {
        const int ft = wv - 4;
        f32x4 av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) av[u] = pm_ld<f32x4>(Hd.w1t + (size_t)(32 * ft) * 64, (unsigned)((el * 64 + 8 * u + 4 * hi) * 4));
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = pm_mma8(av, sC + PM_TILE + el * PM_LD + 4 * hi, acc);
        if (el < na) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            pm_st<f32x4>(a.gq_out + (size_t)a0 * F + 32 * ft, (unsigned)((el * F + 8 * q + 4 * hi) * 4), f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
        }
      }
      // wave 1 holds the molecule id of atom (lane) in my_mol: segment heads add their run, one atomic per (group, molecule)
      if (wv == 1) {
        float y = 0.f;
        if (lane < 32) {
          y = (Hd.b2 ? Hd.b2[0] : 0.f) + sH[lane] + sH[32 + lane];
          if (lane >= na) y = 0.f;
        }
        const long long first_mol = __shfl(my_mol, 0, 64);
        if (__all(lane >= na || my_mol == first_mol)) {
          float sum = y;          // the whole group is one molecule (the usual case): a butterfly over the 32 atom lanes
#pragma unroll
          for (int mk = 16; mk >= 1; mk >>= 1) sum += __shfl_xor(sum, mk, 64);
          if (lane == 0 && na > 0) { if (Hd.direct_store) Hd.E[my_mol] = sum; else unsafeAtomicAdd(Hd.E + my_mol, sum); }
        } else {
          const long long prev = __shfl_up(my_mol, 1, 64);
          const bool head_of_run = lane < na && (lane == 0 || prev != my_mol);
          float sum = 0.f;
          for (int b = 0; b < 32; ++b) {               // runs are contiguous: every head walks forward while the id matches
            const float yb = spk_readlane_f(y, b);
            const long long mb = __shfl(my_mol, b, 64);
            if (head_of_run && b >= lane && b < na && mb == my_mol) sum += yb;
          }
          if (head_of_run) { if (Hd.direct_store) Hd.E[my_mol] = sum; else unsafeAtomicAdd(Hd.E + my_mol, sum); }
        }
      }
    }
  }
}

// =====================================================================================================================
// Backward: dL/dr_ij (and dL/dq0 on request) from dL/dq_L, dL/dmu_L and the saved tensors of the forward, one launch.
// LDS: running gradients gq [32][LD], gmu [3][32][LD]; four work tiles X0..X3; per-edge gradient sums sG (written to gr once).
// Per interaction, last to first (B = barrier):
//   M1  element-wise: ga_mu = sum_x gmu_x W_x, ga_qmu = gq sum_x V_x W_x, U = gq a_qmu                              (painn.py:110-116 transposed)
//   M2  g_hid = ([gq | ga_mu | ga_qmu] W_b2) * silu'(pre_b)      K = 384 split over the two teams, partial sums through LDS
//   M3  [gq += | g_nv =] g_hid W_b1                              8 feature tiles over the 8 waves
//   M4  per component x: gV = U W_x + g_nv V_x / |V|, gW = U V_x + gmu_x a_mu;  gmu_x += [gV | gW] W_mix (K = 256 split over the teams)
//   message: (A) the transposed sums of the centre atom through the reverse edge (symmetric lists: Phi_ij = Phi_ji, u_ji = -u_ij)
//            -> gc, new gmu;  (B) the geometry gradient of every edge of the row -> sG.  One wavefront per centre atom, a lane owns
//            two channels, gq / gmu / mu rows of the neighbours from LDS, its context rows c_j from L2 (requested one edge ahead)
//   ctx  gq += ((gc W_a2) * silu'(pre_a)) W_a1
// The first interaction of an eval-mode backward (nobody asks for dL/dq0) forms the geometry gradient only.
struct PmLayerBwd {
  const float* ic2T_p;    // packed image of A = ictx_w2^T  (rows F,  K = 3F)
  const float* ic1T_p;    // A = ictx_w1^T  (rows 2F, K = F)
  const float* mixT_p;    // A = mix_w^T    (rows F,  K = 2F)
  const float* ctx2T_p;   // A = ctx_w2^T   (rows F,  K = 3F)
  const float* ctx1T_p;   // A = ctx_w1^T   (rows F,  K = F)
  const float *wf, *bf;
};
struct PmBwdArgs {
  PmLayerBwd L[PM_MAXL];
  int n_layers;
  const float* gq_out;      // [N, F] or null (zeros)
  const float* gmu_out;     // [N, 3, F] or null (zeros)
  const float* rij;
  const int32_t* rev;       // POT: reverse edge of every edge (symmetric list); rij, gq_out = pair vectors and dL/dq_L written by the forward launch
  float* forces;            // POT: [N, 3] = -dE/dR, written instead of gr
  int split;                // host side: 1 = a.L[] holds the SPLIT weight images, launch the SP instances
  const int64_t* idx_j;
  const int32_t* rowptr;
  const int32_t* grp_atom0;
  int n_groups;
  const float* saved;
  float* gc_scratch;        // [2][N, 3F]: gc rows | new gmu rows of the message backward
  float* gr;                // [E, 3], every entry written once
  float* gq0;               // [N, F] or null
  int64_t N;
  float eps;
  RadialDev rb;
  long long* dbg;
  int assign;
};
#define PM_BSTAMP(n) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[64 + (n)] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ float pm_silu_grad(float x) {
  const float sg = spk_sigmoid(x);
  return sg * (1.0f + x * (1.0f - sg));
}
// FOUR sums over the 64 lanes at once (the per-edge scalars of the geometry gradient): two select-and-add steps inside the quads
// leave lane r (mod 4) with value r summed over its quad, two row rotations by 4 and 8 lanes sum the rows of 16, a swizzle and a
// half-wave exchange sum the four rows -- 13 lane-crossing operations instead of 4 x (4 DPP + 4 v_readlane)
__device__ __forceinline__ float pm_dpp_add(float keep, float send, int ctrl_sel) {
  // keep + (send of the partner lane); the four DPP patterns used below
  float o;
  switch (ctrl_sel) {
    case 0: o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true)); break;    // quad_perm [1,0,3,2]
    case 1: o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xF, 0xF, true)); break;    // quad_perm [2,3,0,1]
    case 2: o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x124, 0xF, 0xF, true)); break;   // row_ror:4
    default: o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xF, 0xF, true)); break;  // row_ror:8
  }
  return keep + o;
}
__device__ __forceinline__ void pm_wave_sum4(float& a, float& b, float& c, float& d, int lane) {
  const bool o1 = lane & 1, o2 = lane & 2;
  const float ab = pm_dpp_add(o1 ? b : a, o1 ? a : b, 0);
  const float cd = pm_dpp_add(o1 ? d : c, o1 ? c : d, 0);
  float v = pm_dpp_add(o2 ? cd : ab, o2 ? ab : cd, 1);      // lane & 3 == 0: a, 1: b, 2: c, 3: d, summed over the quad
  v = pm_dpp_add(v, v, 2);
  v = pm_dpp_add(v, v, 3);                                  // ... over the row of 16
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));      // lane ^ 16
  v += __shfl_xor(v, 32, 64);
  a = spk_readlane_f(v, 0); b = spk_readlane_f(v, 1); c = spk_readlane_f(v, 2); d = spk_readlane_f(v, 3);
}
// phi_k(d), phi_k'(d) for the lane's own k, parameters preloaded
__device__ __forceinline__ void pm_phi_d(int kind, float p0k, float p1k, float d, float& phi, float& dphi) {
  if (kind == SPK_RBF_GAUSSIAN) {
    const float c = -0.5f / (p1k * p1k);
    const float t = d - p0k;
    phi = expf(c * t * t);
    dphi = 2.0f * c * t * phi;
  } else {
    float sn, co;
    sincosf(p0k * d, &sn, &co);
    if (d == 0.0f) { phi = sn; dphi = 0.f; }
    else { const float inv = 1.0f / d; phi = sn * inv; dphi = (p0k * co - phi) * inv; }
  }
}


// message backward of the atoms of one wave (see the header comment of this section); rm = the new gmu rows (written by the
// caller after the barrier), gc -> global scratch, geometry gradients -> sG
// MU0 / GEOM as template parameters: as run-time flags the compiler turned the small guarded blocks into per-component selects
// NIT x 512 threads x 16 bytes from global rows into LDS tiles: element `it` of a thread = (plane it >> 1, row (tid >> 5) + 16 (it & 1),
// float4 column tid & 31).  Branch-free (rows >= na read row 0 and store zeros) so that ALL loads are in flight before the first LDS
// store: a `for (s = tid; ...; s += 512) if (row < na) ...` loop is compiled into one exposed memory round trip PER ITERATION.
template <int NIT, bool NT, class SrcFn>
__device__ __forceinline__ void pm_fill_tiles(float* dst0 /* LDS plane 0, planes PM_TILE apart */, int tid, int na, SrcFn src /* (plane, row) -> row base */) {
  f32x4 v[NIT];
  const int r0 = tid >> 5, c4 = tid & 31;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = r0 + 16 * (it & 1), rr = row < na ? row : 0;
    const f32x4* p = (const f32x4*)(src(it >> 1, rr)) + c4;
    if (NT) v[it] = __builtin_nontemporal_load(p);
    else v[it] = *p;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int row = r0 + 16 * (it & 1);
    *(f32x4*)(dst0 + (it >> 1) * PM_TILE + row * PM_LD + 4 * c4) = row < na ? v[it] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

template <int K, bool MU0, bool GEOM>
__device__ __forceinline__ void pm_message_bwd(const PmFilt<K>& Wf, const float* __restrict__ bf, const float* __restrict__ sGq, const float* __restrict__ sGmu,
                                               const float* __restrict__ sMuIn, const float* __restrict__ c_g, float* __restrict__ gc_g,
                                               const f32x4* __restrict__ sEa, const float* __restrict__ sEd, float* __restrict__ sG,
                                               const int* __restrict__ sRow, const int* __restrict__ myAsg, float* __restrict__ myPhi, float* __restrict__ myFc,
                                               int rbf_kind, float p0k, float p1k, float cutoff, int lane, float* __restrict__ gmu_g, long long* dbgw /* tuning: 16 stamps of this wave, or null */,
                                               int* __restrict__ ctr, int na, bool dyn) {
  constexpr bool mu0 = MU0, geom = GEOM;
#define PM_MSTAMP(n) do { if (dbgw && lane == 0) dbgw[n] = (long long)__builtin_readcyclecounter(); } while (0)
  PM_MSTAMP(0);
  // gmu_g: global scratch [n, 3, F] receiving the new gmu rows (the LDS rows are still being read by other waves; kept in
  // registers across the row loops instead, the 24 values pushed the loop into scratch reloads on every edge)
  const int hi = lane >> 5;
  (void)bf;
  const pm_f2 zero2 = {0.f, 0.f};
  // global rows are addressed as (wave-uniform base, 32-bit lane offset): 64-bit per-lane addresses cost register pairs, and a
  // spilled one is reloaded from scratch between the result stores of an atom -- a wait for their acknowledgement (loads, stores
  // and scratch reloads share one in-order counter)
  const unsigned lo = (unsigned)(8 * lane);
  pm_f2 cn0 = zero2, cn1 = zero2, cn2 = zero2;          // context rows of the NEXT edge's neighbour (from L2)
  int at = pm_next_atom(myAsg, ctr, na, 0, lane, dyn);
  if (at >= 0 && sRow[at] < sRow[at + 1]) {
    const unsigned co = (unsigned)__float_as_int(sEa[sRow[at]].x) * 1536u + lo;
    cn0 = pm_ld<pm_f2>(c_g, co); cn1 = pm_ld<pm_f2>(c_g, co + 512u); cn2 = pm_ld<pm_f2>(c_g, co + 1024u);
  }
  // (a plain loop: one copy of the row code per instance instead of four -- and in the dynamic mode a wave may take more than four atoms)
  for (int it = 0; at >= 0; ++it) {
    pm_f2 accq = zero2, accR = zero2, av0 = zero2, av1 = zero2, av2 = zero2;
    pm_f2 gma0 = zero2, gma1 = zero2, gma2 = zero2;
    int at_next = -1;
    {
      PM_MSTAMP(1 + 3 * (it < 4 ? it : 3));
      const int rs = sRow[at], re = sRow[at + 1];
      // cutoff value and slope of the edges of the row: lanes = edges
      {
        float fc = 0.f, dfc = 0.f;
        if (rs + lane < re) spk_cutoff_eval(cutoff, sEd[rs + lane], fc, dfc);
        myFc[lane] = fc; myFc[64 + lane] = dfc;
      }
      const int io = at * PM_LD + 2 * lane;
      const pm_f2 gqa = *(const pm_f2*)(sGq + io);
      gma0 = *(const pm_f2*)(sGmu + io); gma1 = *(const pm_f2*)(sGmu + PM_TILE + io); gma2 = *(const pm_f2*)(sGmu + 2 * PM_TILE + io);
      for (int le = rs; le < re; ++le) {
        const pm_f2 cq = cn0, cR = cn1, cm = cn2;
        if (le + 1 < re) {
          const unsigned co = (unsigned)__float_as_int(sEa[le + 1].x) * 1536u + lo;
          cn0 = pm_ld<pm_f2>(c_g, co); cn1 = pm_ld<pm_f2>(c_g, co + 512u); cn2 = pm_ld<pm_f2>(c_g, co + 1024u);
        }
        const float fc = myFc[le - rs], dfc = myFc[64 + le - rs];
        if (fc == 0.f && dfc == 0.f) continue;          // pairs at / beyond the cutoff contribute exactly zero
        const f32x4 ea = sEa[le];
        const int jl = __float_as_int(ea.x);
        const float ux = ea.y, uy = ea.z, uz = ea.w, d = sEd[le];
        {
          float bs, dbs;
          pm_basis<K>(rbf_kind, lane & 31, p0k, p1k, d, fc, dfc, bs, dbs);
          myPhi[lane] = hi ? dbs : bs;
        }
        const int jo = jl * PM_LD + 2 * lane;
        f32x4 pa[K / 4 + 1], pd[K / 4 + 1];
#pragma unroll
        for (int c = 0; c < K / 4 + 1; ++c) { pa[c] = *(const f32x4*)(myPhi + 4 * c); pd[c] = *(const f32x4*)(myPhi + 32 + 4 * c); }
        const pm_f2 gu = gma0 * ux + gma1 * uy + gma2 * uz;
        pm_f2 ddv, mR;
        {   // q part
          const pm_f2 Fq = pm_filter2<K / 4 + 1>(Wf.w[0][0], Wf.w[0][1], pa), dFq = pm_filter2<K / 4 + 1>(Wf.w[0][0], Wf.w[0][1], pd);
          if (!geom) accq += Fq * *(const pm_f2*)(sGq + jo);
          ddv = cq * gqa * dFq;
        }
        {   // R part
          const pm_f2 FR = pm_filter2<K / 4 + 1>(Wf.w[1][0], Wf.w[1][1], pa), dFR = pm_filter2<K / 4 + 1>(Wf.w[1][0], Wf.w[1][1], pd);
          if (!geom) {
            const pm_f2 gb0 = *(const pm_f2*)(sGmu + jo), gb1 = *(const pm_f2*)(sGmu + PM_TILE + jo), gb2 = *(const pm_f2*)(sGmu + 2 * PM_TILE + jo);
            accR -= FR * (gb0 * ux + gb1 * uy + gb2 * uz);
          }
          ddv += cR * gu * dFR;
          mR = FR * cR;
        }
        if (!mu0) {   // mu part
          const pm_f2 Fm = pm_filter2<K / 4 + 1>(Wf.w[2][0], Wf.w[2][1], pa), dFm = pm_filter2<K / 4 + 1>(Wf.w[2][0], Wf.w[2][1], pd);
          if (!geom) {
            av0 += Fm * *(const pm_f2*)(sGmu + jo); av1 += Fm * *(const pm_f2*)(sGmu + PM_TILE + jo); av2 += Fm * *(const pm_f2*)(sGmu + 2 * PM_TILE + jo);
          }
          const pm_f2 gm = gma0 * *(const pm_f2*)(sMuIn + jo) + gma1 * *(const pm_f2*)(sMuIn + PM_TILE + jo) + gma2 * *(const pm_f2*)(sMuIn + 2 * PM_TILE + jo);
          ddv += cm * gm * dFm;
        }
        const pm_f2 t0 = gma0 * mR, t1 = gma1 * mR, t2 = gma2 * mR;
        float dd = ddv.x + ddv.y, tux = t0.x + t0.y, tuy = t1.x + t1.y, tuz = t2.x + t2.y;
        pm_wave_sum4(dd, tux, tuy, tuz, lane);
        if (lane == 0 && d > 0.f) {
          const float dot = tux * ux + tuy * uy + tuz * uz;
          const float invd = 1.0f / d;
          sG[3 * le] += dd * ux + (tux - dot * ux) * invd;
          sG[3 * le + 1] += dd * uy + (tuy - dot * uy) * invd;
          sG[3 * le + 2] += dd * uz + (tuz - dot * uz) * invd;
        }
      }
      PM_MSTAMP(2 + 3 * (it < 4 ? it : 3));
      // the first context row of the wave's NEXT atom is requested BEFORE the result rows of this one are stored: behind the stores
      // the wait for it is a wait for their acknowledgement by L2
      pm_f2 cma = zero2;
      if (!geom && !mu0) cma = pm_ld<pm_f2>(c_g, (unsigned)at * 1536u + 1024u + lo);
      {
        at_next = pm_next_atom(myAsg, ctr, na, it + 1, lane, dyn);
        if (at_next >= 0 && sRow[at_next] < sRow[at_next + 1]) {
          const unsigned co = (unsigned)__float_as_int(sEa[sRow[at_next]].x) * 1536u + lo;
          cn0 = pm_ld<pm_f2>(c_g, co); cn1 = pm_ld<pm_f2>(c_g, co + 512u); cn2 = pm_ld<pm_f2>(c_g, co + 1024u);
        }
      }
      if (!geom) {
        // gc_i = (acc_q, acc_R, sum_x mu_i[x] acc_v[x]);  gmu_i[x] = gmu1_i[x] + c_i^mu acc_v[x]
        pm_f2 gcm = zero2;
        if (!mu0) gcm = *(const pm_f2*)(sMuIn + io) * av0 + *(const pm_f2*)(sMuIn + PM_TILE + io) * av1 + *(const pm_f2*)(sMuIn + 2 * PM_TILE + io) * av2;
        const unsigned go = (unsigned)at * 1536u + lo;
        pm_st<pm_f2>(gc_g, go, accq); pm_st<pm_f2>(gc_g, go + 512u, accR); pm_st<pm_f2>(gc_g, go + 1024u, gcm);
        gma0 += cma * av0; gma1 += cma * av1; gma2 += cma * av2;
        pm_st<pm_f2>(gmu_g, go, gma0); pm_st<pm_f2>(gmu_g, go + 512u, gma1); pm_st<pm_f2>(gmu_g, go + 1024u, gma2);
      }
      PM_MSTAMP(3 + 3 * (it < 4 ? it : 3));
    }
    at = at_next;
  }
  PM_MSTAMP(13);
#undef PM_MSTAMP
}

// K-split GEMM epilogue shared by the two teams: wave t of either team holds a partial accumulator of output tile t.  Each team
// finishes HALF of the tile (team 0 the channel groups q = 0, 1, team 1 q = 2, 3): it hands the other half of its partial sums
// to the partner through the output tile, and after the barrier adds the partner's half to its own, applies silu'(pre) and writes
// the result in place.  (One team finishing the whole tile took 5 k cycles with the other idle.)
template <int Q0>
__device__ __forceinline__ void pm_split_send(float* __restrict__ X, const f32x16& acc, int el, int t, int hi) {
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const int q = 2 - Q0 + qq;
    *(f32x4*)(X + el * PM_LD + 32 * t + 8 * q + 4 * hi) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  }
}
template <int Q0>
__device__ __forceinline__ void pm_split_load_pre(f32x4 (&pre)[2], const float* __restrict__ pre_g /* + 32 t */, int el, int na, int hi) {
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) pre[qq] = pm_ld<f32x4>(pre_g, (unsigned)(((el < na ? el : 0) * 128 + 8 * (Q0 + qq) + 4 * hi) * 4));
}
template <int Q0>
__device__ __forceinline__ void pm_split_finish(float* __restrict__ X, const f32x16& acc, const f32x4 (&pre)[2], int el, int na, int t, int hi) {
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const int q = Q0 + qq;
    float* xp = X + el * PM_LD + 32 * t + 8 * q + 4 * hi;
    const f32x4 part = *(const f32x4*)xp;
    const f32x4 pv = el < na ? pre[qq] : f32x4{0.f, 0.f, 0.f, 0.f};
    *(f32x4*)xp = f32x4{(acc[4 * q] + part.x) * pm_silu_grad(pv.x), (acc[4 * q + 1] + part.y) * pm_silu_grad(pv.y),
                        (acc[4 * q + 2] + part.z) * pm_silu_grad(pv.z), (acc[4 * q + 3] + part.w) * pm_silu_grad(pv.w)};
  }
}

// forces = -dE/dR: r_ij = R_j - R_i, so atom i collects -gr of its own edges and +gr of their reverse edges (symmetric list: the
// reverse of a row's edge ends at the row's atom) -- one thread per (atom, component), fixed order, no atomics
__device__ __forceinline__ void pm_pot_forces(const float* sG, const int* sRow, const int* sRev, float* __restrict__ forces, int a0, int na) {
  for (int s = threadIdx.x; s < 3 * na; s += 512) {
    const int at = s / 3, c = s - 3 * at;
    float acc = 0.f;
    for (int le = sRow[at]; le < sRow[at + 1]; ++le) acc += sG[3 * le + c] - sG[3 * sRev[le] + c];
    forces[3 * (size_t)(a0 + at) + c] = acc;
  }
}

template <int K, bool POT, bool SP>
__global__ __launch_bounds__(512) void k_painn_mol_bwd(PmBwdArgs a) {
  constexpr int F = 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sGq = smem;                         // [32][LD]    running dL/dq
  float* sGmu = sGq + PM_TILE;               // [3][32][LD] running dL/dmu
  float* X0 = sGmu + 3 * PM_TILE;            // work tiles
  float* X1 = X0 + PM_TILE;
  float* X2 = X1 + PM_TILE;
  float* X3 = X2 + PM_TILE;
  float* sG = X3 + PM_TILE;                  // [PM_MAXEDGES][3] geometry gradient of every directed edge of the group
  float* sPhi = sG + 3 * PM_MAXEDGES;        // [8][64]
  float* sFc = sPhi + 8 * 64;                // [8][128] cutoff value | slope of the edges of the row a wave works on
  int* sRow = (int*)(sFc + 8 * 128);         // [33]
  int* sAsg = sRow + 36;                     // [8][4]
  int* sRev = sAsg + 32;                     // POT: [PM_MAXEDGES] local index of the reverse edge
  // during the message phase: X0..X2 = mu entering the interaction (component planes), X3 = edge records
  f32x4* sEa = (f32x4*)X3;                   // [PM_MAXEDGES] (local neighbour, unit vector)
  float* sEd = X3 + 4 * PM_MAXEDGES;         // [PM_MAXEDGES] distance

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, el = lane & 31;
  const int team = wv >> 2, t = wv & 3;
  const float p0k = (el < K && a.rb.p0) ? a.rb.p0[el] : 0.f;
  const float p1k = (el < K && a.rb.p1) ? a.rb.p1[el] : 1.f;
  const float cutoff = a.rb.cutoff;
  const int64_t nf = a.N * (int64_t)F;
  const int64_t per = 17 * nf;
  float* myPhi = sPhi + wv * 64;
  float* myFc = sFc + wv * 128;
  const int* myAsg = sAsg + wv * 4;
  const bool dyn = (a.assign == 0);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
    const int a0 = a.grp_atom0[grp], na = a.grp_atom0[grp + 1] - a0;
    const int e0 = a.rowptr[a0], ne = a.rowptr[a0 + na] - e0;
    PM_BARRIER();
    PM_BSTAMP(0);
    // ---- group set-up: incoming gradients, per-edge sums, local CSR and the wave -> atoms map of the message phase
    if (a.gq_out) pm_fill_tiles<2, false>(sGq, tid, na, [&](int, int r) { return a.gq_out + (size_t)(a0 + r) * F; });
    else for (int s = tid; s < 32 * 32; s += 512) *(f32x4*)(sGq + (s >> 5) * PM_LD + 4 * (s & 31)) = z4;
    if (a.gmu_out) pm_fill_tiles<6, false>(sGmu, tid, na, [&](int x, int r) { return a.gmu_out + ((size_t)(a0 + r) * 3 + x) * F; });
    else for (int s = tid; s < 3 * 32 * 32; s += 512) *(f32x4*)(sGmu + (s >> 10) * PM_TILE + ((s >> 5) & 31) * PM_LD + 4 * (s & 31)) = z4;
    for (int s = tid; s < 3 * PM_MAXEDGES; s += 512) sG[s] = 0.f;
    if (POT) for (int le = tid; le < ne; le += 512) sRev[le] = a.rev[e0 + le] - e0;
    if (wv == 7) {
      int r0 = 0;
      if (lane <= 32) r0 = a.rowptr[a0 + (lane < na ? lane : na)] - e0;
      const int r1 = __shfl_down(r0, 1, 64);
      if (lane <= 32) sRow[lane] = r0;
      pm_assign_atoms(lane < na ? r1 - r0 : -1, na, lane, sAsg, a.assign);
      if (lane == 0) { sRow[34] = 0; sRow[35] = 0; }
    }
    for (int l = a.n_layers - 1; l >= 0; --l) {
      // (lane-derived LDS / global offsets are re-derived inside every interaction: hoisted out of the loop by the compiler they
      //  fill the prologue with dozens of live registers -- and their spills)
      int lane_o_ = lane, tid_o_ = tid;
      asm volatile("" : "+v"(lane_o_), "+v"(tid_o_));
      const int lane = lane_o_, tid = tid_o_, hi = lane >> 5, el = lane & 31;
      (void)tid; (void)hi; (void)el;
      const PmLayerBwd& P = a.L[l];
      const float* S = a.saved + (int64_t)l * per;
      const float* preA_g = S + (size_t)a0 * F;
      const float* c_g = S + nf + (size_t)a0 * 3 * F;
      const float* muin_g = S + 4 * nf + (size_t)a0 * 3 * F;
      const float* mix_g = S + 7 * nf + (size_t)a0 * 6 * F;
      const float* preB_g = S + 13 * nf + (size_t)a0 * F;
      const float* a_g = S + 14 * nf + (size_t)a0 * 3 * F;
      const bool mu0 = (l == 0);
      const bool geom = (l == 0 && a.gq0 == nullptr);
      PM_BARRIER();
      PM_BSTAMP(1 + 12 * (a.n_layers - 1 - l));

      // ================= M1: ga_mu -> X0, ga_qmu -> X1, U = gq a_qmu -> X3                 (weights of M2 requested meanwhile)
      PmW W3;
      pm_wload(W3, P.ic2T_p, 48, t, 24 * team, lane);
      if (a.dbg && blockIdx.x == 0 && lane == 0 && l == a.n_layers - 1) { asm volatile("s_nop 0" :: "v"(W3.a0[7].x)); a.dbg[64 + 52 + wv] = (long long)__builtin_readcyclecounter(); }
      {
        // (both row halves of a thread: 14 loads in flight, then the arithmetic -- branch-free, rows >= na read row 0 and store zeros)
        f32x4 mV[2][3], mW[2][3], aqm[2];
        const int r0 = tid >> 5, c4 = tid & 31;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const int row = r0 + 16 * rep, rr = row < na ? row : 0;
          const float* mp = mix_g + (size_t)rr * 6 * F + 4 * c4;
#pragma unroll
          for (int x = 0; x < 3; ++x) { mV[rep][x] = *(const f32x4*)(mp + x * 2 * F); mW[rep][x] = *(const f32x4*)(mp + x * 2 * F + F); }
          aqm[rep] = *(const f32x4*)(a_g + (size_t)rr * 3 * F + 2 * F + 4 * c4);
        }
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const int row = r0 + 16 * rep;
          const f32x4 gq = *(const f32x4*)(sGq + row * PM_LD + 4 * c4);
          f32x4 Ssum = z4, gam = z4;
#pragma unroll
          for (int x = 0; x < 3; ++x) {
            const f32x4 gm = *(const f32x4*)(sGmu + x * PM_TILE + row * PM_LD + 4 * c4);
            Ssum += mV[rep][x] * mW[rep][x];
            gam += gm * mW[rep][x];
          }
          const bool live = row < na;
          *(f32x4*)(X0 + row * PM_LD + 4 * c4) = live ? gam : z4;
          *(f32x4*)(X1 + row * PM_LD + 4 * c4) = live ? gq * Ssum : z4;
          *(f32x4*)(X3 + row * PM_LD + 4 * c4) = live ? gq * aqm[rep] : z4;
        }
      }
      PM_BARRIER();
      PM_BSTAMP(2 + 12 * (a.n_layers - 1 - l));

      // ================= M2: g_hid = ([gq | ga_mu | ga_qmu] W_b2) silu'(pre_b) -> X2; team 0: k-blocks 0..23, team 1: 24..47
      PmW Wn;       // weights of M3 (tile 4 team + t of A = W_b1^T)
      {
        const float* b0 = team ? X0 + 64 : sGq;
        const float* b1 = team ? X1 : sGq + 64;
        const float* b2 = team ? X1 + 64 : X0;
        const size_t bo = (size_t)((lane & 31) * PM_LD + 4 * (lane >> 5));
        f32x4 pb[2];      // pre_b of this team's half of the epilogue, requested before the MFMAs (rows >= na: row 0, never used)
        if (team == 0) pm_split_load_pre<0>(pb, preB_g + 32 * t, el, na, hi);
        else pm_split_load_pre<2>(pb, preB_g + 32 * t, el, na, hi);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = pm_wmma_3chunks<SP>(W3, b0 + bo, b1 + bo, b2 + bo, lane, acc);
        if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0 && l == a.n_layers - 1) { asm volatile("s_nop 0" :: "v"(acc[0])); a.dbg[64 + 40] = (long long)__builtin_readcyclecounter(); }
        if (a.dbg && blockIdx.x == 0 && lane == 0 && l == a.n_layers - 1) { asm volatile("s_nop 0" :: "v"(acc[0])); a.dbg[64 + 44 + wv] = (long long)__builtin_readcyclecounter(); }
        if (team == 0) pm_split_send<0>(X2, acc, el, t, hi);
        else pm_split_send<2>(X2, acc, el, t, hi);
        PM_BARRIER();
        if (l == a.n_layers - 1) PM_BSTAMP(41);
        if (team == 0) pm_split_finish<0>(X2, acc, pb, el, na, t, hi);
        else pm_split_finish<2>(X2, acc, pb, el, na, t, hi);
      }
      if (l == a.n_layers - 1) PM_BSTAMP(42);
      PM_BARRIER();
      PM_BSTAMP(3 + 12 * (a.n_layers - 1 - l));

      // ================= M3: g_ctx = g_hid W_b1: tiles 0..3 (q part, team 0): gq += ; tiles 4..7 (|V| part, team 1): g_nv -> X0
      PmWF Wm;      // weights of M4: A = W_mix^T, k-blocks 16 team .. 16 team + 15 (the gV resp. gW half of the contraction)
      {
        // (in this kernel a weight chunk held across a barrier is spilled right behind its load by the register allocator -- a
        //  chain of L2 round trips; requested at the start of the phase that uses it the chunk costs one exposed round trip)
        pm_wload(Wn, P.ic1T_p, 16, 4 * team + t, 0, lane);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = pm_wmma<SP>(Wn, X2, lane, acc);
        float* dst = team ? X0 : sGq;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float* xp = dst + el * PM_LD + 32 * t + 8 * q + 4 * hi;
          f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          if (team == 0) v += *(const f32x4*)xp;
          *(f32x4*)xp = v;
        }
      }
      PM_BARRIER();
      PM_BSTAMP(4 + 12 * (a.n_layers - 1 - l));

      // ================= M4: per component x: gV -> X1, gW -> X2; gmu_x += [gV | gW] W_mix  (K = 256 split over the teams)
      pm_wfload(Wm, P.mixT_p, 32, t, 16 * team, lane);
      // (the saved rows are read ONCE per thread where possible: g_nv / |V| and a_mu before the loop, V_x / W_x of the next component
      //  requested while the MFMAs of this one run -- a global round trip per component sat in front of every barrier before)
      f32x4 cV[2], cW[2], Gn[2], am[2];
      {
        f32x4 V1[2], V2[2];
        const int r0 = tid >> 5, c4 = tid & 31;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const int row = r0 + 16 * rep, rr = row < na ? row : 0;
          const float* mp = mix_g + (size_t)rr * 6 * F + 4 * c4;
          cV[rep] = *(const f32x4*)mp; V1[rep] = *(const f32x4*)(mp + 2 * F); V2[rep] = *(const f32x4*)(mp + 4 * F);
          cW[rep] = *(const f32x4*)(mp + F);
          am[rep] = *(const f32x4*)(a_g + (size_t)rr * 3 * F + F + 4 * c4);
        }
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
          const int row = r0 + 16 * rep;
          const f32x4 gnv = *(const f32x4*)(X0 + row * PM_LD + 4 * c4);
#pragma unroll
          for (int v = 0; v < 4; ++v) Gn[rep][v] = gnv[v] / sqrtf(cV[rep][v] * cV[rep][v] + V1[rep][v] * V1[rep][v] + V2[rep][v] * V2[rep][v] + a.eps);
        }
      }
      for (int x = 0; x < 3; ++x) {
        f32x4 nV[2], nW[2];
        {
          const int r0 = tid >> 5, c4 = tid & 31;
#pragma unroll
          for (int rep = 0; rep < 2; ++rep) {
            const int row = r0 + 16 * rep;
            const f32x4 U = *(const f32x4*)(X3 + row * PM_LD + 4 * c4);
            const f32x4 gm = *(const f32x4*)(sGmu + x * PM_TILE + row * PM_LD + 4 * c4);
            const bool live = row < na;
            *(f32x4*)(X1 + row * PM_LD + 4 * c4) = live ? U * cW[rep] + Gn[rep] * cV[rep] : z4;
            *(f32x4*)(X2 + row * PM_LD + 4 * c4) = live ? U * cV[rep] + gm * am[rep] : z4;
          }
          const int xn = x < 2 ? x + 1 : 2;
#pragma unroll
          for (int rep = 0; rep < 2; ++rep) {
            const int row = r0 + 16 * rep, rr = row < na ? row : 0;
            const float* mp = mix_g + (size_t)rr * 6 * F + 4 * c4 + xn * 2 * F;
            nV[rep] = *(const f32x4*)mp; nW[rep] = *(const f32x4*)(mp + F);
          }
        }
        PM_BARRIER();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = pm_wfmma<SP>(Wm, team ? X2 : X1, lane, acc);
        // (the two K halves meet through X0 -- g_nv is in registers by now: each team hands the partner half a tile and adds the other
        //  half into gmu_x; two barriers per component instead of three, nobody idle in the epilogue)
        if (team == 0) pm_split_send<0>(X0, acc, el, t, hi);
        else pm_split_send<2>(X0, acc, el, t, hi);
        PM_BARRIER();
        {
          float* gp = sGmu + x * PM_TILE + el * PM_LD + 32 * t + 4 * hi;
          const float* pp = X0 + el * PM_LD + 32 * t + 4 * hi;
          if (team == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) *(f32x4*)(gp + 8 * q) += *(const f32x4*)(pp + 8 * q) + f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          } else {
#pragma unroll
            for (int q = 2; q < 4; ++q) *(f32x4*)(gp + 8 * q) += *(const f32x4*)(pp + 8 * q) + f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          }
        }
        cV[0] = nV[0]; cV[1] = nV[1]; cW[0] = nW[0]; cW[1] = nW[1];
      }
      PM_BARRIER();
      PM_BSTAMP(5 + 12 * (a.n_layers - 1 - l));

      // ================= message backward: mu entering the interaction -> X0..X2, edge records -> X3
      PmFilt<K> Wf;
      if (!mu0) pm_fill_tiles<6, false>(X0, tid, na, [&](int x, int r) { return muin_g + ((size_t)r * 3 + x) * F; });
      for (int le = tid; le < ne; le += 512) {
        const int64_t e = (int64_t)e0 + le;
        const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
        const float d = sqrtf(rx * rx + ry * ry + rz * rz);
        const float inv = 1.0f / d;
        sEa[le] = f32x4{__int_as_float((int)(a.idx_j[e] - a0)), rx * inv, ry * inv, rz * inv};
        sEd[le] = d;
      }
      pm_filt_load<K>(Wf, P.wf, P.bf, lane, mu0);      // (requested before the tile fill the 120 registers pushed the message loop into scratch reloads)
      PM_BARRIER();
      PM_BSTAMP(6 + 12 * (a.n_layers - 1 - l));
      {
        float* gcs_w = a.gc_scratch + (size_t)a0 * 3 * F;
        long long* dbgw = (a.dbg && blockIdx.x == 0 && l == a.n_layers - 1) ? a.dbg + 128 + 16 * wv : nullptr;
        int* ctr = sRow + 34 + (l & 1);
        if (tid == 0) sRow[34 + ((l + 1) & 1)] = 0;
        const int* asg = dyn ? sAsg : myAsg;
        if (geom) pm_message_bwd<K, true, true>(Wf, P.bf, sGq, sGmu, X0, c_g, gcs_w, sEa, sEd, sG, sRow, asg, myPhi, myFc, a.rb.kind, p0k, p1k, cutoff, lane, gcs_w + 3 * nf, dbgw, ctr, na, dyn);
        else if (mu0) pm_message_bwd<K, true, false>(Wf, P.bf, sGq, sGmu, X0, c_g, gcs_w, sEa, sEd, sG, sRow, asg, myPhi, myFc, a.rb.kind, p0k, p1k, cutoff, lane, gcs_w + 3 * nf, dbgw, ctr, na, dyn);
        else pm_message_bwd<K, false, false>(Wf, P.bf, sGq, sGmu, X0, c_g, gcs_w, sEa, sEd, sG, sRow, asg, myPhi, myFc, a.rb.kind, p0k, p1k, cutoff, lane, gcs_w + 3 * nf, dbgw, ctr, na, dyn);
        PM_BSTAMP(7 + 12 * (a.n_layers - 1 - l));
        if (geom) break;          // (uniform over the workgroup: the first interaction of an eval-mode backward ends here)
        pm_wload(W3, P.ctx2T_p, 48, t, 24 * team, lane);      // (weights of the context GEMM: they arrive while the wave waits for the others)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the gc / gmu rows of this wave have reached L2
        PM_BARRIER();
      }
      // ================= context net backward: gc -> X0..X2 (part planes), gq += ((gc W_a2) silu'(pre_a)) W_a1
      {
        const float* gcs = a.gc_scratch + (size_t)a0 * 3 * F;
        const float* gms = a.gc_scratch + 3 * nf + (size_t)a0 * 3 * F;
        // (twelve 16-byte loads per thread in flight: one L2 round trip)
        f32x4 vg[6], vm[6];
        const int r0 = tid >> 5, c4 = tid & 31;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
          const int row = r0 + 16 * (it & 1), rr = row < na ? row : 0;
          vg[it] = __builtin_nontemporal_load((const f32x4*)(gcs + (size_t)rr * 3 * F + (it >> 1) * F + 4 * c4));
          vm[it] = __builtin_nontemporal_load((const f32x4*)(gms + (size_t)rr * 3 * F + (it >> 1) * F + 4 * c4));
        }
#pragma unroll
        for (int it = 0; it < 6; ++it) {
          const int row = r0 + 16 * (it & 1);
          *(f32x4*)(X0 + (it >> 1) * PM_TILE + row * PM_LD + 4 * c4) = row < na ? vg[it] : z4;
          *(f32x4*)(sGmu + (it >> 1) * PM_TILE + row * PM_LD + 4 * c4) = row < na ? vm[it] : z4;
        }
      }
      PM_BARRIER();
      PM_BSTAMP(8 + 12 * (a.n_layers - 1 - l));
      {
        const float* b0 = team ? X1 + 64 : X0;
        const float* b1 = team ? X2 : X0 + 64;
        const float* b2 = team ? X2 + 64 : X1;
        const size_t bo = (size_t)((lane & 31) * PM_LD + 4 * (lane >> 5));
        f32x4 pa[2];
        if (team == 0) pm_split_load_pre<0>(pa, preA_g + 32 * t, el, na, hi);
        else pm_split_load_pre<2>(pa, preA_g + 32 * t, el, na, hi);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = pm_wmma_3chunks<SP>(W3, b0 + bo, b1 + bo, b2 + bo, lane, acc);
        if (team == 0) pm_split_send<0>(X3, acc, el, t, hi);
        else pm_split_send<2>(X3, acc, el, t, hi);
        PM_BARRIER();
        if (team == 0) pm_split_finish<0>(X3, acc, pa, el, na, t, hi);
        else pm_split_finish<2>(X3, acc, pa, el, na, t, hi);
      }
      PM_BARRIER();
      PM_BSTAMP(9 + 12 * (a.n_layers - 1 - l));
      if (team == 0) {
        pm_wload(Wn, P.ctx1T_p, 16, t, 0, lane);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = pm_wmma<SP>(Wn, X3, lane, acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float* xp = sGq + el * PM_LD + 32 * t + 8 * q + 4 * hi;
          f32x4 v = *(const f32x4*)xp;
          v += f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          *(f32x4*)xp = v;
          if (l == 0 && a.gq0 && el < na) pm_st<f32x4>(a.gq0 + (size_t)a0 * F + 32 * t, (unsigned)((el * F + 8 * q + 4 * hi) * 4), v);
        }
      }
      PM_BSTAMP(10 + 12 * (a.n_layers - 1 - l));
    }
    // ---- the geometry gradient of every edge of the group, written exactly once
    PM_BARRIER();
    if (POT) {
      pm_pot_forces(sG, sRow, sRev, a.forces, a0, na);
    } else {
      for (int s = tid; s < 3 * ne; s += 512) a.gr[3 * (int64_t)e0 + s] = sG[s];
    }
  }
}

// ------------------------------------------------------------------------------------------ host side
static long long* g_pm_dbg = nullptr;
static int pm_assign_mode() {      // tuning: default 0 = dynamic; SPK_PM_ASSIGN=snake (1), or the cost per edge (>= 20) of the younger wave of a SIMD against 22 of the older (static greedy)
  const char* e = getenv("SPK_PM_ASSIGN");
  if (!e || e[0] == 'd') return 0;
  return e[0] == 's' ? 1 : atoi(e);
}
// tuning aid (scripts/painn_mol_timing.py): device buffer of >= 256 int64 receiving cycle stamps of thread 0 of workgroup 0 at the
// phase boundaries of the forward (entry 0: group start; 1 + 8 l ... 8 + 8 l: P1 .. end of interaction l).  NULL: off (production)
extern "C" void spk_painn_mol_set_debug_buffer(void* p) { g_pm_dbg = (long long*)p; }
static size_t painn_mol_fwd_lds() { return (size_t)(8 * PM_TILE + 8 * 64) * sizeof(float) + (PM_MAXEDGES + 16) * sizeof(PmEdge) + (36 + 32) * sizeof(int); }

// Shapes / lists the molecule-resident forward covers (everything else runs the general driver of spk_painn.hip)
bool spk_painn_mol_eligible(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb) {
  const int variant = spk_get_variant();
  if (variant != SPK_VARIANT_AUTO && variant != SPK_VARIANT_MFMA_PAIR && variant != SPK_VARIANT_MFMA) return false;
  if (getenv("SPK_NO_MOL") || getenv("SPK_NO_PAINN_MOL")) return false;
  if (m->n_atom_basis != 128 || m->n_interactions < 1 || m->n_interactions > PM_MAXL || !m->wpack) return false;
  if (rb->n_rbf != 20 && rb->n_rbf != 16 && rb->n_rbf != 12 && rb->n_rbf != 8) return false;      // instances (filter rows are fetched as 16-byte vectors)
  if (!(g->sorted && g->rowptr && g->idx_j)) return false;
  if (g->n_groups <= 0 || !g->grp_atom0 || g->max_group_atoms > 32) return false;
  if (g->max_group_pairs <= 0 || 2 * g->max_group_pairs > PM_MAXEDGES) return false;
  return true;
}

template <int K, bool TILED, bool POT, bool SP = false>
static int launch_painn_mol_fwd_t(const PmFwdArgs& a, hipStream_t stream) {
  const size_t lds = painn_mol_fwd_lds();
  auto kern = k_painn_mol_fwd<K, TILED, POT, SP>;
  static SpkPerDevice attr_set;
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_dev);
  }
  int grid = a.n_groups;
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  SpkProfScope prof("painn_mol_fwd", stream);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
template <int K>
static int launch_painn_mol_fwd(const PmFwdArgs& a, hipStream_t stream) {
  if (a.split) {      // (the caller put the split weight images into a.L[]; the tile-form experiment keeps the fp32 images)
    if (a.R) return launch_painn_mol_fwd_t<K, false, true, true>(a, stream);
    return launch_painn_mol_fwd_t<K, false, false, true>(a, stream);
  }
  if (a.R) return launch_painn_mol_fwd_t<K, false, true>(a, stream);          // the standard potential
  return a.tiled ? launch_painn_mol_fwd_t<K, true, false>(a, stream) : launch_painn_mol_fwd_t<K, false, false>(a, stream);
}

// r_ij, or (potential mode) R + offsets + head: pair vectors from the positions, q0 == null: rows of `emb` by Z, energies through the head
int spk_painn_mol_forward_ex(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab, const float* q0,
                             const float* r_ij, const float* R, const float* offsets, const float* emb, const int64_t* Z, int n_types,
                             const PmHeadDev* head, float* rij_out, float* gq_head_out, float* q_out, float* mu_out, float* saved, hipStream_t stream) {
  PmFwdArgs a;
  a.R = R; a.offsets = offsets; a.idx_i = g->idx_i; a.emb = emb; a.Z = Z; a.n_types = n_types; a.rij_out = rij_out; a.gq_out = gq_head_out;
  SPK_CHECK_ARG((R != nullptr) == (head != nullptr), "spk_painn_mol_forward: positions and head go together");
  SPK_CHECK_ARG(!R || (rij_out && gq_head_out && head->w1t), "spk_painn_mol_forward: no buffers for the pair vectors / the head gradient");
  SPK_CHECK_ARG(!emb || n_types > 0, "spk_painn_mol_forward: embedding table without a row count");
  if (head) a.head = *head; else { a.head = PmHeadDev(); }
  SPK_CHECK_ARG(!R || (head->H == 64 && head->w1 && head->b1 && head->w2 && head->idx_m && head->E && head->pre_h && (q0 || (emb && Z))), "spk_painn_mol_forward: incomplete potential arguments");
  a.n_layers = m->n_interactions;
  // the matrix-core form of the message is an EXPERIMENT, off by default (SPK_PM_TILED=1): correct (the parity tests run it), but
  // 214 us against 197 us of the row form at cfg 3 -- see the comment at pm_message_tiled and HISTORY.md 4.3a
  { const char* e = getenv("SPK_PM_TILED"); a.tiled = (e && e[0] == '1' && rb->kind == SPK_RBF_GAUSSIAN) ? 1 : 0; }
  const bool split = spk_get_split() != 0 && !(a.tiled && !R);
  a.split = split ? 1 : 0;
  for (int l = 0; l < m->n_interactions; ++l) {
    const spk_painn_layer_t& P = m->layers[l];
    PmLayerDev& D = a.L[l];
    auto img = [&](const float* w) { return split ? spk_packed_split_of(ptab, w, 0) : spk_packed_of(ptab, w, 0); };
    D.ctx1_p = img(P.ctx_w1); D.ctx1_b = P.ctx_b1;
    D.ctx2_p = img(P.ctx_w2); D.ctx2_b = P.ctx_b2;
    D.mix_p = img(P.mix_w);
    D.ic1_p = img(P.ictx_w1); D.ic1_b = P.ictx_b1;
    D.ic2_p = img(P.ictx_w2); D.ic2_b = P.ictx_b2;
    D.wf = P.filt_w; D.bf = P.filt_b;
    SPK_CHECK_ARG(D.ctx1_p && D.ctx2_p && D.mix_p && D.ic1_p && D.ic2_p, "spk_painn_mol_forward: packed weight image missing");
    SPK_CHECK_ARG(D.ctx1_b && D.ctx2_b && D.ic1_b && D.ic2_b && D.wf && D.bf, "spk_painn_mol_forward: null bias / filter weights");
  }
  a.q0 = q0; a.q_out = q_out; a.mu_out = mu_out; a.rij = r_ij;
  a.idx_j = g->idx_j; a.rowptr = g->rowptr; a.grp_atom0 = g->grp_atom0; a.n_groups = g->n_groups;
  a.saved = saved; a.N = g->n_atoms; a.eps = m->epsilon; a.rb = spk_radial_dev(rb); a.dbg = g_pm_dbg; a.assign = pm_assign_mode();
  switch (rb->n_rbf) {
    case 20: return launch_painn_mol_fwd<20>(a, stream);
    case 16: return launch_painn_mol_fwd<16>(a, stream);
    case 12: return launch_painn_mol_fwd<12>(a, stream);
    case 8: return launch_painn_mol_fwd<8>(a, stream);
    default: break;
  }
  SPK_CHECK_ARG(false, "spk_painn_mol_forward: n_rbf = %d has no instance (see spk_painn_mol_eligible)", rb->n_rbf);
  return SPK_OK;
}
int spk_painn_mol_forward(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab, const float* q0,
                          const float* r_ij, float* q_out, float* mu_out, float* saved, hipStream_t stream) {
  return spk_painn_mol_forward_ex(m, g, rb, ptab, q0, r_ij, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, q_out, mu_out, saved, stream);
}

static size_t painn_mol_bwd_lds() {
  return (size_t)(8 * PM_TILE + 3 * PM_MAXEDGES + 8 * 64 + 8 * 128) * sizeof(float) + (36 + 32 + PM_MAXEDGES) * sizeof(int);
}
// the backward additionally needs a symmetric list (the transposed sums run through the reverse edge)
bool spk_painn_mol_bwd_eligible(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb) {
  if (!spk_painn_mol_eligible(m, g, rb)) return false;
  if (getenv("SPK_NO_PAINN_MOL_BWD")) return false;
  return g->symmetric != 0;
}

template <int K, bool POT, bool SP = false>
static int launch_painn_mol_bwd_t(const PmBwdArgs& a, hipStream_t stream) {
  const size_t lds = painn_mol_bwd_lds();
  auto kern = k_painn_mol_bwd<K, POT, SP>;
  static SpkPerDevice attr_set;
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_dev);
  }
  int grid = a.n_groups;
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  SpkProfScope prof("painn_mol_bwd", stream);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

template <int K>
static int launch_painn_mol_bwd(const PmBwdArgs& a, hipStream_t stream) {
  if (a.split) return a.forces ? launch_painn_mol_bwd_t<K, true, true>(a, stream) : launch_painn_mol_bwd_t<K, false, true>(a, stream);
  return a.forces ? launch_painn_mol_bwd_t<K, true>(a, stream) : launch_painn_mol_bwd_t<K, false>(a, stream);
}

// gq_out / gmu_out may be null (zeros); gr [E, 3] is overwritten (no clearing needed); gq0 may be null; gc_scratch: 2 x [N, 3F] floats.
// Potential mode (forces given; r_ij, gq_out = the pair vectors and dL/dq_L = head gradient written by the forward launch; gr / gq0
// null): forces [N, 3] = -dE/dR are written instead of gr.
int spk_painn_mol_backward_ex(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab, const float* gq_out,
                              const float* gmu_out, const float* r_ij, const float* saved,
                              float* gc_scratch, float* gr, float* gq0, float* forces, hipStream_t stream) {
  PmBwdArgs a;
  a.rev = g->rev; a.forces = forces;
  const bool split = spk_get_split() != 0;
  a.split = split ? 1 : 0;
  SPK_CHECK_ARG(!forces || (g->rev && r_ij && gq_out && !gr && !gq0), "spk_painn_mol_backward: incomplete potential arguments");
  a.n_layers = m->n_interactions;
  for (int l = 0; l < m->n_interactions; ++l) {
    const spk_painn_layer_t& P = m->layers[l];
    PmLayerBwd& D = a.L[l];
    auto img = [&](const float* w) { return split ? spk_packed_split_of(ptab, w, 1) : spk_packed_of(ptab, w, 1); };
    D.ic2T_p = img(P.ictx_w2);
    D.ic1T_p = img(P.ictx_w1);
    D.mixT_p = img(P.mix_w);
    D.ctx2T_p = img(P.ctx_w2);
    D.ctx1T_p = img(P.ctx_w1);
    D.wf = P.filt_w; D.bf = P.filt_b;
    SPK_CHECK_ARG(D.ic2T_p && D.ic1T_p && D.mixT_p && D.ctx2T_p && D.ctx1T_p && D.wf && D.bf, "spk_painn_mol_backward: packed weight image missing");
  }
  a.gq_out = gq_out; a.gmu_out = gmu_out; a.rij = r_ij; a.idx_j = g->idx_j; a.rowptr = g->rowptr; a.grp_atom0 = g->grp_atom0;
  a.n_groups = g->n_groups; a.saved = saved; a.gc_scratch = gc_scratch; a.gr = gr; a.gq0 = gq0; a.N = g->n_atoms; a.eps = m->epsilon;
  a.rb = spk_radial_dev(rb); a.dbg = g_pm_dbg; a.assign = pm_assign_mode();
  switch (rb->n_rbf) {
    case 20: return launch_painn_mol_bwd<20>(a, stream);
    case 16: return launch_painn_mol_bwd<16>(a, stream);
    case 12: return launch_painn_mol_bwd<12>(a, stream);
    case 8: return launch_painn_mol_bwd<8>(a, stream);
    default: break;
  }
  SPK_CHECK_ARG(false, "spk_painn_mol_backward: n_rbf = %d has no instance", rb->n_rbf);
  return SPK_OK;
}
int spk_painn_mol_backward(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab, const float* gq_out,
                           const float* gmu_out, const float* r_ij, const float* saved, float* gc_scratch, float* gr, float* gq0, hipStream_t stream) {
  return spk_painn_mol_backward_ex(m, g, rb, ptab, gq_out, gmu_out, r_ij, saved, gc_scratch, gr, gq0, nullptr, stream);
}
