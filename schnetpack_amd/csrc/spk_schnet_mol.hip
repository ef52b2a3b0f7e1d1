// Molecule-resident SchNet representation (representation/schnet.py:147-173) for BATCHES OF SMALL MOLECULES.
//
// A batch produced by the reference's collate function (data/loader.py:35-46) is block diagonal: no edge leaves a
// molecule.  With <= 32 atoms per block the whole representation of a block -- every interaction: in2f, the
// continuous-filter convolution, f2out, the residual -- is local to ONE workgroup: the atom features live in LDS for the
// whole kernel and nothing but the saved-for-backward tensors goes to memory.  One launch replaces the 1 + 2 L launches
// of the general driver (spk_schnet.hip), the float atomics of the pair kernels and their memsets; configs[1] of
// BASELINE.json (256 aspirin frames) maps one molecule onto each of the 256 compute units.
//
// Workgroup = 8 wavefronts, one group of atoms (a block, or several small blocks, <= 32 atoms).  Per interaction:
//   A. task queue over  (pair tile, channel tile) filter tasks  +  4 in2f tasks:
//        filter task: 32 undirected pairs -> phi -> GEMM 1 (hidden = ssp(W1 phi + b1), rows = hidden channels) ->
//                     GEMM 2 with swapped operands (rows = pairs, columns = the 32 channels of the task) -> raw filter
//                     outputs g to memory (the tensor the backward reads anyway; L2-resident, 78 KB per aspirin frame)
//        in2f task:   h[:, 32t:32t+32] = x W_in^T   (T-GEMM, weights streamed from their packed image)
//   B. y[a] = sum_{b in row(a)} h[b] * g[pair(a,b)] * f_c   -- a per-atom row sum over the directed CSR of the block,
//        thread = (channel, atom quarter): no atomics, no scatter, deterministic
//   C. t = ssp(y W3^T + b3);  x += t W4^T + b4      (two T-GEMM phases; the idle half of the workgroup stages the
//        filter weights of the next interaction into LDS meanwhile)
// fp32 MFMA (v_mfma_f32_32x32x2_f32) throughout: 1e-5 parity with the reference rules out bf16.
#include "spk_common.h"
#include "spk_pack.h"
#include "spk_split.h"
#include "spk_filter_split.h"

#define ML_MAXL 6
#define ML_LD 132          // row stride (floats) of the [32][128] activation tiles in LDS: conflict-free 16-byte accesses
#define ML_MAXPAIRS 384    // pairs per group (28 fully connected atoms; LDS budget of the backward)
#define ML_MAXEDGES 768
#define ML_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)

bool spk_schnet_mol_bwd_eligible(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb);

struct MolLayerDev {
  const float* in2f_p;                  // packed forward image of in2f.weight [NF, F] (spk_pack_weight_f32)
  const float *w1, *b1, *w2, *b2;       // filter network, raw state_dict tensors
  const float* w2_img;                  // split-precision LDS image of w2 (wpack), or null: made from w2 while staging
  const float *o1_p, *o1_b, *o2_p, *o2_b;  // packed forward images of f2out.0 / f2out.1 and their biases
};

// The callers either side of the representation, folded into the two launches when the model is the standard potential
// (PairwiseDistances -> SchNet -> Atomwise(sum) -> Forces; atomistic/distances.py:14-26, atomwise.py:69-88, response.py:59-76):
// r_ij is formed from the positions, the default 2-layer energy head runs on the atom tile that is still in LDS, and the
// backward starts from dE/dE_mol and ends at dE/dR.
struct MolHeadDev {
  const float* w1;        // outnet.0.weight [H, F]   (null: no head)
  const float* w1t;       // its transpose   [F, H]   (backward)
  const float* b1;        // [H]
  const float* w2;        // outnet.1.weight [H]
  const float* b2;        // [1]
  int H, act;
  const int64_t* idx_m;   // [N]
  float* E;               // forward: [n_mol], accumulated with one atomic per (group, molecule): cleared by the caller ...
  float* pre_h;           // [N, H] pre-activation of the head's hidden layer (saved for the backward)
  const float* gE;        // backward: dL/dE [n_mol]; null = ones (forces of the summed energy)
  int direct_store;       // ... unless every molecule lies inside one group: plain stores, nothing to clear
  int negate;             // backward: write -dL/dR (= the forces when gE is ones)
};

struct MolFwdArgs {
  MolLayerDev L[ML_MAXL];
  int n_layers;
  const float* x0;          // [N, 128], or null: rows of the embedding table
  const float* emb;         // [n_types, 128] nuclear embedding table (schnet.py:126-128) with
  const int64_t* Z;         // [N] atomic numbers
  int n_types;
  float* x_out;             // [N, 128]
  const float* rij;         // [E, 3], or null: r_ij = R[j] - R[i] + offsets
  const float* R;           // [N, 3]
  const float* offsets;     // [E, 3] or null
  MolHeadDev head;
  const int64_t* idx_i;
  const int64_t* idx_j;
  const int32_t* half;      // canonical edge of every undirected pair (ascending)
  const int32_t* rowptr;    // CSR of idx_i
  const int32_t* edge_pair; // position in `half` of the pair every directed edge belongs to
  const int32_t* grp_atom0; // [G+1]
  const int32_t* grp_pair0; // [G+1]
  int n_groups;
  float* saved;             // per interaction h [N,128] | pre3 [N,128]
  float* gbase;             // per interaction [gsz] raw filter outputs, row = position of the pair in `half`
  int64_t gsz, N;
  RadialDev rb;
  int compact;              // drop the pairs beyond the cutoff from the tiles (lists with a skin); the backward does the same
  long long* dbg;           // tuning aid: cycle stamps, see spk_schnet_mol_set_debug_buffer (null in production)
};
#define ML_STAMP(n) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[n] = (long long)__builtin_readcyclecounter(); } while (0)

// per pair of a group: local atoms i | j << 8, distance, f_c(d), f_c'(d) -- computed once per group, used by every interaction
struct __attribute__((aligned(16))) MolPair { int ij; float d; float fc; float dfc; };

// pair records of a group: geometry is the same for every interaction.  ij = local i | local j << 8 | position of the pair in the
// group's pair list << 16.  With `compact` (lists with a skin: MD) only the pairs INSIDE the cutoff get a record -- the others
// contribute exactly zero (f_c = f_c' = 0) and are not worth a tile of filter GEMMs; the order of the list is kept (ballot +
// prefix over the waves: deterministic, the backward reproduces it), sMap[position] = record index or -1.  Returns the number of
// records (uniform over the workgroup).  Contains workgroup barriers: call it from uniform code, np <= 512.
__device__ __forceinline__ void ml_edge_vector(const float* __restrict__ rij, const float* __restrict__ R, const float* __restrict__ offsets,
                                               const int64_t* __restrict__ idx_i, const int64_t* __restrict__ idx_j, int64_t e, float& rx, float& ry,
                                               float& rz) {
  if (rij) { rx = rij[3 * e]; ry = rij[3 * e + 1]; rz = rij[3 * e + 2]; return; }
  const int64_t i = idx_i[e], j = idx_j[e];
  rx = R[3 * j] - R[3 * i]; ry = R[3 * j + 1] - R[3 * i + 1]; rz = R[3 * j + 2] - R[3 * i + 2];
  if (offsets) { rx += offsets[3 * e]; ry += offsets[3 * e + 1]; rz += offsets[3 * e + 2]; }
}

__device__ __forceinline__ int ml_pair_records(MolPair* sP, short* sMap, int* sScan, const int32_t* __restrict__ half, const float* __restrict__ rij,
                                               const float* __restrict__ R, const float* __restrict__ offsets,
                                               const int64_t* __restrict__ idx_i, const int64_t* __restrict__ idx_j, int p0, int np, int a0,
                                               float cutoff, bool compact, int tid, float* r_keep = nullptr) {
  // r_keep (3 floats of the caller, or null): the vector of the pair at position `tid` of the list (np <= 512: one pair per thread)
  const int lane = tid & 63, wv = tid >> 6;
  MolPair pr;
  bool keep = false;
  if (r_keep) { r_keep[0] = 0.f; r_keep[1] = 0.f; r_keep[2] = 0.f; }
  if (tid < np) {
    const int64_t e = half[p0 + tid];
    float rx, ry, rz;
    ml_edge_vector(rij, R, offsets, idx_i, idx_j, e, rx, ry, rz);
    if (r_keep) { r_keep[0] = rx; r_keep[1] = ry; r_keep[2] = rz; }
    pr.ij = (int)(idx_i[e] - a0) | ((int)(idx_j[e] - a0) << 8) | (tid << 16);
    pr.d = sqrtf(rx * rx + ry * ry + rz * rz);
    spk_cutoff_eval_fast(cutoff, pr.d, pr.fc, pr.dfc);
    keep = !compact || pr.d < cutoff;
  }
  const unsigned long long bal = __ballot(keep);
  if (lane == 0) sScan[wv] = __popcll(bal);
  __syncthreads();
  int off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int c = sScan[w];
    if (w < wv) off += c;
    total += c;
  }
  const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
  if (keep) sP[pos] = pr;
  if (sMap && tid < np) sMap[tid] = keep ? (short)pos : (short)-1;
  __syncthreads();
  return total;
}

// register r of the half hi of a 32x32 accumulator holds row (r & 3) + 8 (r >> 2) + 4 hi
__device__ __forceinline__ int ml_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Packed LDS image of a weight matrix W[NOUT][K] (row-major, K padded with zeros to 8*KB):
//   P[((t * KB + ug) * 64 + lane) * 4 + v] = W[32 t + (lane & 31)][8 ug + 4 (lane >> 5) + v]
template <int NTHREADS, int SLOTS>
__device__ __forceinline__ void ml_stage_packed(float* dst, const float* __restrict__ w, int K, int KB, int tid) {
  constexpr int PER = (SLOTS + NTHREADS - 1) / NTHREADS;
  constexpr int BATCH = PER < 8 ? PER : 8;       // loads in flight per thread (bounds the live registers)
  const bool vec = (K & 3) == 0;
#pragma unroll 1
  for (int p0 = 0; p0 < PER; p0 += BATCH) {
    f32x4 v[BATCH];
#pragma unroll
    for (int p = 0; p < BATCH; ++p) {
      const int s = tid + (p0 + p) * NTHREADS;
      v[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (s < SLOTS) {
        const int lane = s & 63;
        const int ug = (s >> 6) % KB;
        const int t = (s >> 6) / KB;
        const int row = 32 * t + (lane & 31);
        const int k0 = 8 * ug + 4 * (lane >> 5);
        const float* src = w + (int64_t)row * K + k0;
        if (vec) {
          if (k0 < K) v[p] = *(const f32x4*)src;
        } else {
          if (k0 + 0 < K) v[p].x = src[0];
          if (k0 + 1 < K) v[p].y = src[1];
          if (k0 + 2 < K) v[p].z = src[2];
          if (k0 + 3 < K) v[p].w = src[3];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < BATCH; ++p) {
      const int s = tid + (p0 + p) * NTHREADS;
      if (s < SLOTS) *(f32x4*)(dst + (int64_t)s * 4) = v[p];
    }
  }
}

// the same image made once per weight version in global memory (wpack, spk_schnet_pack_weights_f32): staging is then a plain copy
__global__ void k_mol_pack_w2(const float* __restrict__ w2, float* __restrict__ image) {
  ml_stage_w2_split<256>((h16x8*)image, (h16x8*)image + 2048, w2, threadIdx.x);
}
int spk_schnet_mol_pack_w2(const float* w2, float* image, hipStream_t stream) {
  hipLaunchKernelGGL(k_mol_pack_w2, dim3(1), dim3(256), 0, stream, w2, image);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
template <int NTHREADS>
__device__ __forceinline__ void ml_stage_w2_image(float* __restrict__ dst, const float* __restrict__ img, int tid) {
  constexpr int PER = 4096 / NTHREADS;      // 16-byte pieces per thread (64 KB)
#pragma unroll 1
  for (int p0 = 0; p0 < PER; p0 += 8) {
    f32x4 v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) v[p] = *(const f32x4*)((const char*)img + (unsigned)((tid + (p0 + p) * NTHREADS) * 16));
#pragma unroll
    for (int p = 0; p < 8; ++p) *(f32x4*)(dst + (tid + (p0 + p) * NTHREADS) * 4) = v[p];
  }
}
// One 32-column output tile of a Dense layer over the 32 atom rows of the group, T-GEMM convention of spk_dense.hip:
// A = packed weights straight from L2 (16 k-blocks of 8, all requested up front), B = activations [32][ML_LD] in LDS;
// accumulator rows = output features 32 t + ml_row(r, hi), columns = atoms (lane & 31).
// Global accesses as  wave-uniform base (SGPR pair) + 32-bit per-lane byte offset: one VGPR per address instead of a 64-bit
// pair -- with 64-bit per-lane addresses the compiler hoists dozens of address pairs out of the loops and spills them.
template <class T>
__device__ __forceinline__ T ml_ld(const void* sbase, unsigned voff) { return *(const T*)((const char*)sbase + voff); }
template <class T>
__device__ __forceinline__ void ml_st(void* sbase, unsigned voff, T v) { *(T*)((char*)sbase + voff) = v; }

// gh[at][c] = sum over the directed edges of the row of `at`:  sSrc[neighbour][c] * g[pair][c] * f_c(pair)   (the transpose of
// the forward's row sum), as a task of one wavefront per atom.
// sEb: per directed edge (row of the saved filter tensor << 8 | local neighbour [| 1 << 24: pair beyond the cutoff], f_c of the pair).
// A lane owns FOUR channels (16-byte loads: 32 lanes span the 128 channels, a wave-load moves two whole filter rows) and the two
// halves of the wavefront take the even / odd entries of a row, meeting through one shuffle at the end.  RB entries per half are
// in flight at once, so a row of <= 2 RB neighbours costs ONE round trip to the saved filters (which the forward
// left in L2 / Infinity Cache).  Branch-free: entries beyond the end of a row re-read its last entry with weight 0.  Fixed
// summation order.  (The first form -- a lane per channel, 4-byte loads, four dependent round trips per task -- made the
// row-sum waves the stragglers of phase E: 18 k of its 79 k cycles.)
template <int RB>
__device__ __forceinline__ void ml_row_sums4(float* __restrict__ sDst, const float* __restrict__ sSrc, const float* __restrict__ g_g,
                                             const int2* __restrict__ sEb, const int* __restrict__ sRow, int at, int lane) {
  const int hi = lane >> 5;
  const unsigned c4 = 4u * (unsigned)(lane & 31);
  int rs = sRow[at] + hi;
  const int re = sRow[at + 1];
  const int last = re > 0 ? re - 1 : 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  while (true) {
    int2 rec[RB];
    f32x4 gv[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int idx = rs + 2 * u;
      rec[u] = sEb[idx < re ? idx : last];
      if (idx >= re) rec[u].y = 0;                 // weight 0 beyond the row
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) gv[u] = ml_ld<f32x4>(g_g, ((unsigned)((rec[u].x >> 8) & 0xFFFF) * 128u + c4) * 4u);
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const f32x4 sv = *(const f32x4*)(sSrc + (rec[u].x & 255) * ML_LD + c4);
      const float w = __int_as_float(rec[u].y);
      acc.x = fmaf(sv.x * w, gv[u].x, acc.x);
      acc.y = fmaf(sv.y * w, gv[u].y, acc.y);
      acc.z = fmaf(sv.z * w, gv[u].z, acc.z);
      acc.w = fmaf(sv.w * w, gv[u].w, acc.w);
    }
    rs += 2 * RB;
    if (rs - hi >= re) break;                      // (wave-uniform: both halves leave together)
  }
  acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
  acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
  if (hi == 0) *(f32x4*)(sDst + at * ML_LD + c4) = acc;
}

// (the weights do not depend on the data: ml_dense_load() is issued a phase EARLY -- before the barrier that completes the
// activations -- so that a Dense phase pays no L2 round trip)
__device__ __forceinline__ void ml_dense_load(f32x4 (&av)[16], const float* __restrict__ wp, int t /* wave-uniform */, int lane) {
  const char* sb = (const char*)wp + (size_t)t * (16 * 64 * 16);
#pragma unroll
  for (int u = 0; u < 16; ++u) av[u] = ml_ld<f32x4>(sb, (unsigned)(lane * 16 + u * 1024));
}
// (B operands are requested four k-blocks ahead of their MFMAs and pinned there: left alone the compiler issues every ds_read
//  right in front of the four MFMAs that need it and the matrix pipe waits ~100 cycles per group -- round 3, found on spk_painn_mol.hip)
#define ML_PIN() asm volatile("" ::: "memory")
// Split form of a Dense block (SP; spk_split.h): `av` then holds chunks of the SPLIT weight image (k_pack_weight_split: chunk 2 s = the high
// parts of k-step s, chunk 2 s + 1 its low parts -- same bytes, same offsets as the fp32 image), the activations are read from LDS
// as eight consecutive fp32 values per k-step and split in registers; three f16 instructions per 16 k instead of eight f32 ones.
#ifndef SP_AHEAD
#define SP_AHEAD 1      // k-steps of LDS operands requested ahead of their use in the split Dense blocks
#endif
template <int NS>
__device__ __forceinline__ f32x16 ml_dense_mma_sp(const f32x4* __restrict__ av, const float* __restrict__ brow /* lane's row + 8 hi + first k */, f32x16 acc) {
  f32x16 cx;
#pragma unroll
  for (int r = 0; r < 16; ++r) cx[r] = 0.f;
  f32x4 b0[NS], b1[NS];
#pragma unroll
  for (int s = 0; s < SP_AHEAD && s < NS; ++s) { b0[s] = *(const f32x4*)(brow + 16 * s); b1[s] = *(const f32x4*)(brow + 16 * s + 4); }
  ML_PIN();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (s + SP_AHEAD < NS) { b0[s + SP_AHEAD] = *(const f32x4*)(brow + 16 * (s + SP_AHEAD)); b1[s + SP_AHEAD] = *(const f32x4*)(brow + 16 * (s + SP_AHEAD) + 4); ML_PIN(); }
    h16x4 h0, l0, h1, l1;
    sp_split4(b0[s], h0, l0);
    sp_split4(b1[s], h1, l1);
    SP_STEP(__builtin_bit_cast(h16x8, av[2 * s]), __builtin_bit_cast(h16x8, av[2 * s + 1]), sp_cat(h0, h1), sp_cat(l0, l1), acc, cx);
  }
  SP_FOLD(acc, cx);
  return acc;
}
template <bool SP = false>
__device__ __forceinline__ f32x16 ml_dense_mma(const f32x4 (&av)[16], const float* __restrict__ sIn, int lane, f32x16 acc) {
  const int hi = lane >> 5, el = lane & 31;
  if constexpr (SP) return ml_dense_mma_sp<8>(av, sIn + el * ML_LD + 8 * hi, acc);
  const float* brow = sIn + el * ML_LD + 4 * hi;
  f32x4 bv[16];
#pragma unroll
  for (int u = 0; u < 4; ++u) bv[u] = *(const f32x4*)(brow + 8 * u);
  ML_PIN();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    acc = ML_MFMA(av[u].x, bv[u].x, acc);
    acc = ML_MFMA(av[u].y, bv[u].y, acc);
    acc = ML_MFMA(av[u].z, bv[u].z, acc);
    acc = ML_MFMA(av[u].w, bv[u].w, acc);
    if (u + 4 < 16) { bv[u + 4] = *(const f32x4*)(brow + 8 * (u + 4)); ML_PIN(); }
  }
  return acc;
}
// the same in halves of 8 k-blocks: the first half is requested a phase early (32 VGPRs across the phase in between), the
// second half at the start of the phase -- its latency hides behind the 32 MFMAs of the first half
__device__ __forceinline__ void ml_dense_load8(f32x4 (&av)[8], const float* __restrict__ wp, int t /* wave-uniform */, int lane, int half) {
  const char* sb = (const char*)wp + (size_t)t * (16 * 64 * 16) + (size_t)half * (8 * 1024);
#pragma unroll
  for (int u = 0; u < 8; ++u) av[u] = ml_ld<f32x4>(sb, (unsigned)(lane * 16 + u * 1024));
}
template <bool SP = false>
__device__ __forceinline__ f32x16 ml_dense_mma8(const f32x4 (&av)[8], const float* __restrict__ sIn, int lane, int half, f32x16 acc) {
  const int hi = lane >> 5, el = lane & 31;
  if constexpr (SP) return ml_dense_mma_sp<4>(av, sIn + el * ML_LD + 8 * hi + 64 * half, acc);
  const float* brow = sIn + el * ML_LD + 4 * hi + 64 * half;
  f32x4 bv[8];
#pragma unroll
  for (int u = 0; u < 4; ++u) bv[u] = *(const f32x4*)(brow + 8 * u);
  ML_PIN();
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    acc = ML_MFMA(av[u].x, bv[u].x, acc);
    acc = ML_MFMA(av[u].y, bv[u].y, acc);
    acc = ML_MFMA(av[u].z, bv[u].z, acc);
    acc = ML_MFMA(av[u].w, bv[u].w, acc);
    if (u + 4 < 8) { bv[u + 4] = *(const f32x4*)(brow + 8 * (u + 4)); ML_PIN(); }
  }
  return acc;
}
template <bool SP = false>
__device__ __forceinline__ f32x16 ml_dense_tile(const float* __restrict__ wp, const float* __restrict__ sIn, int t, int lane, f32x16 acc) {
  f32x4 av[16];
  ml_dense_load(av, wp, t, lane);
  return ml_dense_mma<SP>(av, sIn, lane, acc);
}

// B operand of a Dense half-tile that is the SUM of two LDS tiles (the two teams' partial cfconv outputs)
template <bool SP = false>
__device__ __forceinline__ f32x16 ml_dense_mma8_sum(const f32x4 (&av)[8], const float* __restrict__ sIn0, const float* __restrict__ sIn1, int lane, int half,
                                                    f32x16 acc) {
  const int hi = lane >> 5, el = lane & 31;
  if constexpr (SP) {
    const int o = el * ML_LD + 8 * hi + 64 * half;
    f32x16 cx;
#pragma unroll
    for (int r = 0; r < 16; ++r) cx[r] = 0.f;
    f32x4 p0[4], p1[4], q0[4], q1[4];
#pragma unroll
    for (int s = 0; s < SP_AHEAD; ++s) {
      p0[s] = *(const f32x4*)(sIn0 + o + 16 * s); p1[s] = *(const f32x4*)(sIn0 + o + 16 * s + 4);
      q0[s] = *(const f32x4*)(sIn1 + o + 16 * s); q1[s] = *(const f32x4*)(sIn1 + o + 16 * s + 4);
    }
    ML_PIN();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + SP_AHEAD < 4) {
        p0[s + SP_AHEAD] = *(const f32x4*)(sIn0 + o + 16 * (s + SP_AHEAD)); p1[s + SP_AHEAD] = *(const f32x4*)(sIn0 + o + 16 * (s + SP_AHEAD) + 4);
        q0[s + SP_AHEAD] = *(const f32x4*)(sIn1 + o + 16 * (s + SP_AHEAD)); q1[s + SP_AHEAD] = *(const f32x4*)(sIn1 + o + 16 * (s + SP_AHEAD) + 4);
        ML_PIN();
      }
      h16x4 h0, l0, h1, l1;
      sp_split4(p0[s] + q0[s], h0, l0);
      sp_split4(p1[s] + q1[s], h1, l1);
      SP_STEP(__builtin_bit_cast(h16x8, av[2 * s]), __builtin_bit_cast(h16x8, av[2 * s + 1]), sp_cat(h0, h1), sp_cat(l0, l1), acc, cx);
    }
    SP_FOLD(acc, cx);
    return acc;
  }
  const int off = el * ML_LD + 4 * hi + 64 * half;
  f32x4 b0[8], b1[8];
#pragma unroll
  for (int u = 0; u < 4; ++u) { b0[u] = *(const f32x4*)(sIn0 + off + 8 * u); b1[u] = *(const f32x4*)(sIn1 + off + 8 * u); }
  ML_PIN();
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    acc = ML_MFMA(av[u].x, b0[u].x + b1[u].x, acc);
    acc = ML_MFMA(av[u].y, b0[u].y + b1[u].y, acc);
    acc = ML_MFMA(av[u].z, b0[u].z + b1[u].z, acc);
    acc = ML_MFMA(av[u].w, b0[u].w + b1[u].w, acc);
    if (u + 4 < 8) { b0[u + 4] = *(const f32x4*)(sIn0 + off + 8 * (u + 4)); b1[u + 4] = *(const f32x4*)(sIn1 + off + 8 * (u + 4)); ML_PIN(); }
  }
  return acc;
}

// Forward, third form.  The 8 waves are two TEAMS of four (one wave per SIMD each); wave t of a team owns channel tile t.
// Per pair tile the team computes the hidden layer ONCE -- wave t its 32 hidden channels (12 MFMAs + 16 softplus per lane
// instead of 48 + 64) -- and shares it through one LDS tile; every wave then runs GEMM 2 for its channel tile with the
// hidden activations as A operand from LDS, writes the raw filter outputs (the tensor the backward reads), and accumulates
//     y[a, c] += sum_pairs ( [i = a] h[j, c] + [j = a] h[i, c] ) W[pair, c]
// ON THE MATRIX CORE: A = the 0/1 incidence of the pair tile (rows = atoms), B = the modulated products (rows = pairs,
// columns = channels), 32 MFMAs per tile and wave, accumulator = 16 registers that live across all tiles of the wave.
// No atomics, no re-read of the filters, no scatter pass; team 1 runs in2f while team 0 starts the first tile.
template <int KPB, bool SP>
__global__ __launch_bounds__(512) void k_schnet_mol_fwd(MolFwdArgs a) {
  constexpr int NF = 128, KB2 = 16;
  constexpr int W1F = SP ? 2 * MlW1Image<KPB>::BYTES / 4 : NF * KPB * 8;     // floats of the W1 image(s)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW2 = smem;                                  // NF*NF  (SP: high image | low image, 32 KB each)
  float* sW1 = sW2 + NF * NF;                         // NF*KPB*8  (SP: high image | low image)
  float* sb1 = sW1 + W1F;                             // NF
  h16x8* const sW2h = (h16x8*)sW2;
  h16x8* const sW2l = sW2h + 2048;
  char* const sW1h = (char*)sW1;
  char* const sW1l = sW1h + MlW1Image<KPB>::BYTES;
  float* sb2 = sb1 + NF;                              // NF
  float* sX = sb2 + NF;                               // [32][ML_LD] atom features x_l
  float* sH = sX + 32 * ML_LD;                        // h = in2f(x); later the hidden layer of f2out
  float* sY = sH + 32 * ML_LD;                        // team 0: hidden activations of its pair tile, then its partial y
  float* sT = sY + 32 * ML_LD;                        // team 1: the same
  MolPair* sP = (MolPair*)(sT + 32 * ML_LD);          // per pair: local atoms, d, f_c, f_c'
  float* sRb = (float*)(sP + ML_MAXPAIRS);            // [2][32] radial basis parameters
  int* sScan = (int*)(sRb + 64);                      // [8] per-wave counts of the pair compaction

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wv: SGPR
  const int hi = lane >> 5, el = lane & 31;
  const int team = wv >> 2, t = wv & 3;
  if (tid < 64) {
    const int k = tid & 31;
    const float* src = (tid < 32) ? a.rb.p0 : a.rb.p1;
    sRb[tid] = (src && k < a.rb.n_rbf) ? src[k] : 1.0f;
  }

  for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
    const int a0 = a.grp_atom0[grp], na = a.grp_atom0[grp + 1] - a0;
    const int p0 = a.grp_pair0[grp], np_list = a.grp_pair0[grp + 1] - p0;
    __syncthreads();   // the previous group is done with every LDS buffer
    ML_STAMP(0);

    // ---- group set-up: features, pair geometry (shared by all interactions), first filter weights
    for (int s = tid; s < 32 * 32; s += 512) {
      const int row = s >> 5, c4 = s & 31;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < na) {
        if (a.x0) v = ml_ld<f32x4>(a.x0 + (size_t)a0 * NF, (unsigned)(s * 16));
        else {
          const long long z = a.Z[a0 + row];
          if (z >= 0 && z < a.n_types) v = *(const f32x4*)(a.emb + (size_t)z * NF + 4 * c4);
          else { const float qn = __builtin_nanf(""); v = f32x4{qn, qn, qn, qn}; }     // no such row (nn.Embedding raises): NaN, never out of bounds
        }
      }
      *(f32x4*)(sX + row * ML_LD + 4 * c4) = v;
    }
    const int np = ml_pair_records(sP, nullptr, sScan, a.half, a.rij, a.R, a.offsets, a.idx_i, a.idx_j, p0, np_list, a0, a.rb.cutoff, a.compact != 0, tid);
    const int ntile = (np + 31) / 32;
    // (W2 of the first interaction -- first read by GEMM 2 of the first tile -- is staged by team 0 behind its first hidden tile,
    // while team 1 is still busy with in2f)
    if constexpr (SP) ml_stage_w1_split<KPB>(sW1h, sW1l, a.L[0].w1, a.rb.n_rbf, tid);
    else ml_stage_packed<512, NF * KPB * 2>(sW1, a.L[0].w1, a.rb.n_rbf, KPB, tid);
    if (tid < NF) { sb1[tid] = a.L[0].b1[tid]; sb2[tid] = a.L[0].b2[tid]; }

    for (int l = 0; l < a.n_layers; ++l) {
      const MolLayerDev& P = a.L[l];
      float* h_g = a.saved + (int64_t)l * a.N * (2 * NF);
      float* pre3_g = h_g + a.N * (int64_t)NF;
      float* g_g = a.gbase + (int64_t)l * a.gsz + (int64_t)p0 * NF;
      __syncthreads();
      ML_STAMP(1 + 5 * l);

      // ================= phase A: pair tiles, team 0 takes tiles 0, 2, 4, ..., team 1 in2f and then tiles 1, 3, ...
      float* zbuf = team ? sT : sY;
      f32x16 yacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc[r] = 0.f;
      const int n0 = (ntile + 1) / 2, n1 = 1 + ntile / 2;
      const int n_iter = n0 > n1 ? n0 : n1;
      for (int it = 0; it < n_iter; ++it) {
        const int tile = team == 0 ? 2 * it : 2 * it - 1;
        const bool in2f = (team == 1 && it == 0);
        const bool active = !in2f && tile < ntile;
        const int pfirst = 32 * tile;
        const int nvalid = active ? ((np - pfirst) < 32 ? (np - pfirst) : 32) : 0;
        if (in2f) {
          // ---- h[:, 32t : 32t+32] = x W_in^T  (kept in LDS for the modulation, saved for the backward)
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          acc = ml_dense_tile<false>(P.in2f_p, sX, t, lane, acc);     // (fp32 image: this tile runs beside the other team's first pair tile, off the
                                                                        //  critical path, and its split form pushed the kernel 170 B/lane further into scratch)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 hv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *(f32x4*)(sH + el * ML_LD + 32 * t + 8 * q + 4 * hi) = hv;
            if (el < na) ml_st<f32x4>(h_g + (size_t)a0 * NF + 32 * t, (unsigned)((el * NF + 8 * q + 4 * hi) * 4), hv);
          }
        } else if (active) {
          // ---- this wave's quarter of the hidden layer: z[pairs][32t : 32t+32] = ssp(W1 phi + b1)
          const float d = sP[pfirst + (el < nvalid ? el : (nvalid - 1))].d;      // lanes 32..63 mirror lanes 0..31
          f32x16 zc;
#pragma unroll
          for (int r = 0; r < 16; ++r) zc[r] = sb1[32 * t + ml_row(r, hi)];
          if constexpr (SP) {
            // split form: A = the (high, low) images of W1, B = this pair's basis values in the lane's k-slots; three f16 matrix
            // instructions per k-step, the cross terms in their own accumulator
            h16x8 ph[2], pl[2], dh_[2], dl_[2];
            ml_basis_split<KPB, false>(a.rb.kind, a.rb.n_rbf, sRb, sRb + 32, hi, d, ph, pl, dh_, dl_);
            f32x16 zx;
#pragma unroll
            for (int r = 0; r < 16; ++r) zx[r] = 0.f;
#pragma unroll
            for (int s = 0; s < (KPB > 2 ? 2 : 1); ++s) {
              h16x8 wh, wl;
              ml_w1_operand<KPB>(sW1h, sW1l, s, t * 64 + lane, wh, wl);
              SP_STEP(wh, wl, ph[s], pl[s], zc, zx);
            }
            SP_FOLD(zc, zx);
            // the hidden tile in LDS as the split A operand of GEMM 2: row = pair, [high: 128 halves | low: 128 halves], contraction
            // index in accumulator order -- the lane's registers 8 s' .. 8 s' + 7 are slot (2 t + s', hi) as they lie
            char* zr = (char*)zbuf + el * (ML_LD * 4) + 16 * hi;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = spk_fast_ssp(zc[8 * sp + e]);
              h16x8 h, l2;
              sp_split8(v, h, l2);
              *(h16x8*)(zr + 32 * (2 * t + sp)) = h;
              *(h16x8*)(zr + 256 + 32 * (2 * t + sp)) = l2;
            }
          } else {
#pragma unroll
          for (int u = 0; u < KPB; ++u) {
            const f32x4 wq = *(const f32x4*)(sW1 + ((t * KPB + u) * 64 + lane) * 4);
            float ph[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              float dp;
              ml_rbf(a.rb.kind, a.rb.n_rbf, sRb, sRb + 32, 8 * u + 4 * hi + v, d, ph[v], dp);
            }
            zc = ML_MFMA(wq.x, ph[0], zc);
            zc = ML_MFMA(wq.y, ph[1], zc);
            zc = ML_MFMA(wq.z, ph[2], zc);
            zc = ML_MFMA(wq.w, ph[3], zc);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(f32x4*)(zbuf + el * ML_LD + 32 * t + 8 * q + 4 * hi) =
                f32x4{spk_fast_ssp(zc[4 * q]), spk_fast_ssp(zc[4 * q + 1]), spk_fast_ssp(zc[4 * q + 2]), spk_fast_ssp(zc[4 * q + 3])};
          }
        }
        if (l == 0 && it == 0 && team == 0) {
          if constexpr (SP) { if (a.L[0].w2_img) ml_stage_w2_image<256>(sW2, a.L[0].w2_img, tid); else ml_stage_w2_split<256>(sW2h, sW2l, a.L[0].w2, tid); }
          else ml_stage_packed<256, NF * NF / 4>(sW2, a.L[0].w2, NF, KB2, tid);
        }
        __syncthreads();     // the team's hidden tile (and, in the first round, h) is complete
        if (active) {
          // ---- GEMM 2, operands swapped (rows = pairs, columns = channels 32 t + el): g = W2 z + b2, A operand from LDS
          f32x16 g;
          const int c0 = 32 * t + el;
          const float bias2 = sb2[c0];
#pragma unroll
          for (int r = 0; r < 16; ++r) g[r] = bias2;
          if constexpr (SP) {
            // A = hidden tile (rows = pairs) from LDS, B = W2 image (columns = channels): both split, requested two k-steps ahead
            f32x16 gx;
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[r] = 0.f;
            const h16x8* wbh = sW2h + (t * 8) * 64 + lane;
            const h16x8* wbl = sW2l + (t * 8) * 64 + lane;
            const char* zr = (const char*)zbuf + el * (ML_LD * 4) + 16 * hi;
            h16x8 zh[8], zl[8], wh[8], wl[8];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              zh[s] = *(const h16x8*)(zr + 32 * s); zl[s] = *(const h16x8*)(zr + 256 + 32 * s);
              wh[s] = wbh[s * 64]; wl[s] = wbl[s * 64];
            }
            ML_PIN();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              if (s + 2 < 8) {
                zh[s + 2] = *(const h16x8*)(zr + 32 * (s + 2)); zl[s + 2] = *(const h16x8*)(zr + 256 + 32 * (s + 2));
                wh[s + 2] = wbh[(s + 2) * 64]; wl[s + 2] = wbl[(s + 2) * 64];
                ML_PIN();
              }
              SP_STEP(zh[s], zl[s], wh[s], wl[s], g, gx);
            }
            SP_FOLD(g, gx);
          } else {
            const float* wbase = sW2 + ((int64_t)t * KB2 * 64 + lane) * 4;
            const float* zrow = zbuf + el * ML_LD + 4 * hi;
            // both operands come from LDS: requested THREE k-blocks ahead and pinned there (the one-ahead form written here
            // before was collapsed by the compiler to "two ds_reads, wait for both, four MFMAs": the LDS round trip of every
            // k-block sat in the MFMA chain)
            f32x4 wb[KB2], zb[KB2];
#pragma unroll
            for (int ug = 0; ug < 3; ++ug) { wb[ug] = *(const f32x4*)(wbase + ug * 256); zb[ug] = *(const f32x4*)(zrow + 8 * ug); }
            ML_PIN();
#pragma unroll
            for (int ug = 0; ug < KB2; ++ug) {
              if (ug + 3 < KB2) { wb[ug + 3] = *(const f32x4*)(wbase + (ug + 3) * 256); zb[ug + 3] = *(const f32x4*)(zrow + 8 * (ug + 3)); ML_PIN(); }
              g = ML_MFMA(zb[ug].x, wb[ug].x, g);
              g = ML_MFMA(zb[ug].y, wb[ug].y, g);
              g = ML_MFMA(zb[ug].z, wb[ug].z, g);
              g = ML_MFMA(zb[ug].w, wb[ug].w, g);
            }
          }
          // raw filter outputs for the backward (row = position of the pair in the list, 128-byte row segments per half wave)
          // ---- and modulation + accumulation on the matrix core: y[atom][c0] += [i = atom] W h[j][c0] + [j = atom] W h[i][c0]
          float* gt = g_g + 32 * t;
          if constexpr (SP) {
            // incidence product on the f16 matrix instruction: A (0 / 1, exact) carries the weight 2^-11 of the low parts itself, so
            // the accumulator that lives across the tiles stays ONE
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              float tI[8], tJ[8];
              h16x8 ai, aj;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int r = 8 * s2 + e;
                const int pr = ml_row(r, hi);
                const bool ok = pr < nvalid;
                const MolPair rec = sP[pfirst + (ok ? pr : 0)];
                const int pi = rec.ij & 255, pj = (rec.ij >> 8) & 255;
                if (ok) ml_st<float>(gt, (unsigned)(((rec.ij >> 16) * NF + el) * 4), g[r]);
                const float W = ok ? g[r] * rec.fc : 0.f;
                tI[e] = W * sH[pj * ML_LD + c0]; tJ[e] = W * sH[pi * ML_LD + c0];
                ai[e] = pi == el ? (_Float16)1.0f : (_Float16)0.0f;
                aj[e] = pj == el ? (_Float16)1.0f : (_Float16)0.0f;
              }
              h16x8 ih, il, jh, jl;
              sp_split8(tI, ih, il);
              sp_split8(tJ, jh, jl);
              const h16x8 ais = ai * (_Float16)SP_DOWN, ajs = aj * (_Float16)SP_DOWN;
              yacc = SP_MFMA(ai, ih, yacc);
              yacc = SP_MFMA(ais, il, yacc);
              yacc = SP_MFMA(aj, jh, yacc);
              yacc = SP_MFMA(ajs, jl, yacc);
            }
          } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int pr = ml_row(r, hi);
            const bool ok = pr < nvalid;
            const MolPair rec = sP[pfirst + (ok ? pr : 0)];
            const int pi = rec.ij & 255, pj = (rec.ij >> 8) & 255;
            if (ok) ml_st<float>(gt, (unsigned)(((rec.ij >> 16) * NF + el) * 4), g[r]);
            const float W = ok ? g[r] * rec.fc : 0.f;
            const float tI = W * sH[pj * ML_LD + c0], tJ = W * sH[pi * ML_LD + c0];
            yacc = ML_MFMA(pi == el ? 1.0f : 0.0f, tI, yacc);
            yacc = ML_MFMA(pj == el ? 1.0f : 0.0f, tJ, yacc);
          }
          }
        }
        __syncthreads();     // every wave of the team is done with the hidden tile
      }
      ML_STAMP(2 + 5 * l);
      // the two teams' partial sums: rows = atoms ml_row(r, hi), columns = channels 32 t + el
#pragma unroll
      for (int r = 0; r < 16; ++r) zbuf[ml_row(r, hi) * ML_LD + 32 * t + el] = yacc[r];
      __syncthreads();
      ML_STAMP(4 + 5 * l);

      // ================= phase C1: pre3 = (y0 + y1) W3^T + b3 (saved), hidden = ssp(pre3) -> sH; the other team stages weights
      f32x4 avC[8];        // team 0: first half of the weight tile of phase C2, requested BEFORE the stores of pre3 -- loads and
                           // stores share one in-order counter, a load issued behind a store cannot be waited for without it
      if (team == 0) {
        f32x4 avA[8], avB[8];
        ml_dense_load8(avA, P.o1_p, t, lane, 0);
        ml_dense_load8(avB, P.o1_p, t, lane, 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = P.o1_b[32 * t + ml_row(r, hi)];
        acc = ml_dense_mma8_sum<SP>(avA, sY, sT, lane, 0, acc);
        acc = ml_dense_mma8_sum<SP>(avB, sY, sT, lane, 1, acc);
        ml_dense_load8(avC, P.o2_p, t, lane, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 pv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          if (el < na) ml_st<f32x4>(pre3_g + (size_t)a0 * NF + 32 * t, (unsigned)((el * NF + 8 * q + 4 * hi) * 4), pv);
          *(f32x4*)(sH + el * ML_LD + 32 * t + 8 * q + 4 * hi) = f32x4{spk_fast_ssp(pv.x), spk_fast_ssp(pv.y), spk_fast_ssp(pv.z), spk_fast_ssp(pv.w)};
        }
      } else if (l + 1 < a.n_layers) {     // the filter GEMMs of this interaction are done: their LDS images can be replaced
        const int t2 = tid - 256;
        if constexpr (SP) {
          if (a.L[l + 1].w2_img) ml_stage_w2_image<256>(sW2, a.L[l + 1].w2_img, t2); else ml_stage_w2_split<256>(sW2h, sW2l, a.L[l + 1].w2, t2);
          ml_stage_w1_split<KPB>(sW1h, sW1l, a.L[l + 1].w1, a.rb.n_rbf, t2);
        } else {
          ml_stage_packed<256, NF * NF / 4>(sW2, a.L[l + 1].w2, NF, KB2, t2);
          ml_stage_packed<256, NF * KPB * 2>(sW1, a.L[l + 1].w1, a.rb.n_rbf, KPB, t2);
        }
        if (t2 < NF) { sb1[t2] = a.L[l + 1].b1[t2]; sb2[t2] = a.L[l + 1].b2[t2]; }
      }
      __syncthreads();
      ML_STAMP(5 + 5 * l);

      // ================= phase C2: x += hidden W4^T + b4
      if (team == 0) {
        f32x4 avB[8];
        ml_dense_load8(avB, P.o2_p, t, lane, 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = P.o2_b[32 * t + ml_row(r, hi)];
        acc = ml_dense_mma8<SP>(avC, sH, lane, 0, acc);
        acc = ml_dense_mma8<SP>(avB, sH, lane, 1, acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float* xp = sX + el * ML_LD + 32 * t + 8 * q + 4 * hi;
          f32x4 xv = *(const f32x4*)xp;
          xv.x += acc[4 * q]; xv.y += acc[4 * q + 1]; xv.z += acc[4 * q + 2]; xv.w += acc[4 * q + 3];
          if (el >= na) xv = f32x4{0.f, 0.f, 0.f, 0.f};       // padding rows stay zero (in2f has no bias: h pads stay zero too)
          *(f32x4*)xp = xv;
          if (l + 1 == a.n_layers && el < na) ml_st<f32x4>(a.x_out + (size_t)a0 * NF + 32 * t, (unsigned)((el * NF + 8 * q + 4 * hi) * 4), xv);
        }
      }
      // (the barrier at the top of the next interaction / group closes this phase)
    }
    if (a.head.w1) {
      // ================= energy head on the atom tile that is still in LDS: y = w2 . act(W1 x + b1) + b2, E[mol] += sum_atoms y
      const MolHeadDev& Hd = a.head;
      const int HT = Hd.H / 32;
      long long my_mol = -1;
      if (wv == 1 && lane < 32) my_mol = lane < na ? Hd.idx_m[a0 + lane] : -2;      // wave 1: molecule id of atom `lane`
      __syncthreads();                    // x_L is complete
      const int hw = wv;
      if (hw < HT) {
        f32x4 av[16];                     // (requested after the barrier: register arrays that live across one get spilled)
#pragma unroll
        for (int u = 0; u < 16; ++u) av[u] = ml_ld<f32x4>(Hd.w1 + (size_t)(32 * hw) * NF, (unsigned)((el * NF + 8 * u + 4 * hi) * 4));
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = Hd.b1[32 * hw + ml_row(r, hi)];
        acc = ml_dense_mma(av, sX, lane, acc);
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 pv = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
          if (el < na) ml_st<f32x4>(Hd.pre_h + (size_t)a0 * Hd.H + 32 * hw, (unsigned)((el * Hd.H + 8 * q + 4 * hi) * 4), pv);
          const f32x4 wv2 = *(const f32x4*)(Hd.w2 + 32 * hw + 8 * q + 4 * hi);
          if (Hd.act == SPK_ACT_SILU)
            part += pv.x * spk_sigmoid(pv.x) * wv2.x + pv.y * spk_sigmoid(pv.y) * wv2.y + pv.z * spk_sigmoid(pv.z) * wv2.z + pv.w * spk_sigmoid(pv.w) * wv2.w;
          else
            part += spk_ssp(pv.x) * wv2.x + spk_ssp(pv.y) * wv2.y + spk_ssp(pv.z) * wv2.z + spk_ssp(pv.w) * wv2.w;
        }
        part += __shfl_xor(part, 32, 64);
        if (hi == 0) sH[hw * 32 + el] = part;          // (sH: the hidden tile of the last f2out is no longer needed)
      }
      __syncthreads();
      // wave 1 holds the molecule id of atom (lane) in my_mol: segment heads add their run, one atomic per (group, molecule)
      if (wv == 1) {
        float y = 0.f;
        if (lane < 32) {
          y = Hd.b2 ? Hd.b2[0] : 0.f;
          for (int w = 0; w < HT; ++w) y += sH[w * 32 + lane];
          if (lane >= na) y = 0.f;
        }
        const long long first_mol = __shfl(my_mol, 0, 64);
        if (__all(lane >= na || my_mol == first_mol)) {
          // the whole group is one molecule (the usual case): a butterfly over the 32 atom lanes (y is 0 beyond na)
          float sum = y;
#pragma unroll
          for (int m = 16; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
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
    ML_STAMP(31);
  }
}

// ------------------------------------------------------------------------------------------ host side
static long long* g_mol_dbg = nullptr;
// tuning aid (scripts/mol_timing.py; include/spk_hip.h): device buffer of int64 that receives cycle stamps -- entries
// [0, 128): phases of thread 0 of workgroup 0; [128 + 4 b, 128 + 4 b + 4): real-time and cycle stamps at the start / end of workgroup b
// of the backward launch, so the buffer must hold 128 + 4 * (number of groups) entries.  NULL: off (production)
extern "C" void spk_schnet_mol_set_debug_buffer(void* p) { g_mol_dbg = (long long*)p; }

static size_t mol_w1_floats(int kpb, bool sp) {      // LDS floats of the staged W1 image(s): MlW1Image<KPB>::BYTES twice in the split form
  return sp ? (size_t)2 * (4096 + (kpb > 2 ? (kpb == 4 ? 4096 : 2048) : 0)) / 4 : (size_t)128 * kpb * 8;
}
static size_t mol_fwd_lds(int kpb, bool sp) {
  return (size_t)(128 * 128 + mol_w1_floats(kpb, sp) + 2 * 128 + 4 * 32 * ML_LD + 64 + 8) * sizeof(float) + ML_MAXPAIRS * sizeof(MolPair);
}

// Shapes / lists the molecule-resident kernels cover (everything else runs the general driver of spk_schnet.hip).
bool spk_schnet_mol_eligible(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb) {
  const int variant = spk_get_variant();
  if (variant != SPK_VARIANT_AUTO && variant != SPK_VARIANT_MFMA_PAIR && variant != SPK_VARIANT_MFMA) return false;
  if (getenv("SPK_NO_MOL")) return false;
  if (m->n_atom_basis != 128 || m->n_filters != 128 || m->n_interactions < 1 || m->n_interactions > ML_MAXL || !m->wpack) return false;
  const int kpb = (rb->n_rbf + 7) / 8;
  if (kpb < 1 || kpb > 4) return false;
  if (!(g->symmetric && g->sorted && g->half && g->rev && g->edge_pair && g->rowptr && g->n_half > 0)) return false;
  if (g->n_groups <= 0 || !g->grp_atom0 || !g->grp_pair0 || g->max_group_atoms > 32) return false;
  if (g->max_group_pairs <= 0 || g->max_group_pairs > ML_MAXPAIRS) return false;
  if (g->n_half_dev) return false;       // a per-call compacted pair list is already in place
  // (skin lists, g->filter_pairs: pairs beyond the cutoff carry f_c = f_c' = 0 and contribute exactly zero here -- no compaction)
  return true;
}

template <int KPB, bool SP>
static int launch_mol_fwd_sp(const MolFwdArgs& a, hipStream_t stream);
template <int KPB>
static int launch_mol_fwd(const MolFwdArgs& a, hipStream_t stream) {
  return spk_get_split() ? launch_mol_fwd_sp<KPB, true>(a, stream) : launch_mol_fwd_sp<KPB, false>(a, stream);
}
template <int KPB, bool SP>
static int launch_mol_fwd_sp(const MolFwdArgs& a, hipStream_t stream) {
  const size_t lds = mol_fwd_lds(KPB, SP);
  auto kern = k_schnet_mol_fwd<KPB, SP>;
  static SpkPerDevice attr_set;
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_dev);
  }
  int grid = a.n_groups;
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  SpkProfScope prof("schnet_mol_fwd", stream);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// `saved` as laid out by spk_schnet_saved_floats_graph(): L x (h | pre3), then L x gsz floats of raw filter outputs.
int spk_schnet_mol_forward_ex(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                              const float* x0, const float* r_ij, const float* R, const float* offsets, const MolHeadDev* head, float* x_out,
                              float* saved, int64_t gsz, hipStream_t stream, const float* emb = nullptr, const int64_t* Z = nullptr, int n_types = 0);
int spk_schnet_mol_forward(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                           const float* x0, const float* r_ij, float* x_out, float* saved, int64_t gsz, hipStream_t stream) {
  return spk_schnet_mol_forward_ex(m, g, rb, ptab, x0, r_ij, nullptr, nullptr, nullptr, x_out, saved, gsz, stream);
}
int spk_schnet_mol_forward_ex(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                              const float* x0, const float* r_ij, const float* R, const float* offsets, const MolHeadDev* head, float* x_out,
                              float* saved, int64_t gsz, hipStream_t stream, const float* emb, const int64_t* Z, int n_types) {
  MolFwdArgs a;
  a.R = R; a.offsets = offsets;
  a.emb = emb; a.Z = Z; a.n_types = n_types;
  if (head) a.head = *head; else a.head.w1 = nullptr;
  const bool split = spk_get_split() != 0;     // the launch takes the same decision (launch_mol_fwd): split images for the split kernels
  a.n_layers = m->n_interactions;
  for (int l = 0; l < m->n_interactions; ++l) {
    const spk_schnet_layer_t& P = m->layers[l];
    MolLayerDev& D = a.L[l];
    D.in2f_p = spk_packed_of(ptab, P.in2f_w, 0);
    D.o1_p = split ? spk_packed_split_of(ptab, P.f2out_w1, 0) : spk_packed_of(ptab, P.f2out_w1, 0);
    D.o2_p = split ? spk_packed_split_of(ptab, P.f2out_w2, 0) : spk_packed_of(ptab, P.f2out_w2, 0);
    SPK_CHECK_ARG(D.in2f_p && D.o1_p && D.o2_p, "spk_schnet_mol_forward: packed weight images missing");
    D.w1 = P.fn_w1; D.b1 = P.fn_b1; D.w2 = P.fn_w2; D.b2 = P.fn_b2; D.o1_b = P.f2out_b1; D.o2_b = P.f2out_b2;
    D.w2_img = split ? ptab.extra_of(P.fn_w2) : nullptr;
  }
  a.x0 = x0; a.x_out = x_out; a.rij = r_ij; a.idx_i = g->idx_i; a.idx_j = g->idx_j;
  a.half = g->half; a.rowptr = g->rowptr; a.edge_pair = g->edge_pair; a.grp_atom0 = g->grp_atom0; a.grp_pair0 = g->grp_pair0;
  a.n_groups = g->n_groups; a.saved = saved; a.N = g->n_atoms; a.gsz = gsz;
  a.gbase = saved + (int64_t)m->n_interactions * g->n_atoms * (m->n_filters + m->n_atom_basis);
  a.rb = spk_radial_dev(rb);
  a.compact = spk_schnet_mol_bwd_eligible(m, g, rb) ? 1 : 0;     // only when the molecule-resident backward (which compacts too) will consume `saved`
  a.dbg = g_mol_dbg;
  switch ((rb->n_rbf + 7) / 8) {
    case 1: return launch_mol_fwd<1>(a, stream);
    case 2: return launch_mol_fwd<2>(a, stream);
    case 3: return launch_mol_fwd<3>(a, stream);
    case 4: return launch_mol_fwd<4>(a, stream);
  }
  spk_set_error("spk_schnet_mol_forward: n_rbf = %d unsupported", rb->n_rbf);
  return SPK_ERR_ARG;
}

// ==========================================================================================================
// Backward (first order, what Forces asks for): dL/dr_ij and optionally dL/dx0 from dL/dx_L, one launch.
//
// Per interaction, last to first (notation of SURVEY.md Appendix B; everything below is local to the group):
//   D1. gt = (gx W4) * ssp'(pre3)          D2. gy = gt W3                                (two T-GEMM phases)
//   E.  task queue:  (pair tile, channel-tile pair) derivative tasks  +  row-sum tasks  +  the four tiles of G
//         derivative task: phi, phi' -> GEMM 1 value and derivative (a, a') -> z' = sigmoid(a) a' -> GEMM 2' (rows =
//           channels, columns = pairs: lane = pair) -> D = g' f_c + g f_c' with the SAVED raw filter outputs g ->
//           s1 = sum_c gy_i h_j D,  s2 = sum_c gy_j h_i D  (in-lane sums over the 16 channels a lane owns, one LDS add per
//           pair and task); the per-pair sums are kept in LDS across all interactions and become dL/dr once, at the end
//         row-sum task:    gh[a] = sum_{b in row(a)} gy[b] * g[pair(a,b)] * f_c           (the transpose of the forward row sum)
//         G task (channel tile t): gx[:, 32 t : 32 t + 32] += gh W_in -- waits for the row sums only (LDS counter)
// ==========================================================================================================
struct MolBwdLayerDev {
  const float *w1, *b1, *w2;               // filter network (raw)
  const float* w2_img;                     // split-precision LDS image of w2 (wpack), or null
  const float *in2f_t, *o1_t, *o2_t;       // packed input-gradient images of in2f / f2out.0 / f2out.1
};

struct MolBwdArgs {
  MolBwdLayerDev L[ML_MAXL];
  int n_layers;
  const float* gx_out;      // [N, 128]
  float* gx0;               // [N, 128] or null
  float* gr;                // [E, 3], assigned (may be null when gR is asked for)
  const float* rij;         // [E, 3], or null: from R / offsets
  const float* R;
  const float* offsets;
  float* gR;                // [N, 3] or null: dL/dR, every entry written (the pairwise transpose folded in)
  MolHeadDev head;          // head.w1t != null: dL/dx_L comes from the energy head (+ gx_out when that is given)
  const int64_t* idx_i;
  const int64_t* idx_j;
  const int32_t *half, *rev, *rowptr, *edge_pair, *grp_atom0, *grp_pair0;
  int n_groups;
  const float* saved;
  const float* gbase;
  int64_t gsz, N;
  RadialDev rb;
  int compact;              // as in the forward
  long long* dbg;
};

template <int KPB, bool SP>
__global__ __launch_bounds__(512) void k_schnet_mol_bwd(MolBwdArgs a) {
  constexpr int NF = 128, NT = 4, KB2 = 16;
  constexpr int W1F = SP ? 2 * MlW1Image<KPB>::BYTES / 4 : NF * KPB * 8;     // floats of the W1 image(s)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW2 = smem;                                  // NF*NF  (SP: high image | low image, 32 KB each)
  float* sW1 = sW2 + NF * NF;                         // NF*KPB*8  (SP: high image | low image)
  float* sb1 = sW1 + W1F;                             // NF (+ NF unused: same footprint as the forward)
  h16x8* const sW2h = (h16x8*)sW2;
  h16x8* const sW2l = sW2h + 2048;
  char* const sW1h = (char*)sW1;
  char* const sW1l = sW1h + MlW1Image<KPB>::BYTES;
  float* sGx = sb1 + 2 * NF;                          // [32][ML_LD] dL/dx of the current level
  float* sH = sGx + 32 * ML_LD;                       // h_l (saved by the forward)
  float* sGy = sH + 32 * ML_LD;                       // dL/dy_l
  float* sGh = sGy + 32 * ML_LD;                      // dL/dh_l; before that the hidden gradient of f2out
  MolPair* sP = (MolPair*)(sGh + 32 * ML_LD);         // per pair: local atoms, d, f_c, f_c'
  int2* sEb = (int2*)(sP + ML_MAXPAIRS);              // per directed edge: ((local pair << 8) | local neighbour, f_c)
  int* sRow = (int*)(sEb + ML_MAXEDGES);              // [33]
  int* sCnt = sRow + 36;                              // [4]
  float* sRb = (float*)(sCnt + 4);                    // [2][32] radial basis parameters
  float* sS = sRb + 64;                               // [ML_MAXPAIRS][2] per-pair geometry sums, all interactions
  short* sMap = (short*)(sS + 2 * ML_MAXPAIRS);       // [ML_MAXPAIRS] position in the pair list -> record (-1: beyond the cutoff)
  int* sScan = (int*)(sMap + ML_MAXPAIRS);            // [8]

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wv: SGPR
  const int hi = lane >> 5, el = lane & 31;
  if (tid < 64) {
    const int k = tid & 31;
    const float* src = (tid < 32) ? a.rb.p0 : a.rb.p1;
    sRb[tid] = (src && k < a.rb.n_rbf) ? src[k] : 1.0f;
  }

  if (a.dbg && tid == 0) { a.dbg[128 + 4 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime(); a.dbg[130 + 4 * blockIdx.x] = (long long)__builtin_readcyclecounter(); }
  for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
    const int a0 = a.grp_atom0[grp], na = a.grp_atom0[grp + 1] - a0;
    const int p0 = a.grp_pair0[grp], np_list = a.grp_pair0[grp + 1] - p0;
    const int e0 = a.rowptr[a0], ne = a.rowptr[a0 + na] - e0;
    const int Ltop = a.n_layers - 1;
    __syncthreads();

    // ---- group set-up.  Everything here is short and latency-bound, so the independent pieces share their round trips and
    //      barriers: the head's hidden gradient is filled while the pair records are loaded (the first barrier inside
    //      ml_pair_records() publishes both), the head GEMM (waves 0-3) runs beside the per-edge records (waves 4-7), and the
    //      filter weights of the top interaction are staged by the idle half of the two Dense phases below -- the derivative tasks
    //      of phase E are their first reader.
    const bool with_head = a.head.w1t != nullptr;
    for (int s = tid; s < 32 * 32; s += 512) {
      const int row = s >> 5, c4 = s & 31;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < na && a.gx_out) v = ml_ld<f32x4>(a.gx_out + (size_t)a0 * NF, (unsigned)(s * 16));
      *(f32x4*)(sGx + row * ML_LD + 4 * c4) = v;
      *(f32x4*)(sH + row * ML_LD + 4 * c4) = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!with_head) *(f32x4*)(sGh + row * ML_LD + 4 * c4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    ML_STAMP(120);
    if (with_head) {
      // ---- dL/dx_L through the energy head, part 1: the hidden gradient gE[mol] w2 . act'(pre_h) -> sGh[:, :H]
      //      (the columns beyond H are not read by part 2, and the first Dense phase rewrites sGh in full)
      const MolHeadDev& Hd = a.head;
      for (int s = tid; s < 32 * Hd.H; s += 512) {
        const int row = s / Hd.H, k = s - row * Hd.H;
        float v = 0.f;
        if (row < na) {
          const float pre = Hd.pre_h[(size_t)(a0 + row) * Hd.H + k];
          const float sg = spk_sigmoid(pre);
          const float da = Hd.act == SPK_ACT_SILU ? sg * (1.0f + pre * (1.0f - sg)) : sg;
          v = (Hd.gE ? Hd.gE[Hd.idx_m[a0 + row]] : 1.0f) * Hd.w2[k] * da;
        }
        sGh[row * ML_LD + k] = v;
      }
    }
    float pr3[3];             // this thread's pair vector: used again at the very end (dL/dr, dL/dR) without going back to memory
    const int np = ml_pair_records(sP, sMap, sScan, a.half, a.rij, a.R, a.offsets, a.idx_i, a.idx_j, p0, np_list, a0, a.rb.cutoff, a.compact != 0, tid, pr3);
    const int ntile = (np + 31) / 32;
    ML_STAMP(123);
    if (wv < NT) {
      if (with_head) {
        // ---- part 2: gx += hidden gradient x W1
        const MolHeadDev& Hd = a.head;
        const int KH = Hd.H / 8;
        const int t = wv;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* brow = sGh + el * ML_LD + 4 * hi;
        for (int c = 0; c < KH; c += 8) {            // H is a multiple of 32 (KH of 4): chunks of eight k-blocks, masked (A = rows of W1^T)
          f32x4 a8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            a8[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c + u < KH) a8[u] = ml_ld<f32x4>(Hd.w1t + (size_t)(32 * t) * Hd.H, (unsigned)((el * Hd.H + 8 * (c + u) + 4 * hi) * 4));
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (c + u < KH) {
              const f32x4 bv = *(const f32x4*)(brow + 8 * (c + u));
              acc = ML_MFMA(a8[u].x, bv.x, acc);
              acc = ML_MFMA(a8[u].y, bv.y, acc);
              acc = ML_MFMA(a8[u].z, bv.z, acc);
              acc = ML_MFMA(a8[u].w, bv.w, acc);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float* xp = sGx + el * ML_LD + 32 * t + 8 * q + 4 * hi;
          f32x4 xv = *(const f32x4*)xp;
          xv.x += acc[4 * q]; xv.y += acc[4 * q + 1]; xv.z += acc[4 * q + 2]; xv.w += acc[4 * q + 3];
          if (el >= na) xv = f32x4{0.f, 0.f, 0.f, 0.f};
          *(f32x4*)xp = xv;
        }
      }
    } else {
      // per directed edge: (row of the saved filter tensor << 8 | local neighbour, f_c); edges of dropped pairs point at the
      // first record's row with weight 0 (their own row was never written)
      const int t2 = tid - 256;
      const int row0 = np > 0 ? (sP[0].ij >> 16) : 0;
      for (int s = t2; s < ne; s += 256) {
        const int pos = ml_ld<int>(a.edge_pair + e0, (unsigned)s * 4u) - p0;
        const int rec = sMap[pos];
        const int nb = (int)(ml_ld<long long>(a.idx_j + e0, (unsigned)s * 8u) - a0);
        sEb[s] = rec >= 0 ? make_int2((pos << 8) | nb, __float_as_int(sP[rec].fc)) : make_int2((row0 << 8) | nb | (1 << 24), 0);
      }
      for (int s = t2; s < 2 * np; s += 256) sS[s] = 0.f;
      if (t2 <= na) sRow[t2] = a.rowptr[a0 + t2] - e0;
    }
    __syncthreads();
    ML_STAMP(32);

    for (int l = Ltop; l >= 0; --l) {
      const MolBwdLayerDev& P = a.L[l];
      const float* h_g = a.saved + (int64_t)l * a.N * (2 * NF);
      const float* pre3_g = h_g + a.N * (int64_t)NF;
      const float* g_g = a.gbase + (int64_t)l * a.gsz + (int64_t)p0 * NF;
      const bool last = (l == 0) && !a.gx0;     // nothing below consumes dL/dh_0: no row sums, no in2f transpose

      // ================= D1: gt = (gx W4) * ssp'(pre3);  the other half stages W2 of this interaction's filter network
      if (wv < NT) {
        const int t = wv;
        f32x4 avA[8], avB[8];
        ml_dense_load8(avA, P.o2_t, t, lane, 0);
        ml_dense_load8(avB, P.o2_t, t, lane, 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        f32x4 pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          pv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (el < na) pv[q] = ml_ld<f32x4>(pre3_g + (size_t)a0 * NF + 32 * t, (unsigned)((el * NF + 8 * q + 4 * hi) * 4));
        }
        acc = ml_dense_mma8<SP>(avA, sGx, lane, 0, acc);
        acc = ml_dense_mma8<SP>(avB, sGx, lane, 1, acc);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4*)(sGh + el * ML_LD + 32 * t + 8 * q + 4 * hi) =
              f32x4{acc[4 * q] * spk_sigmoid(pv[q].x), acc[4 * q + 1] * spk_sigmoid(pv[q].y), acc[4 * q + 2] * spk_sigmoid(pv[q].z), acc[4 * q + 3] * spk_sigmoid(pv[q].w)};
      } else {
        // (sW2 was last read by the derivative tasks of the interaction above: two barriers ago)
        if constexpr (SP) { if (P.w2_img) ml_stage_w2_image<256>(sW2, P.w2_img, tid - 256); else ml_stage_w2_split<256>(sW2h, sW2l, P.w2, tid - 256); }
        else ml_stage_packed<256, NF * NF / 4>(sW2, P.w2, NF, KB2, tid - 256);
      }
      if (tid == 0) { sCnt[0] = 0; sCnt[1] = 0; }
      __syncthreads();
      ML_STAMP(33 + 6 * (Ltop - l));

      // ================= D2: gy = gt W3;  the other half loads h_l and stages W1, b1
      if (wv < NT) {
        const int t = wv;
        f32x4 avA[8], avB[8];
        ml_dense_load8(avA, P.o1_t, t, lane, 0);
        ml_dense_load8(avB, P.o1_t, t, lane, 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = ml_dense_mma8<SP>(avA, sGh, lane, 0, acc);
        acc = ml_dense_mma8<SP>(avB, sGh, lane, 1, acc);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4*)(sGy + el * ML_LD + 32 * t + 8 * q + 4 * hi) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
      } else {
        const int t2 = tid - 256;
        for (int s = t2; s < na * 32; s += 256) {
          const int row = s >> 5, c4 = s & 31;
          *(f32x4*)(sH + row * ML_LD + 4 * c4) = ml_ld<f32x4>(h_g + (size_t)a0 * NF, (unsigned)(s * 16));
        }
        if constexpr (SP) ml_stage_w1_split<KPB>(sW1h, sW1l, P.w1, a.rb.n_rbf, t2);
        else ml_stage_packed<256, NF * KPB * 2>(sW1, P.w1, a.rb.n_rbf, KPB, t2);
        if (t2 < NF) sb1[t2] = P.b1[t2];
      }
      __syncthreads();
      ML_STAMP(34 + 6 * (Ltop - l));

      // ================= E: derivative tasks (pair tile, pair of channel tiles) + row-sum tasks (one atom, all channels)
      // The queue hands out the derivative tasks first, then the row sums, then -- unless nothing below consumes dL/dx -- the four
      // channel tiles of G (gx += gh W_in): G only waits for the row sums (a counter in LDS), not for the derivative tasks, so it
      // runs on the waves that the second round of derivative tasks leaves idle instead of being a phase of its own.
      const int nder = 2 * ntile;
      const int nrow = (last || np == 0) ? 0 : na;
      const int ng = last ? 0 : NT;
      if (np == 0 && !last) {    // no pair inside the cutoff: dL/dh = 0 (the buffer still holds the hidden gradient of f2out)
        for (int s = tid; s < 32 * 32; s += 512) *(f32x4*)(sGh + (s >> 5) * ML_LD + 4 * (s & 31)) = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();         // (uniform over the workgroup)
      }
      while (true) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&sCnt[0], 1);
        k = __builtin_amdgcn_readfirstlane(k);
        if (k >= nder + nrow + ng) break;
        if (k >= nder + nrow) {
          // ---- G, channel tile t: gx[:, 32 t : 32 t + 32] += gh W_in (weights requested before the wait for the row sums)
          const int t = k - nder - nrow;
          f32x4 avA[8], avB[8];
          ml_dense_load8(avA, P.in2f_t, t, lane, 0);
          ml_dense_load8(avB, P.in2f_t, t, lane, 1);
          if (nrow > 0) {
            while (__hip_atomic_load(&sCnt[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nrow) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          }
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          acc = ml_dense_mma8<SP>(avA, sGh, lane, 0, acc);
          acc = ml_dense_mma8<SP>(avB, sGh, lane, 1, acc);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float* xp = sGx + el * ML_LD + 32 * t + 8 * q + 4 * hi;
            f32x4 xv = *(const f32x4*)xp;
            xv.x += acc[4 * q]; xv.y += acc[4 * q + 1]; xv.z += acc[4 * q + 2]; xv.w += acc[4 * q + 3];
            if (el >= na) xv = f32x4{0.f, 0.f, 0.f, 0.f};
            *(f32x4*)xp = xv;
            if (l == 0 && el < na) ml_st<f32x4>(a.gx0 + (size_t)a0 * NF + 32 * t, (unsigned)((el * NF + 8 * q + 4 * hi) * 4), xv);
          }
          continue;
        }
        if (k >= nder) {
          // ---- gh[a][c] = sum over the row of a of gy[b][c] g[pair][c] f_c
          ml_row_sums4<10>(sGh, sGy, g_g, sEb, sRow, k - nder, lane);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) atomicAdd(&sCnt[1], 1);
          continue;
        }
        const int tile = k >> 1, tp = k & 1;
        const int pfirst = 32 * tile;
        const int nvalid = (np - pfirst) < 32 ? (np - pfirst) : 32;
        const bool valid = el < nvalid;
        const int pl = pfirst + (valid ? el : (nvalid - 1));
        const MolPair pr = sP[pl];
        const float fc = valid ? pr.fc : 0.f, dfc = valid ? pr.dfc : 0.f;
        const int pi = pr.ij & 255, pj = (pr.ij >> 8) & 255;
        const int grow = pr.ij >> 16;                       // row of this pair in the saved filter tensor
        // the saved raw filter outputs of this lane's pair for both channel tiles of the task: requested first, used last
        f32x4 gl[2][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) gl[tt][q] = ml_ld<f32x4>(g_g + 64 * tp, (unsigned)((grow * NF + 4 * hi + 32 * tt + 8 * q) * 4));
        float s1 = 0.f, s2 = 0.f;
        if constexpr (SP) {
          // ---- split form (spk_split.h).  GEMM 1, value and derivative, for the four hidden tiles: A = the W1 images, B = this pair's
          // basis values / slopes in the lane's k-slots; z' = sigmoid(a) a' goes from the accumulator registers into the B operand
          // of GEMM 2' as it lies (W2 is staged with its contraction index in accumulator order).
          // (lane re-derived through an opaque asm: the LDS offsets below are then formed here and not hoisted out of the task loop
          //  into the prologue of the kernel, where the allocator parks them in scratch -- section 4.3a of DESIGN.md)
          int lane_o = lane;
          asm volatile("" : "+v"(lane_o));
          const int hi_o = lane_o >> 5;
          h16x8 ph[2], pl[2], dh[2], dl[2];
          ml_basis_split<KPB, true>(a.rb.kind, a.rb.n_rbf, sRb, sRb + 32, hi_o, pr.d, ph, pl, dh, dl);
          h16x8 zph[NT][2], zpl[NT][2];
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            f32x16 zc, zcx, zq, zqx;
#pragma unroll
            for (int r = 0; r < 16; ++r) { zc[r] = sb1[32 * c + ml_row(r, hi_o)]; zcx[r] = 0.f; zq[r] = 0.f; zqx[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < (KPB > 2 ? 2 : 1); ++s) {
              h16x8 wh, wl;
              ml_w1_operand<KPB>(sW1h, sW1l, s, c * 64 + lane_o, wh, wl);
              SP_STEP(wh, wl, ph[s], pl[s], zc, zcx);
              SP_STEP(wh, wl, dh[s], dl[s], zq, zqx);
            }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int r = 8 * sp + e;
                float spv, sg;
                spk_fast_softplus_sigmoid(fmaf(zcx[r], SP_DOWN, zc[r]), spv, sg);
                v[e] = fmaf(zqx[r], SP_DOWN, zq[r]) * sg;
              }
              sp_split8(v, zph[c][sp], zpl[c][sp]);
            }
          }
          // ---- GEMM 2' per channel tile (rows = channels 32 t + ml_row(r, hi), columns = pairs): g' = W2 z'
          const float* gyi_p = sGy + pi * ML_LD + 4 * hi_o;
          const float* gyj_p = sGy + pj * ML_LD + 4 * hi_o;
          const float* hi_p = sH + pi * ML_LD + 4 * hi_o;
          const float* hj_p = sH + pj * ML_LD + 4 * hi_o;
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * tp + tt;
            f32x16 gp, gpx;
#pragma unroll
            for (int r = 0; r < 16; ++r) { gp[r] = 0.f; gpx[r] = 0.f; }
            const h16x8* wbh = sW2h + (t * 8) * 64 + lane_o;
            const h16x8* wbl = sW2l + (t * 8) * 64 + lane_o;
            h16x8 wh[8], wl[8];
#pragma unroll
            for (int s = 0; s < 2; ++s) { wh[s] = wbh[s * 64]; wl[s] = wbl[s * 64]; }
            ML_PIN();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              if (s + 2 < 8) { wh[s + 2] = wbh[(s + 2) * 64]; wl[s + 2] = wbl[(s + 2) * 64]; ML_PIN(); }
              SP_STEP(wh[s], wl[s], zph[s >> 1][s & 1], zpl[s >> 1][s & 1], gp, gpx);
            }
            SP_FOLD(gp, gpx);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int col = 32 * t + 8 * q;
              const f32x4 gyi = *(const f32x4*)(gyi_p + col), gyj = *(const f32x4*)(gyj_p + col);
              const f32x4 hvi = *(const f32x4*)(hi_p + col), hvj = *(const f32x4*)(hj_p + col);
              const f32x4 gq = gl[tt][q];
              const float D0 = gp[4 * q] * fc + gq.x * dfc, D1 = gp[4 * q + 1] * fc + gq.y * dfc;
              const float D2 = gp[4 * q + 2] * fc + gq.z * dfc, D3 = gp[4 * q + 3] * fc + gq.w * dfc;
              s1 += gyi.x * hvj.x * D0 + gyi.y * hvj.y * D1 + gyi.z * hvj.z * D2 + gyi.w * hvj.w * D3;
              s2 += gyj.x * hvi.x * D0 + gyj.y * hvi.y * D1 + gyj.z * hvi.z * D2 + gyj.w * hvi.w * D3;
            }
          }
        } else {
        float phi[KPB][4], dphi[KPB][4];
#pragma unroll
        for (int u = 0; u < KPB; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) ml_rbf(a.rb.kind, a.rb.n_rbf, sRb, sRb + 32, 8 * u + 4 * hi + v, pr.d, phi[u][v], dphi[u][v]);
        // ---- GEMM 1, value and derivative (rows = hidden channels, columns = pairs): z' = sigmoid(W1 phi + b1) * (W1 phi')
        // Software-pipelined by hand: a wave issues in order, so an activation block placed between two MFMA groups leaves the
        // matrix pipe idle for its whole length.  The MFMAs of hidden tile c + 1 are therefore interleaved with the activations
        // of tile c (one or two of its 16 accumulator rows per MFMA pair): the VALU work runs in the shadow of the MFMAs.
        f32x16 zp[NT];
        {
          constexpr int NSTEP = 4 * KPB;
          f32x4 wq = *(const f32x4*)(sW1 + lane * 4);
          f32x16 zc, zq;
#pragma unroll
          for (int r = 0; r < 16; ++r) { zc[r] = sb1[ml_row(r, hi)]; zq[r] = 0.f; }
#pragma unroll
          for (int u = 0; u < KPB; ++u) {
            f32x4 wn = wq;
            if (u + 1 < NT * KPB) wn = *(const f32x4*)(sW1 + ((u + 1) * 64 + lane) * 4);
            zc = ML_MFMA(wq.x, phi[u][0], zc); zq = ML_MFMA(wq.x, dphi[u][0], zq);
            zc = ML_MFMA(wq.y, phi[u][1], zc); zq = ML_MFMA(wq.y, dphi[u][1], zq);
            zc = ML_MFMA(wq.z, phi[u][2], zc); zq = ML_MFMA(wq.z, dphi[u][2], zq);
            zc = ML_MFMA(wq.w, phi[u][3], zc); zq = ML_MFMA(wq.w, dphi[u][3], zq);
            wq = wn;
          }
#pragma unroll
          for (int c = 0; c < NT; ++c) {
            f32x16 zcn = zc, zqn = zq;
            if (c + 1 < NT) {
#pragma unroll
              for (int r = 0; r < 16; ++r) { zcn[r] = sb1[32 * (c + 1) + ml_row(r, hi)]; zqn[r] = 0.f; }
#pragma unroll
              for (int u = 0; u < KPB; ++u) {
                const int nxt = (c + 1) * KPB + u + 1;
                f32x4 wn = wq;
                if (nxt < NT * KPB) wn = *(const f32x4*)(sW1 + (nxt * 64 + lane) * 4);
                const float wv4[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                  zcn = ML_MFMA(wv4[v], phi[u][v], zcn);
                  zqn = ML_MFMA(wv4[v], dphi[u][v], zqn);
                  const int st = 4 * u + v;
#pragma unroll
                  for (int r = (16 * st) / NSTEP; r < (16 * (st + 1)) / NSTEP; ++r) {
                    float sp, sg;
                    spk_fast_softplus_sigmoid(zc[r], sp, sg);
                    zq[r] *= sg;
                  }
                }
                wq = wn;
              }
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                float sp, sg;
                spk_fast_softplus_sigmoid(zc[r], sp, sg);
                zq[r] *= sg;
              }
            }
            zp[c] = zq;
            zc = zcn; zq = zqn;
          }
        }
        // ---- GEMM 2' per channel tile (rows = channels 32 t + ml_row(r, hi), columns = pairs): g' = W2 z'
        const float* gyi_p = sGy + pi * ML_LD + 4 * hi;
        const float* gyj_p = sGy + pj * ML_LD + 4 * hi;
        const float* hi_p = sH + pi * ML_LD + 4 * hi;
        const float* hj_p = sH + pj * ML_LD + 4 * hi;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * tp + tt;
          f32x16 gp;
#pragma unroll
          for (int r = 0; r < 16; ++r) gp[r] = 0.f;
          {
            // weights from LDS, requested three k-blocks ahead and pinned there (the one-ahead form was collapsed by the compiler
            // to "ds_read, wait, four MFMAs": a derivative task alone on its SIMD then waits out the LDS round trip of every k-block)
            const float* wbase = sW2 + ((int64_t)t * KB2 * 64 + lane) * 4;
            f32x4 wb[KB2];
#pragma unroll
            for (int ug = 0; ug < 3; ++ug) wb[ug] = *(const f32x4*)(wbase + ug * 256);
            ML_PIN();
#pragma unroll
            for (int ug = 0; ug < KB2; ++ug) {
              const int c = ug >> 2, q = ug & 3;
              if (ug + 3 < KB2) { wb[ug + 3] = *(const f32x4*)(wbase + (ug + 3) * 256); ML_PIN(); }
              gp = ML_MFMA(wb[ug].x, zp[c][4 * q + 0], gp);
              gp = ML_MFMA(wb[ug].y, zp[c][4 * q + 1], gp);
              gp = ML_MFMA(wb[ug].z, zp[c][4 * q + 2], gp);
              gp = ML_MFMA(wb[ug].w, zp[c][4 * q + 3], gp);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 32 * t + 8 * q;
            const f32x4 gyi = *(const f32x4*)(gyi_p + col), gyj = *(const f32x4*)(gyj_p + col);
            const f32x4 hvi = *(const f32x4*)(hi_p + col), hvj = *(const f32x4*)(hj_p + col);
            const f32x4 gq = gl[tt][q];
            const float D0 = gp[4 * q] * fc + gq.x * dfc, D1 = gp[4 * q + 1] * fc + gq.y * dfc;
            const float D2 = gp[4 * q + 2] * fc + gq.z * dfc, D3 = gp[4 * q + 3] * fc + gq.w * dfc;
            s1 += gyi.x * hvj.x * D0 + gyi.y * hvj.y * D1 + gyi.z * hvj.z * D2 + gyi.w * hvj.w * D3;
            s2 += gyj.x * hvi.x * D0 + gyj.y * hvi.y * D1 + gyj.z * hvi.z * D2 + gyj.w * hvi.w * D3;
          }
        }
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (hi == 0 && valid) { atomicAdd(&sS[2 * pl], s1); atomicAdd(&sS[2 * pl + 1], s2); }
      }
      ML_STAMP(35 + 6 * (Ltop - l));
      __syncthreads();
      ML_STAMP(36 + 6 * (Ltop - l));
      if (last) break;

    }

    // ---- dL/dr of both directions of every pair, once for all interactions (pairs beyond the cutoff: zero); with gR the pair's
    //      contribution (s1 + s2) r / d is parked in LDS for the per-atom pass below (sGh is free by now)
    float* sV = sGh;                                    // [ML_MAXPAIRS][3]
    if (a.gR) __syncthreads();
    if (tid < np_list) {                                // one pair per thread (np_list <= ML_MAXPAIRS < 512), its vector still in registers
      const int s = tid;
      const int rec = sMap[s];
      float s1 = 0.f, s2 = 0.f;
      if (rec >= 0) {
        const float d = sP[rec].d;
        const float inv = d > 0.f ? 1.0f / d : 0.f;
        s1 = sS[2 * rec] * inv; s2 = sS[2 * rec + 1] * inv;
      }
      const float rx = pr3[0], ry = pr3[1], rz = pr3[2];
      if (a.gr) {
        const int64_t e = a.half[p0 + s];
        const int64_t e2 = a.rev[e];
        a.gr[3 * e] = s1 * rx; a.gr[3 * e + 1] = s1 * ry; a.gr[3 * e + 2] = s1 * rz;
        a.gr[3 * e2] = -s2 * rx; a.gr[3 * e2 + 1] = -s2 * ry; a.gr[3 * e2 + 2] = -s2 * rz;
      }
      if (a.gR && rec >= 0) {
        const float w = s1 + s2;
        sV[3 * rec] = w * rx; sV[3 * rec + 1] = w * ry; sV[3 * rec + 2] = w * rz;
      }
    }
    // ---- dL/dR: the transpose of r_ij = R_j - R_i + offsets, per atom over its row (each pair of the atom appears once
    //      there): the pair (i, j) with canonical vector r gives -(s1 + s2) r / d to i and +(s1 + s2) r / d to j.  Fixed order.
    if (a.gR) {
      __syncthreads();
      // eight lanes per (atom, component): they walk the row's edges with stride 8 and meet by shuffles (fixed order)
      const int slot = tid & 7, q = tid >> 3;          // q < 64 >= 3 * na / ... : na <= 21 atoms fill 63 of the 64 groups; larger groups loop
      for (int s = q; s < 3 * na; s += 64) {
        const int at = s / 3, comp = s - 3 * at;
        float acc = 0.f;
        for (int e = sRow[at] + slot; e < sRow[at + 1]; e += 8) {
          const int x = sEb[e].x;
          if (x & (1 << 24)) continue;
          const int rec = sMap[(x >> 8) & 0xFFFF];
          const float v = sV[3 * rec + comp];
          acc += ((sP[rec].ij & 255) == at) ? -v : v;
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (slot == 0) a.gR[3 * (size_t)(a0 + at) + comp] = a.head.negate ? -acc : acc;
      }
    }
    ML_STAMP(63);
  }
  if (a.dbg && tid == 0) { a.dbg[129 + 4 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime(); a.dbg[131 + 4 * blockIdx.x] = (long long)__builtin_readcyclecounter(); }
}

static size_t mol_bwd_lds(int kpb, bool sp) {
  return (size_t)(128 * 128 + mol_w1_floats(kpb, sp) + 2 * 128 + 4 * 32 * ML_LD + 64 + 2 * ML_MAXPAIRS) * sizeof(float) + ML_MAXPAIRS * sizeof(MolPair) +
         (2 * ML_MAXEDGES + 36 + 4 + 8) * sizeof(int) + ML_MAXPAIRS * sizeof(short);
}

template <int KPB, bool SP>
static int launch_mol_bwd_sp(const MolBwdArgs& a, hipStream_t stream);
template <int KPB>
static int launch_mol_bwd(const MolBwdArgs& a, hipStream_t stream) {
  return spk_get_split() ? launch_mol_bwd_sp<KPB, true>(a, stream) : launch_mol_bwd_sp<KPB, false>(a, stream);
}
template <int KPB, bool SP>
static int launch_mol_bwd_sp(const MolBwdArgs& a, hipStream_t stream) {
  const size_t lds = mol_bwd_lds(KPB, SP);
  auto kern = k_schnet_mol_bwd<KPB, SP>;
  static SpkPerDevice attr_set;
  int attr_dev;
  if (attr_set.pending(&attr_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_dev);
  }
  int grid = a.n_groups;
  const int maxg = spk_num_cus();
  if (grid > maxg) grid = maxg;
  SpkProfScope prof("schnet_mol_bwd", stream);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// n_rbf up to 24 (3 k-blocks): the LDS budget of the backward (per-pair sums) ends there; wider bases take the general driver
bool spk_schnet_mol_bwd_eligible(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb) {
  return spk_schnet_mol_eligible(m, g, rb) && (rb->n_rbf + 7) / 8 <= 3 && !getenv("SPK_NO_MOL_BWD");
}

int spk_schnet_mol_backward_ex(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                               const float* gx_out, const float* r_ij, const float* R, const float* offsets, const MolHeadDev* head,
                               const float* saved, int64_t gsz, float* gr, float* gR, float* gx0, hipStream_t stream);
int spk_schnet_mol_backward(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                            const float* gx_out, const float* r_ij, const float* saved, int64_t gsz, float* gr, float* gx0,
                            hipStream_t stream) {
  return spk_schnet_mol_backward_ex(m, g, rb, ptab, gx_out, r_ij, nullptr, nullptr, nullptr, saved, gsz, gr, nullptr, gx0, stream);
}
int spk_schnet_mol_backward_ex(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                               const float* gx_out, const float* r_ij, const float* R, const float* offsets, const MolHeadDev* head,
                               const float* saved, int64_t gsz, float* gr, float* gR, float* gx0, hipStream_t stream) {
  MolBwdArgs a;
  a.R = R; a.offsets = offsets; a.gR = gR;
  if (head) a.head = *head; else { a.head.w1 = nullptr; a.head.w1t = nullptr; a.head.negate = 0; }
  const bool split = spk_get_split() != 0;
  a.n_layers = m->n_interactions;
  for (int l = 0; l < m->n_interactions; ++l) {
    const spk_schnet_layer_t& P = m->layers[l];
    MolBwdLayerDev& D = a.L[l];
    D.in2f_t = split ? spk_packed_split_of(ptab, P.in2f_w, 1) : spk_packed_of(ptab, P.in2f_w, 1);
    D.o1_t = split ? spk_packed_split_of(ptab, P.f2out_w1, 1) : spk_packed_of(ptab, P.f2out_w1, 1);
    D.o2_t = split ? spk_packed_split_of(ptab, P.f2out_w2, 1) : spk_packed_of(ptab, P.f2out_w2, 1);
    SPK_CHECK_ARG(D.in2f_t && D.o1_t && D.o2_t, "spk_schnet_mol_backward: packed weight images missing");
    D.w1 = P.fn_w1; D.b1 = P.fn_b1; D.w2 = P.fn_w2;
    D.w2_img = split ? ptab.extra_of(P.fn_w2) : nullptr;
  }
  a.gx_out = gx_out; a.gx0 = gx0; a.gr = gr; a.rij = r_ij; a.idx_i = g->idx_i; a.idx_j = g->idx_j;
  a.half = g->half; a.rev = g->rev; a.rowptr = g->rowptr; a.edge_pair = g->edge_pair; a.grp_atom0 = g->grp_atom0; a.grp_pair0 = g->grp_pair0;
  a.n_groups = g->n_groups; a.saved = saved; a.N = g->n_atoms; a.gsz = gsz;
  a.gbase = saved + (int64_t)m->n_interactions * g->n_atoms * (m->n_filters + m->n_atom_basis);
  a.rb = spk_radial_dev(rb);
  a.compact = 1;
  a.dbg = g_mol_dbg;
  switch ((rb->n_rbf + 7) / 8) {
    case 1: return launch_mol_bwd<1>(a, stream);
    case 2: return launch_mol_bwd<2>(a, stream);
    case 3: return launch_mol_bwd<3>(a, stream);
  }
  spk_set_error("spk_schnet_mol_backward: n_rbf = %d unsupported", rb->n_rbf);
  return SPK_ERR_ARG;
}
