/*
 * spk_hip.h -- C ABI of the MI355X-native (gfx950) message-passing core for SchNetPack.
 *
 * This is the drop-in boundary of the hot path: plain pointers and
 * sizes, no torch types.  All pointers are DEVICE pointers unless a parameter name starts with
 * `host_`.  All floating point data is fp32, row-major contiguous; index arrays are int64 exactly
 * as the reference's batch dict delivers them (`_idx_i`, `_idx_j`, `_idx_m`,
 * src/schnetpack/properties.py:23-32).  `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Every entry point returns 0 on success or a negative SPK_ERR_* code;
 * spk_last_error() returns a human readable message for the calling thread.  Nothing here
 * falls back to a CPU implementation: without a GPU every compute entry point fails.
 *
 * Each entry point cites the reference interface it replaces (paths into
 * /root/reference/src/schnetpack).
 */
#ifndef SPK_HIP_H
#define SPK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPK_OK 0
#define SPK_ERR_ARG (-1)     /* bad shape / unsupported size / null pointer */
#define SPK_ERR_HIP (-2)     /* HIP runtime error (launch failure, no device) */
#define SPK_ERR_INDEX (-3)   /* index out of range (reported by spk_edge_plan) */

/* activation ids (nn/activations.py:9-22 shifted_softplus, torch.nn.functional.silu) */
#define SPK_ACT_NONE 0
#define SPK_ACT_SSP 1
#define SPK_ACT_SILU 2

/* radial basis kinds (nn/radial.py:18-48 GaussianRBF, :82-110 BesselRBF) */
#define SPK_RBF_GAUSSIAN 0
#define SPK_RBF_BESSEL 1

/* kernel variant selection for the fused edge kernels */
#define SPK_VARIANT_AUTO 0    /* MFMA kernels when the shape is supported, else simple */
#define SPK_VARIANT_SIMPLE 1  /* straightforward HIP kernels (any shape), used as cross-check */
#define SPK_VARIANT_MFMA 2    /* force the MFMA kernels (error if shape unsupported) */
#define SPK_VARIANT_MFMA_DIRECTED 3 /* MFMA kernels, but one filter per DIRECTED edge even on symmetric lists */
#define SPK_VARIANT_MFMA_PAIR 4     /* pair kernels (the default on symmetric lists) */
#define SPK_VARIANT_MFMA_MOL 5      /* experiment: group-local pair kernel accumulating in LDS (block-diagonal lists) */

/* Radial basis x cosine cutoff description (nn/radial.py, nn/cutoff.py:14-57).
 * gaussian: p0 = offsets[n_rbf], p1 = widths[n_rbf];  bessel: p0 = freqs[n_rbf], p1 unused. */
typedef struct {
  int32_t kind;
  int32_t n_rbf;
  const float* p0;
  const float* p1;
  float cutoff;
} spk_radial_t;

/* EXPERIMENT, opt-in.  Block plan of a list sorted by idx_i for the block kernels of the PaiNN message (spk_painn_blk.hip; built by
 * spk_blocks_build into buffers of the caller, sizes from spk_blocks_sizes): atoms in groups of SPK_BLK_ATOMS consecutive
 * atoms (split into 2 / 4 / 8 sub-blocks where the unique neighbours of a group exceed `cap`), per sub-block the
 * ascending list of its unique neighbour atoms and per edge the position of idx_j in it, 16-edge tiles aligned to atoms;
 * plus the per-call workspace the kernels fill from r_ij (radial basis x cutoff in MFMA operand order, per-edge records,
 * per-slice partial sums of the geometry gradient). */
#define SPK_BLK_ATOMS 8
#define SPK_BLK_STG_MAX 9          /* 16-byte staging units per thread and 16-channel slice: two LDS buffers of 9 x 512 x 16 B = 144 KB */
#define SPK_BLK_CAP (SPK_BLK_STG_MAX * 64 * SPK_BLK_ATOMS / 24)   /* = 192 unique neighbours per block (6 row pieces of 64 B each) */
typedef struct {
  int32_t n_groups;          /* ceil(n_atoms / SPK_BLK_ATOMS) */
  int32_t max_unique;        /* largest unique-neighbour count of a sub-block */
  int32_t n_tiles;
  int32_t cap;               /* capacity the plan was built for */
  int32_t ks;                /* k-steps of 4 radial functions: 5 (n_rbf <= 20) or 8 (n_rbf <= 32) */
  int32_t ok;                /* 0: the list does not fit (a single atom has more than cap / 2048 neighbours): keep the row kernels */
  int32_t n_blocks;          /* sub-blocks of all groups, in atom order */
  int32_t reserved;
  const int32_t* blk_desc;   /* [n_blocks][4] first atom, atoms (0..SPK_BLK_ATOMS), first edge, unique neighbours */
  const int32_t* sub_n;      /* [n_groups] sub-blocks per group: 1, 2, 4 or 8 */
  const int32_t* sub_u;      /* [n_groups * SPK_BLK_ATOMS] unique neighbours of sub-block s of group g at [SPK_BLK_ATOMS g + s] */
  const int32_t* uniq;       /* [n_edges] unique neighbours of a sub-block from uniq[rowptr[first atom]] on */
  const uint16_t* jl;        /* [n_edges] position of idx_j[e] in its sub-block's list */
  const int32_t* atom_tile0; /* [n_atoms + 1] first tile of every atom */
  const int32_t* tile_info;  /* [n_tiles][2] first edge, number of edges (1..16) */
  float* apack;              /* [n_tiles][ks][64]  f_c phi_k           (workspace, written per call) */
  float* adpack;             /* [n_tiles][ks][64]  d(f_c phi_k)/dd */
  float* rec;                /* [n_tiles][6][16]   local neighbour, unit vector, f_c, f_c' */
  float* part;               /* [F / 16][n_edges][4] */
} spk_blocks_t;

/* The list sorted by NEIGHBOUR (spk_transposed_build): for lists that are sorted by idx_i but not symmetric (half lists, one-sided
 * lists of external back-ends; transform/neighborlist.py:446-456 and interfaces/lammps/pair_schnetpack.cpp:240-267 need not deliver
 * both directions) the "scatter over idx_j" of every backward runs as a ROW pass over this transposed list -- the forward kernel
 * with (idx_i, idx_j) exchanged -- instead of float atomics. */
typedef struct {
  const int64_t* idx_i;   /* [E] = idx_j[perm], ascending */
  const int64_t* idx_j;   /* [E] = idx_i[perm] */
  const int32_t* rowptr;  /* [N + 2] CSR of the above (row N collects out-of-range neighbours: empty on a valid list) */
  const int32_t* perm;    /* [E] pair k of the transposed list is pair perm[k] of the list (stable sort by idx_j) */
  float* r_perm;          /* [E, 3] workspace: r_ij[perm], refilled per call */
} spk_transposed_t;

/* Neighbour-list description: what `_idx_i`, `_idx_j` look like plus the CSR row pointers and
 * flags that spk_edge_plan() derives once per neighbour list. */
typedef struct {
  int64_t n_atoms;
  int64_t n_edges;
  const int64_t* idx_i;   /* [E] centre atom of every directed edge */
  const int64_t* idx_j;   /* [E] neighbour atom */
  const int32_t* rowptr;  /* [N+1] CSR offsets into the edge list; valid iff sorted != 0 */
  int32_t sorted;         /* idx_i ascending (every reference neighbour list; neighborlist.py:450-453) */
  int32_t symmetric;      /* for every edge (i<-j, r) the list also holds (j<-i, -r) */
  const int32_t* rev;     /* [E] index of the reversed edge (valid iff symmetric); may be NULL */
  const int32_t* half;    /* [n_half] the canonical edge of every undirected pair (e < rev[e]), ascending; may be NULL */
  int64_t n_half;         /* = E/2 on a symmetric list */
  /* optional block-diagonal structure (batches of molecules): consecutive atom ranges that no edge leaves.
   * [n_groups+1] prefix arrays: first atom, first entry in `half`, first 32-pair tile (tiles aligned to
   * groups).  n_groups == 0: unknown / not block diagonal. */
  const int32_t* grp_atom0;
  const int32_t* grp_pair0;
  const int32_t* grp_tile0;
  int32_t n_groups;
  int32_t max_group_atoms;
  int64_t n_tiles_grouped; /* = grp_tile0[n_groups] */
  /* Lists with pairs at or beyond the cutoff (MD skin lists, md/neighborlist_md.py:36-38): with filter_pairs != 0
   * the fused SchNet representation compacts the pair list per call (pairs with d < cutoff keep their order) into
   * its `saved` buffer and the cfconv kernels walk only those; the dropped pairs contribute exactly zero
   * (f_c = f_c' = 0).  The PaiNN message dispatch reads the flag as "this list has a skin" and keeps the row kernels,
   * which drop such pairs before their rows are fetched.  n_half_dev is set by the library (a device count that replaces n_half inside the kernels);
   * callers leave it NULL. */
  int32_t filter_pairs;
  int32_t reserved0;
  const int32_t* n_half_dev;
  /* optional ([n_edges] or NULL): position in `half` of the undirected pair every DIRECTED edge belongs to
   * (edge_pair[half[k]] = edge_pair[rev[half[k]]] = k).  With it and the block-diagonal structure above (groups of at
   * most 32 atoms) the fused SchNet representation runs molecule-resident: one workgroup per group, all interactions in
   * one launch, per-atom row sums instead of float atomics (spk_schnet_mol.hip). */
  const int32_t* edge_pair;
  int32_t max_group_pairs; /* largest number of undirected pairs inside one group (0: unknown) */
  int32_t reserved1;
  const spk_blocks_t* blocks; /* optional block plan (large lists; spk_blocks_build), NULL: none */
  const spk_transposed_t* transposed; /* optional: the list sorted by neighbour (asymmetric lists; spk_transposed_build), NULL: none */
} spk_graph_t;

/* ------------------------------------------------------------------ library / device info */
int spk_version(void);
const char* spk_last_error(void);
/* host_info[0]=compute units, [1]=wavefront size, [2]=LDS bytes per workgroup, [3]=gfx arch number */
int spk_device_info(int32_t* host_info);
void spk_set_variant(int variant);
int spk_get_variant(void);
/* Split-precision matrix path (csrc/spk_split.h): the filter-network / filter products of the fused kernels as three
 * v_mfma_f32_32x32x16_f16 products of (high, low) fp16 operand pairs with fp32 accumulation -- fp32-quality results at 3/16 of the
 * time of v_mfma_f32_32x32x2_f32.  1 (default; environment SPK_SPLIT=0 starts with 0) or 0 = the fp32 matrix instruction everywhere
 * (the A/B partner of profiles/r06_split_mfma.md; also the path for operands beyond the fp16 range, |x| >= 65504).  Replaces no
 * reference interface: the reference computes these products with F.linear (nn/base.py:52-55). */
void spk_set_split(int on);
int spk_get_split(void);
/* Per-kernel timing with HIP events recorded on the launch stream (measurement aid for bench.py;
 * off by default).  spk_profile_report() synchronises the device and returns lines
 * "<kernel-tag> <launch count> <total ms>". */
void spk_profile_enable(int on);
const char* spk_profile_report(void);

/* ------------------------------------------------------------------ neighbour-list plan
 * Derives what the fused kernels need from the delivered index arrays: CSR row pointers,
 * sortedness, index range check and (if r_ij != NULL) whether the list is symmetric.
 * host_flags[0]=sorted, [1]=in_range, [2]=symmetric.  `rev` ([E] int32, may be NULL) receives the
 * index of the reversed edge of every edge (-1 if none).  Synchronises the stream (one D2H copy of
 * 16 bytes) -- call once per neighbour list, not per force call.  `scratch` >= 16 bytes device. */
int spk_edge_plan(const int64_t* idx_i, const int64_t* idx_j, const float* r_ij, int64_t n_edges,
                  int64_t n_atoms, int32_t* rowptr, int32_t* rev, int32_t* scratch,
                  int32_t* host_flags, void* stream);

/* Block plan (see spk_blocks_t).  spk_blocks_sizes: element counts of the ten buffers for a list of this size
 * (sizes[0..8]: sub_n, sub_u, uniq, jl (uint16), atom_tile0, tile_info, apack = adpack, rec, part; sizes[9] = ks;
 * sizes[10]: blk_desc (int32)).
 * spk_blocks_build fills the index buffers of `out` (the caller has set the pointers), n_groups / max_unique / n_tiles /
 * n_blocks / ok; cap <= 0: SPK_BLK_CAP.  dev_stats: >= 16 bytes device scratch; host_stats [4]: largest
 * unique count, 1 if the list does not fit, number of tiles, number of blocks.  Synchronises the stream once -- per list, not per call.
 * spk_blocks_prepare_f32 fills the per-call tables from r_ij (the drivers do this themselves).
 * spk_painn_set_block: 1 = use the block kernels whenever a usable plan hangs on the graph; 0 (default) = never -- they are an
 * opt-in experiment: parity-green, measured slower than the row / tile kernels on the 32k-atom water box (spk_painn_blk.hip). */
int spk_blocks_sizes(int64_t n_atoms, int64_t n_edges, int32_t n_rbf, int32_t n_atom_basis, int64_t* sizes);
int spk_blocks_build(const spk_graph_t* g, int32_t n_rbf, int32_t cap, spk_blocks_t* out, int32_t* dev_stats,
                     int32_t* host_stats, void* stream);
int spk_blocks_prepare_f32(const spk_graph_t* g, const spk_radial_t* rb, const float* r_ij, void* stream);
void spk_painn_set_block(int mode);
int spk_blocks_group_atoms(void);   /* SPK_BLK_ATOMS of the built library */
void spk_painn_blk_set_debug_buffer(void* device_buffer, int block);   /* tuning aid: cycle stamps of one workgroup; NULL = off */

/* ------------------------------------------------------------------ nn/scatter.py:7-34
 * y[o, k, c] = sum_{e: idx[e]==k} x[o, e, c]   (x: [outer, E, inner], y: [outer, dim_size, inner]).
 * rowptr may be NULL (then atomics are used); with rowptr the reduction is a deterministic
 * segmented sum and y is written exactly once.  y is fully overwritten. */
int spk_scatter_add_f32(const float* x, const int64_t* idx, const int32_t* rowptr, int64_t outer,
                        int64_t n_edges, int64_t inner, int64_t dim_size, float* y, void* stream);
/* transpose of the above (its backward): y[o, e, c] = x[o, idx[e], c] */
int spk_gather_f32(const float* x, const int64_t* idx, int64_t outer, int64_t n_rows,
                   int64_t n_edges, int64_t inner, float* y, void* stream);

/* CSR row pointers rowptr [n_rows + 1] of an ASCENDING index (idx_i of a neighbour list, idx_m) computed
 * on the device without a host round trip, so that the call can be part of a captured HIP graph whose index
 * buffers are refilled between replays (static-shape training steps).  err (device int32, may be NULL) is
 * OR-ed with 1 if idx is not ascending and with 2 if an entry is outside [0, n_rows); the caller polls it. */
int spk_segment_rowptr_i32(const int64_t* idx, int64_t n, int64_t n_rows, int32_t* rowptr, int32_t* err,
                           void* stream);
/* err[0] |= 2 if an entry of idx lies outside [0, hi): device-only validation of the unsorted indices (idx_j, Z) of a
 * static-shape training step, launched next to the row-pointer refresh inside the captured graph */
int spk_index_range_check(const int64_t* idx, int64_t n, int64_t hi, int32_t* err, void* stream);

/* ------------------------------------------------------------------ atomistic/distances.py:14-26
 * r_ij[e] = (R[idx_j[e]] - R[idx_i[e]]) + offsets[e]   (offsets may be NULL). */
int spk_pairwise_f32(const float* R, const int64_t* idx_i, const int64_t* idx_j,
                     const float* offsets, int64_t n_edges, float* r_ij, void* stream);
/* the same with n_atoms known: out-of-range neighbour indices are clamped for the read (never an out-of-bounds access; the
 * reference's torch indexing raises a device assert) -- spk_edge_plan of the list reports SPK_ERR_INDEX */
int spk_pairwise_n_f32(const float* R, const int64_t* idx_i, const int64_t* idx_j, const float* offsets,
                       int64_t n_edges, int64_t n_atoms, float* r_ij, void* stream);
/* its backward w.r.t. R:  gR[a] = sum_{e: idx_j[e]==a} gr[e] - sum_{e: idx_i[e]==a} gr[e]
 * ([N,3], overwritten) -- where dE/dR_ij lands on the atoms (forces = -gR). */
int spk_pairwise_bwd_f32(const float* gr, const int64_t* idx_i, const int64_t* idx_j,
                         int64_t n_edges, int64_t n_atoms, float* gR, void* stream);
/* same result with the plan of the list: on sorted + symmetric lists (g->rowptr, g->rev) a segmented
 * row sum gR[a] = sum_{e in row(a)} (gr[rev[e]] - gr[e]) -- no atomics, deterministic; other lists
 * take the atomic kernel above. */
int spk_pairwise_bwd_graph_f32(const float* gr, const spk_graph_t* g, float* gR, void* stream);

/* ------------------------------------------------------------------ transform/neighborlist.py:438-507
 * (TorchNeighborList) and md/neighborlist_md.py:100-159 -- cell-list neighbour list on the device for
 * a batch of independent systems (molecules / MD replicas).  Result: every DIRECTED pair (i, j, S)
 * with |R_j - R_i + S.cell| < cutoff (i != j or S != 0), S integer shifts along the periodic axes,
 * offsets = S.cell, idx_i ascending (rowptr is its CSR), deterministic order inside a row,
 * symmetric by construction.
 *   R [n_atoms,3]; idx_m [n_atoms] int64 ascending system index or NULL (one system);
 *   cell [n_sys,3,3] row vectors or NULL; pbc [n_sys,3] bytes (torch.bool) or NULL (not periodic).
 * Two phases because the pair count is data dependent and the caller owns all memory:
 *   spk_nbl_count_f32  bins the atoms, counts, writes rowptr [n_atoms+1] and returns the number of
 *                      pairs through n_edges_host (synchronises the stream once);
 *   spk_nbl_fill_f32   writes idx_i, idx_j [n_edges] int64, shifts [n_edges,3] int32 (may be NULL)
 *                      and offsets [n_edges,3]; same R / idx_m / cutoff / workspace as the count. */
int64_t spk_nbl_workspace_bytes(int64_t n_atoms, int64_t n_sys);
int spk_nbl_count_f32(const float* R, const int64_t* idx_m, const float* cell, const uint8_t* pbc,
                      int64_t n_atoms, int64_t n_sys, float cutoff, void* workspace,
                      int32_t* rowptr, int64_t* n_edges_host, void* stream);
int spk_nbl_fill_f32(const float* R, const int64_t* idx_m, int64_t n_atoms, int64_t n_sys,
                     float cutoff, const void* workspace, const int32_t* rowptr, int64_t n_edges,
                     int64_t* idx_i, int64_t* idx_j, int32_t* shifts, float* offsets, void* stream);

/* ------------------------------------------------------------------ md/integrators.py
 * Elementwise MD steps on device arrays (any unit system; dt in the caller's time unit).
 * spk_md_half_step_f32      p += half_dt * F over n floats                       (:59-70)
 * spk_md_kick_drift_f32     first half step + velocity-Verlet main step fused    (:59-70, :97-110):
 *                           p += dt/2 F (skipped when F is NULL);  R += dt p / m,  masses [n_atoms];
 *                           with R_ref / flag (int32 [2]): flag[0] |= 1 when any atom is further than
 *                           sqrt(max_disp2) from R_ref (neighbour-list skin, md/neighborlist_md.py:80-90);
 *                           flag[1] = max(flag[1], bits of the largest squared one-step displacement)
 * spk_md_ring_polymer_step_f32   ring-polymer main step (:204-229) for the beads
 *                           [bead0, bead0+n_local) of this rank from ALL beads q_all, p_all
 *                           [n_beads, n_atoms, 3]; A [4, n_beads, n_beads] = C^T diag(P_ij) C for
 *                           (ij) = pp, pq, qp, qq (C: normal_model_transformation.py:38-68,
 *                           P: integrators.py:152-199); q_out, p_out [n_local, n_atoms, 3].  Optional skin test as
 *                           in spk_md_kick_drift_f32 (R_ref [n_local, n_atoms, 3] and flag int32 [2], or NULL). */
int spk_md_half_step_f32(float* p, const float* F, float half_dt, int64_t n, void* stream);
int spk_md_kick_drift_f32(float* R, float* p, const float* F, const float* masses, float dt,
                          int64_t n_atoms, const float* R_ref, float max_disp2, int32_t* flag,
                          void* stream);
int spk_md_ring_polymer_step_f32(const float* q_all, const float* p_all, const float* masses,
                                 const float* A, int32_t n_beads, int64_t n_atoms, int32_t bead0,
                                 int32_t n_local, float* q_out, float* p_out, const float* R_ref,
                                 float max_disp2, int32_t* flag, void* stream);

/* PILE-L thermostat of ring-polymer MD (md/simulation_hooks/thermostats_rpmd.py:33-119), applied at the begin and the end of
 * a step (md/simulation_hooks/thermostats.py:97-123): in normal modes p_nm' = c1_k p_nm + sqrt(m kB n_beads T) c2_k xi.  Folded
 * into bead space:  p_out[b] = sum_b' M[0][b][b'] p_all[b'] + sqrt(m) noise_scale sum_k M[1][b][k] xi_k  for the beads
 * [bead0, bead0 + n_local) of this rank; M [2, n_beads, n_beads] = (C^T diag(c1) C, C^T diag(c2)), noise_scale =
 * sqrt(kB n_beads T).  xi_k ~ N(0, 1) comes from a counter-based generator (Philox-4x32-10, key = seed, counter = (atom
 * component, mode pair, step, which)): it is a function of the counter alone, so bead-parallel ranks agree on it without an
 * exchange.  step: host value, or read from the device word step_dev when that is not NULL (HIP-graph replays);
 * which = 0 / 1 for the application at step begin / end.  n_beads <= 64. */
int spk_md_pile_f32(const float* p_all, const float* masses, const float* M, float noise_scale, uint64_t seed,
                    uint64_t step, const int64_t* step_dev, int32_t which, int32_t n_beads, int64_t n_atoms,
                    int32_t bead0, int32_t n_local, float* p_out, void* stream);

/* ------------------------------------------------------------------ atomistic/atomwise.py:69-88
 * The default output head, build_mlp(n_in, 1, n_layers=2) (nn/blocks.py:38-57) + sum over idx_m:
 *   y_n = w2 . act(W1 x_n + b1) + b2,   E[idx_m[n]] += y_n           (E [n_mol] is overwritten)
 * x [N, n_in], w1 [n_hidden, n_in], b1 [n_hidden] or NULL, w2 [n_hidden], b2 [1] or NULL,
 * idx_m [N] int64 (entries outside [0, n_mol) are ignored) or NULL (then E must be NULL too),
 * pre [N, n_hidden] receives the hidden pre-activations (NULL: not kept), y_atom [N] or NULL.
 * Fused kernel shapes: n_in % 32 == 0, n_hidden % 32 == 0, act in {SSP, SILU}
 * (spk_atomwise_supported); other heads run through spk_dense_f32 + spk_scatter_add_f32. */
int spk_atomwise_supported(int32_t n_in, int32_t n_hidden, int32_t act);
int spk_atomwise_fwd_f32(const float* x, const float* w1, const float* b1, const float* w2,
                         const float* b2, const int64_t* idx_m, int64_t n_atoms, int32_t n_in,
                         int32_t n_hidden, int32_t act, int64_t n_mol, float* pre, float* y_atom,
                         float* E, void* stream);
/* first-order backward w.r.t. x:  gx[n] = (gE[idx_m[n]] + gy_atom[n]) * W1^T (w2 * act'(pre_n));
 * either of (gE, idx_m) / gy_atom may be NULL. */
int spk_atomwise_bwd_f32(const float* gE, const float* gy_atom, const float* pre, const float* w1,
                         const float* w2, const int64_t* idx_m, int64_t n_atoms, int32_t n_in,
                         int32_t n_hidden, int32_t act, int64_t n_mol, float* gx, void* stream);

/* ------------------------------------------------------------------ nn/radial.py, nn/cutoff.py
 * d: [n] distances -> phi [n, n_rbf] (may be NULL), fcut [n] (may be NULL). */
int spk_radial_cutoff_f32(const float* d, int64_t n, const spk_radial_t* rb, float* phi,
                          float* fcut, void* stream);
/* backward of the pair (phi, fcut) w.r.t. d:  gd[n] = sum_k gphi[n,k] phi_k'(d) + gfcut[n] f'(d) */
int spk_radial_cutoff_bwd_f32(const float* d, int64_t n, const spk_radial_t* rb, const float* gphi,
                              const float* gfcut, float* gd, void* stream);
/* r_ij [E,3] -> d [E] (representation/schnet.py:156) and optionally unit vectors u [E,3]
 * (representation/painn.py:227-228) */
int spk_edge_norm_f32(const float* r_ij, int64_t n_edges, float* d, float* u, void* stream);

/* ------------------------------------------------------------------ nn/base.py:52-55 (Dense)
 * y[m, o] = act(sum_k x[m,k] w[o,k] + b[o]) + res[m,o].  w is the torch Linear weight
 * [n_out, k] row-major.  b, res, pre may be NULL; `pre` receives the pre-activation (saved for
 * backward).  k % 4 == 0 and n_out % 4 == 0 select the fp32-MFMA kernel (partial tiles are masked), otherwise the
 * simple kernel runs. */
int spk_dense_f32(const float* x, const float* w, const float* b, const float* res, float* y,
                  float* pre, int64_t m, int32_t k, int32_t n_out, int32_t act, void* stream);
/* input gradient: dx[m,k] = sum_o (dy[m,o] * act'(pre[m,o])) w[o,k]  (+ res[m,k]).
 * pre may be NULL iff act == SPK_ACT_NONE. */
int spk_dense_bwd_input_f32(const float* dy, const float* pre, const float* w, const float* res,
                            float* dx, int64_t m, int32_t k, int32_t n_out, int32_t act,
                            void* stream);

/* A Dense layer on a (value, tangent) PAIR of activations in ONE launch -- the unit of the force-matching engine (spk_*_fm_*), where
 * every activation of nn/base.py:52-55 travels with its directional derivative along t = -dL/dF (atomistic/response.py:59-68 under
 * create_graph=True).  p = x w^T (trans = 0, w [n_out, k_in]) or x w (trans = 1, w [k_in, n_out]) for both members, then
 *   SPK_DD_FWD       p_v += b;  y_v = act(p_v) + res_v,  y_t = act'(p_v) p_t + res_t;  with fc (act NONE): y_v = p_v fc[m], y_t = p_t fc[m] + p_v fc1[m]
 *                    (the cutoff product of schnet.py:62 / painn.py:232-236 with its d-derivative);  pre_v / pre_t (optional) receive p_v, p_t
 *   SPK_DD_TANGENT   x_v == NULL:  y_t = act'(pre_v_in) p_t + res_t;  pre_t (optional) receives p_t
 *   SPK_DD_DUAL_BWD  (p_v, p_t) = (g_z, h_z), the cotangents of z = act(a), z_t = act'(a) a_t with a = pre_v_in, a_t = pre_t_in (NULL = 0):
 *                    y_v = g_z act'(a) + h_z act''(a) a_t,  y_t = h_z act'(a)
 * All row-major [m, .]; every pointer 16-byte aligned; k_in % 4 == 0, n_out % 4 == 0 and at most 4 tiles of 32 x 32 per compute unit
 * (spk_dense_dual_supported): one workgroup per tile; the forward pair (FWD, trans = 0) also runs at any size on a grid-stride kernel
 * (spk_dense_dual_fwd_supported) -- other modes of larger problems are two spk_dense_f32 launches and an element-wise one. */
#define SPK_DD_FWD 0
#define SPK_DD_TANGENT 1
#define SPK_DD_DUAL_BWD 2
typedef struct {
  const float *x_v, *x_t;             /* [m, k_in] */
  const float *w, *b;
  const float *res_v, *res_t;         /* [m, n_out] or NULL */
  const float *pre_v_in, *pre_t_in;   /* [m, n_out] saved pre-activations (TANGENT, DUAL_BWD) */
  const float *fc, *fc1;              /* [m] or NULL */
  float *y_v, *y_t, *pre_v, *pre_t;   /* [m, n_out] */
  int64_t m;
  int32_t k_in, n_out, act, mode, trans;
} spk_dense_dual_t;
int spk_dense_dual_supported(int64_t m, int32_t k_in, int32_t n_out);
int spk_dense_dual_fwd_supported(int64_t m, int32_t k_in, int32_t n_out);   /* mode FWD with trans = 0: any number of tiles (grid-stride kernel) */
int spk_dense_dual_f32(const spk_dense_dual_t* d, void* stream);

/* Chain of up to 3 Dense layers in ONE launch (e.g. f2out.0 -> f2out.1 (+residual) -> next in2f of
 * representation/schnet.py:33-36,60,69,168, or the input-gradient transposes of such a chain).  The
 * activations of a 32-row tile stay in LDS between layers.  Layer l maps [m, k_l] -> [m, n_out_l] with
 * k_{l+1} == n_out_l.  Shapes outside (k % 8 == 0, n_out % 32 == 0, <= 384) run layer by layer on the
 * single-layer kernels and then need the `tmp` buffers for outputs that are not stored. */
typedef struct {
  const float* w;        /* torch Linear weight [n_out, k]; with trans != 0 the layer is the input-gradient
                            of Linear([n, k_w] -> ...): w is [k, n_out] and y = x w */
  const float* b;        /* [n_out] or NULL */
  const float* res;      /* [m, n_out] or NULL, added after the activation (may alias out) */
  float* out;            /* [m, n_out] or NULL (result not stored, only handed to the next layer) */
  float* pre_out;        /* [m, n_out] or NULL: pre-activation, saved for backward */
  const float* post_pre; /* [m, n_out] or NULL: the copy handed to the NEXT layer is multiplied by
                            post_act'(post_pre) (backward through an activation) */
  int32_t k, n_out, act, trans, post_act;
} spk_chain_layer_t;

typedef struct {
  int32_t n_layers;      /* 1..3 */
  int32_t in_act;        /* with in_pre: the input is multiplied by in_act'(in_pre) */
  int64_t m;             /* rows */
  const float* in;       /* [m, k_0] */
  const float* in_pre;   /* [m, k_0] or NULL */
  float* zero_ptr;       /* optional: buffer cleared by the same launch (must not alias any operand) */
  int64_t zero_count;    /* floats */
  float* tmp[2];         /* [m, max n_out] scratch, only used by the layer-by-layer path */
  spk_chain_layer_t layers[3];
} spk_chain_t;

int spk_dense_chain_f32(const spk_chain_t* chain, void* stream);
/* tuning / test hook: force the row-tile height of the fused chain kernel (16 or 32; 0 = by size:
 * 16-row tiles on v_mfma_f32_16x16x4_f32 while 32-row tiles would not give every CU two workgroups). */
/* Packed image of a Linear weight w [n_out, k_in] for the fused chain kernels (layer `trans` code 2):
 *   transposed == 0: the forward layer y = x W^T  (contraction k_in,  width n_out)
 *   transposed == 1: the input-gradient layer gx = gy W (contraction n_out, width k_in)
 *   P[((t KB + ug) 64 + lane) 4 + v] = A[32 t + (lane & 31)][8 ug + 4 (lane >> 5) + v],  KB = contraction / 8;
 * contraction % 8 == 0 and width % 32 == 0.  packed holds n_out * k_in floats. */
int spk_pack_weight_f32(const float* w, int32_t n_out, int32_t k_in, int32_t transposed, float* packed, void* stream);
void spk_chain_set_rows(int32_t rows);
/* tuning aid: device buffer of >= 16 int64 receiving cycle stamps of workgroup 0 (NULL: off) */
void spk_chain_set_debug_buffer(void* p);

/* ------------------------------------------------------------------ representation/schnet.py:60-67
 * Fused continuous-filter convolution of one interaction block:
 *   W_e = (ssp(phi(d_e) W1^T + b1) W2^T + b2) * fcut(d_e);   y[i] = sum_{e->i} h[idx_j[e]] * W_e
 * h: [N, nf] (output of in2f), r_ij: [E,3], filter weights w1 [nf, n_rbf], b1 [nf], w2 [nf, nf],
 * b2 [nf].  y [N, nf] is fully overwritten.  Nothing of size E x nf is ever written to memory. */
int spk_schnet_cfconv_fwd_f32(const spk_graph_t* g, const spk_radial_t* rb, const float* h,
                              const float* r_ij, const float* w1, const float* b1,
                              const float* w2, const float* b2, int32_t nf, float* y,
                              void* stream);
/* First-order backward w.r.t. h and r_ij (what Forces, atomistic/response.py:59-76, asks for):
 *   gh[j] = sum_{e: idx_j[e]==j} gy[idx_i[e]] * W_e        (fully overwritten)
 *   gr[e] += (sum_f gy[i,f] h[j,f] dW_e[f]/dd) * r_e/d_e    (ACCUMULATED: interactions share r_ij)
 * With g->symmetric the transposed sum runs as a second row-local segmented reduction (no
 * atomics); otherwise float atomics are used. */
int spk_schnet_cfconv_bwd_f32(const spk_graph_t* g, const spk_radial_t* rb, const float* h,
                              const float* gy, const float* r_ij, const float* w1,
                              const float* b1, const float* w2, const float* b2, int32_t nf,
                              float* gh, float* gr, void* stream);

/* Kernel-tuning aid: device buffer (>= 32 int64) that receives shader-clock stamps of wave 0 of workgroup 0
 * at the phase boundaries of the pair kernels; NULL disables it (default). */
void spk_cfconv_set_debug_buffer(void* device_buffer);

/* Whole SchNet representation (representation/schnet.py:147-173), eval-mode force path.
 * Parameter block: pointers to the reference state_dict tensors of every interaction. */
typedef struct {
  const float* in2f_w;  /* interactions.l.in2f.weight             [nf, F]  */
  const float* fn_w1;   /* interactions.l.filter_network.0.weight [nf, n_rbf] */
  const float* fn_b1;   /* interactions.l.filter_network.0.bias   [nf] */
  const float* fn_w2;   /* interactions.l.filter_network.1.weight [nf, nf] */
  const float* fn_b2;   /* interactions.l.filter_network.1.bias   [nf] */
  const float* f2out_w1; /* interactions.l.f2out.0.weight         [F, nf] */
  const float* f2out_b1; /* interactions.l.f2out.0.bias           [F]  */
  const float* f2out_w2; /* interactions.l.f2out.1.weight         [F, F] */
  const float* f2out_b2; /* interactions.l.f2out.1.bias           [F]  */
  /* optional transposed copies ([in, out] row-major) of the three atom-wise weights: with them the
   * forward chains read the weights with coalesced loads; NULL = use the [out, in] originals */
  const float* in2f_wT;   /* [F, nf] */
  const float* f2out_w1T; /* [nf, F] */
  const float* f2out_w2T; /* [F, F]  */
} spk_schnet_layer_t;

typedef struct {
  int32_t n_atom_basis;   /* F */
  int32_t n_filters;      /* nf */
  int32_t n_interactions;
  int32_t reserved;       /* bit 0: `saved` was sized with spk_schnet_saved_floats_graph(): the forward keeps
                             the raw filter outputs of every (undirected) edge so that the backward runs
                             only the derivative GEMM */
  const spk_schnet_layer_t* layers; /* HOST array of n_interactions entries (device pointers inside) */
  const float* wpack;     /* optional (may be NULL): packed images of the Dense weights, spk_schnet_packed_floats()
                             floats filled by spk_schnet_pack_weights_f32(); refresh it when the weights change */
} spk_schnet_t;

/* floats needed in `saved` (kept from forward to backward) and `scratch` (per call) */
/* Packed weight images: the fused Dense-chain kernels read every weight as fully coalesced 16-byte-per-lane
 * tiles (spk_pack_weight_f32).  0 floats = these shapes have no packed form (the drivers then read the plain
 * / transposed weights); otherwise allocate that many floats, fill them once per weight update and put the
 * pointer into m->wpack (the `wpack` field of the struct passed here is ignored). */
int64_t spk_schnet_packed_floats(const spk_schnet_t* m);
int spk_schnet_pack_weights_f32(const spk_schnet_t* m, float* wpack, void* stream);
int64_t spk_schnet_saved_floats(const spk_schnet_t* m, int64_t n_atoms);
int64_t spk_schnet_saved_floats_graph(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb);
int64_t spk_schnet_scratch_floats(const spk_schnet_t* m, int64_t n_atoms);
/* x0 [N,F] = embedding rows (+ electronic embeddings) computed by the caller;
 * x_out [N,F] = scalar_representation. */
int spk_schnet_forward_f32(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb,
                           const float* x0, const float* r_ij, float* x_out, float* saved,
                           float* scratch, void* stream);
/* gx_out [N,F] = dL/d scalar_representation  ->  gr [E,3] = dL/d r_ij (overwritten) and
 * gx0 [N,F] = dL/d x0 (may be NULL). */
int spk_schnet_backward_f32(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb,
                            const float* gx_out, const float* r_ij, const float* saved,
                            float* scratch, float* gr, float* gx0, void* stream);

/* ------------------------------------------------------------------ the standard potential in two launches
 * PairwiseDistances -> SchNet -> Atomwise(sum) -> Forces  (atomistic/distances.py:14-26, representation/schnet.py:147-173,
 * atomistic/atomwise.py:69-88 with the default head build_mlp(F, 1, n_layers=2), atomistic/response.py:59-76) for the lists the
 * molecule-resident kernels cover (block-diagonal, <= 32 atoms and <= 384 pairs per block, F = n_filters = 128, n_rbf <= 24):
 * r_ij is formed from the positions inside the forward launch, the energy head runs on the atom tile that is still in LDS, the
 * backward launch starts from dL/dE and ends at dL/dR.  `m` as for spk_schnet_forward_f32 with bit 0 of `reserved` set, `saved`
 * of spk_schnet_saved_floats_graph() floats, pre_h [N, n_hidden] kept from forward to backward.  _supported() returns 1 when
 * the pair (model, list) is covered; the two entry points return SPK_ERR_ARG otherwise (run the separate entry points then). */
typedef struct {
  const float* w1;   /* outnet.0.weight [n_hidden, F] */
  const float* w1t;  /* its transpose   [F, n_hidden] */
  const float* b1;   /* outnet.0.bias   [n_hidden] */
  const float* w2;   /* outnet.1.weight [1, n_hidden] */
  const float* b2;   /* outnet.1.bias   [1] (may be NULL) */
  int32_t n_hidden;  /* multiple of 32, <= 128 */
  int32_t act;       /* SPK_ACT_SSP | SPK_ACT_SILU */
} spk_head_t;
int spk_schnet_potential_supported(const spk_schnet_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb);
/* x0 [N,F] embedding rows, R [N,3], offsets [E,3] or NULL, idx_m [N] -> x_out [N,F], E [n_mol] (overwritten) */
int spk_schnet_potential_forward_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb,
                                     const float* x0, const float* R, const float* offsets, const int64_t* idx_m, int64_t n_mol,
                                     float* x_out, float* E, float* pre_h, float* saved, void* stream);
/* gE [n_mol] = dL/dE, gx_out [N,F] = dL/d scalar_representation or NULL -> gR [N,3] = dL/dR (overwritten; forces = -gR for
 * L = sum E), gx0 [N,F] = dL/dx0 or NULL */
int spk_schnet_potential_backward_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb,
                                      const float* gE, const float* gx_out, const float* R, const float* offsets, const int64_t* idx_m,
                                      const float* pre_h, const float* saved, float* gR, float* gx0, void* stream);

/* Energies and FORCES (= -dE/dR of the summed energy) from the same two launches and nothing else.  x0 may be NULL: the rows of
 * the nuclear embedding table emb [n_types, F] (schnet.py:126-128) are then looked up by Z inside the forward launch.
 * all_inside != 0: the caller guarantees that the atoms of every molecule lie inside one group of the plan and that every molecule
 * has an atom; the energies are then stored instead of accumulated (no clearing launch). */
int spk_schnet_potential_forces_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb,
                                    const float* x0, const float* emb, const int64_t* Z, int32_t n_types, const float* R,
                                    const float* offsets, const int64_t* idx_m, int64_t n_mol, int32_t all_inside, float* x_out,
                                    float* E, float* F, float* pre_h, float* saved, void* stream);

/* EXPERIMENT (eval only, not on any default path; scripts/tab_filter_experiment.py): continuous-filter convolution
 * (representation/schnet.py:60-67) with the filter W_l(d) f_c(d) read from a cubic-Hermite table instead of the filter network:
 *   y[i, c] = sum_{e in row(i)} h[idx_j[e], c] * T_c(|r_e|),  table [n_knots, 128, 4] = (value, slope * step, next value - value, next slope * step) at d = n d_max / (n_knots - 1),
 * zero at and beyond the cutoff.  Sorted list (g->rowptr), nf = 128; y [N, 128] is overwritten. */
int spk_cfconv_tab_f32(const spk_graph_t* g, const float* r_ij, const float* h, const float* table, int32_t n_knots, float d_max,
                       float cutoff, int32_t nf, float* y, void* stream);
/* Attach a table to the interaction whose filter_network.1.weight lies at device address `key` (table == NULL detaches it):
 * spk_schnet_forward_f32 / _backward_f32 then run that interaction's convolution (and its first-order backward, on symmetric
 * sorted lists) through the table kernels.  The caller rebuilds the tables when the weights change.  Not used by default. */
int spk_filter_table_set(const float* key, const float* table, int32_t n_knots, float d_max);
void spk_filter_table_clear(void);
/* A table is a snapshot of the weights: set_stamp records the version of the weights it was built from (any non-zero number the caller
 * can reproduce, e.g. the tensor's version counter + 1); drop_if_stale(key, stamp) detaches the table when its recorded stamp differs
 * (returns 1), so that changed weights run the exact filter network instead of a stale table. */
int spk_filter_table_set_stamp(const float* key, uint64_t stamp);
int spk_filter_table_drop_if_stale(const float* key, uint64_t stamp);

/* Kernel-tuning aid of the molecule-resident SchNet kernels (spk_schnet_mol.hip: block-diagonal lists with <= 32 atoms per
 * block run every interaction inside one workgroup): device buffer of int64 receiving cycle stamps -- entries [0, 128): thread 0
 * of workgroup 0 at the phase boundaries; [128 + 4 b, 128 + 4 b + 4): start / end of workgroup b of the backward launch (real time and
 * shader clock) -- so it must hold 128 + 4 * (number of groups) entries; NULL disables it (default). */
void spk_schnet_mol_set_debug_buffer(void* device_buffer);
/* The same for the molecule-resident PaiNN kernels (spk_painn_mol.hip; representation/painn.py:207-256 with q / mu / context rows
 * of a <= 32-atom block in LDS, all interactions in one launch): >= 256 int64; entry 0 = group start, 1 + 8 l .. 8 + 8 l = phase
 * boundaries of interaction l of the forward (thread 0 of workgroup 0); entries 64.. the backward, 128 + 16 w .. the message phase
 * of wave w in its top interaction.  NULL disables it (default). */
void spk_painn_mol_set_debug_buffer(void* device_buffer);

/* ------------------------------------------------------------------ representation/painn.py:31-67
 * Fused PaiNN message of one interaction block (filters never materialised; the reference
 * allocates [E,1,3F*n_int], painn.py:232):
 *   Phi_e = (phi(d_e) Wf^T + bf) * fcut(d_e)          (Wf, bf: the 3F rows of this layer)
 *   m_e = Phi_e * c[idx_j[e]]  -> (m_q | m_R | m_mu)
 *   q_out[i] = q[i] + sum m_q;  mu_out[i] = mu[i] + sum (m_R (x) u_e + m_mu * mu[idx_j[e]])
 * c: [N,3F] output of interatomic_context_net, q: [N,F], mu: [N,3,F]. */
int spk_painn_message_fwd_f32(const spk_graph_t* g, const spk_radial_t* rb, const float* c,
                              const float* q, const float* mu, const float* r_ij,
                              const float* wf, const float* bf, int32_t F, float* q_out,
                              float* mu_out, void* stream);
/* First-order backward of the message w.r.t. c, mu and r_ij given gq_out [N,F], gmu_out [N,3,F]:
 *   gc [N,3F] (overwritten), gmu [N,3,F] (overwritten, includes the residual path gmu_out),
 *   gr [E,3] (ACCUMULATED).  The residual path of q is the caller's (gq = gq_out + ctx-net bwd). */
int spk_painn_message_bwd_f32(const spk_graph_t* g, const spk_radial_t* rb, const float* c,
                              const float* mu, const float* gq_out, const float* gmu_out,
                              const float* r_ij, const float* wf, const float* bf, int32_t F,
                              float* gc, float* gmu, float* gr, void* stream);

/* Message kernel family: 0 = automatic -- the MFMA tile kernels where they were measured faster than the row kernels
 * (forward: lists of >= 2^19 edges without a skin; backward: the geometry-only launch of the first interaction of an
 * eval-mode backward, lists without a skin), 1 = tile kernels whenever the shape has one (F in {64, 128},
 * 16 <= n_rbf <= 39 forward / <= 31 backward, sorted (+ symmetric) list), -1 = never.  All compute the same function; the
 * tests run each family against the oracle and the reference fixtures. */
void spk_painn_set_tile(int32_t mode);
/* row kernels of the PaiNN message with the radial values of a 64-edge chunk in a wave-private LDS table (F = 128, n_rbf <= 20):
 * -1 = on large lists (default), 0 = never, 1 = whenever the shape has the instance */
void spk_painn_set_row_table(int32_t mode);
/* Row-tile backward of the PaiNN message (round 6; representation/painn.py:50-66 transposed): a wavefront per row of the sorted, symmetric list,
 * the filter and its slope from a split-precision GEMM (fp16 high / low operand pairs, fp32 accumulation) over 32-pair chunks of the row, no
 * atomics; the full backward runs as a geometry launch and a transposed-sums launch.  F = 128, 16 <= n_rbf <= 31, split path on
 * (spk_set_split).  0 = automatic (lists of >= 2^19 pairs, with or without a skin), 1 = whenever the shape has it, -1 = never. */
void spk_painn_set_rowtile(int32_t mode);

/* representation/painn.py:92-117 -- the elementwise parts of PaiNNMixing around its three
 * Dense layers.  mix [N,3,2F] = mu_channel_mix(mu) = (V | W).
 *   ctx[n] = (q[n] | sqrt(sum_x V[n,x]^2 + eps))                         [N,2F] */
int spk_painn_mix_ctx_f32(const float* q, const float* mix, int64_t n_atoms, int32_t F, float eps,
                          float* ctx, void* stream);
/*   a [N,3F] = intraatomic_context_net(ctx) = (a_q | a_mu | a_qmu)
 *   q_out = q + a_q + a_qmu * sum_x V W ;  mu_out = mu + a_mu * W */
int spk_painn_mix_update_f32(const float* q, const float* mu, const float* mix, const float* a,
                             int64_t n_atoms, int32_t F, float* q_out, float* mu_out,
                             void* stream);
/* backward of the update w.r.t. a and mix (the direct residual paths gq_out -> q and
 * gmu_out -> mu are the caller's): ga [N,3F], gmix [N,3,2F] (overwritten; `mu` is unused). */
int spk_painn_mix_update_bwd_f32(const float* mu, const float* mix, const float* a,
                                 const float* gq_out, const float* gmu_out, int64_t n_atoms,
                                 int32_t F, float* ga, float* gmix, void* stream);
/* backward of the ctx assembly given g_ctx [N,2F] (from the Dense backward):
 *   gmix[:, :, :F] += g_ctx[:, F:] V / |V| ;  gq = gq_out + g_ctx[:, :F] */
int spk_painn_mix_ctx_bwd_f32(const float* mix, const float* g_ctx, const float* gq_out,
                              int64_t n_atoms, int32_t F, float eps, float* gmix_inout,
                              float* gq, void* stream);

/* Whole PaiNN representation (representation/painn.py:207-256), eval-mode force path. */
typedef struct {
  const float* ctx_w1;  /* interactions.l.interatomic_context_net.0.weight [F, F]  */
  const float* ctx_b1;  /* ...0.bias [F] */
  const float* ctx_w2;  /* interactions.l.interatomic_context_net.1.weight [3F, F] */
  const float* ctx_b2;  /* ...1.bias [3F] */
  const float* filt_w;  /* filter_net.weight rows [3F*l, 3F*(l+1)) (row 0 if shared_filters) [3F, n_rbf] */
  const float* filt_b;  /* filter_net.bias, same rows [3F] */
  const float* mix_w;   /* mixing.l.mu_channel_mix.weight [2F, F] */
  const float* ictx_w1; /* mixing.l.intraatomic_context_net.0.weight [F, 2F] */
  const float* ictx_b1; /* ...0.bias [F] */
  const float* ictx_w2; /* mixing.l.intraatomic_context_net.1.weight [3F, F] */
  const float* ictx_b2; /* ...1.bias [3F] */
  /* optional transposed copies ([in, out] row-major) for coalesced weight reads in the forward; NULL = unused */
  const float* ctx_w1T;  /* [F, F]   */
  const float* ctx_w2T;  /* [F, 3F]  */
  const float* mix_wT;   /* [F, 2F]  */
  const float* ictx_w1T; /* [2F, F]  */
  const float* ictx_w2T; /* [F, 3F]  */
} spk_painn_layer_t;

typedef struct {
  int32_t n_atom_basis;
  int32_t n_interactions;
  float epsilon;          /* PaiNNMixing epsilon (painn.py:73) */
  int32_t reserved;
  const spk_painn_layer_t* layers; /* HOST array of n_interactions entries */
  const float* wpack;     /* optional packed weight images (spk_painn_packed_floats / spk_painn_pack_weights_f32) */
} spk_painn_t;

int64_t spk_painn_packed_floats(const spk_painn_t* m);
int spk_painn_pack_weights_f32(const spk_painn_t* m, float* wpack, void* stream);
int64_t spk_painn_saved_floats(const spk_painn_t* m, int64_t n_atoms);
int64_t spk_painn_scratch_floats(const spk_painn_t* m, int64_t n_atoms);
/* q0 [N,F] = embedding rows; outputs scalar_representation q_out [N,F] and
 * vector_representation mu_out [N,3,F]. */
int spk_painn_forward_f32(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb,
                          const float* q0, const float* r_ij, float* q_out, float* mu_out,
                          float* saved, float* scratch, void* stream);
/* gq_out / gmu_out may be NULL (treated as zero, not both) -> gr [E,3] (overwritten),
 * gq0 [N,F] (may be NULL). */
int spk_painn_backward_f32(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb,
                           const float* gq_out, const float* gmu_out, const float* r_ij,
                           const float* saved, float* scratch, float* gr, float* gq0,
                           void* stream);

/* The standard potential in two launches (see spk_schnet_potential_forces_f32) for PaiNN (representation/painn.py:207-256; head with n_hidden = 64, i.e. the default build_mlp(128, 1, n_layers = 2)):
 * pair vectors from the positions, rows of the embedding table (q0 == NULL), every interaction and the energy head in the forward
 * launch; head gradient, every interaction and the forces (fixed summation order over each atom's edges and their reverse edges:
 * bit-reproducible, no atomics) in the backward launch.  Lists: as spk_painn_forward_f32's molecule-resident path (block-diagonal,
 * symmetric, <= 32 atoms and <= 384 pairs per block), g->idx_i and g->rev given.  q_out [N,F], mu_out [N,3,F], E [n_mol], F [N,3],
 * pre_h [N,64], saved = spk_painn_saved_floats(), scratch = spk_painn_scratch_floats() floats. */
int spk_painn_potential_supported(const spk_painn_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb);
int spk_painn_potential_forces_f32(const spk_painn_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb,
                                   const float* q0, const float* emb, const int64_t* Z, int32_t n_types, const float* R,
                                   const float* offsets, const int64_t* idx_m, int64_t n_mol, int32_t all_inside, float* q_out,
                                   float* mu_out, float* E, float* F, float* pre_h, float* saved, float* scratch, void* stream);


/* ------------------------------------------------------------------ deployment runtime
 * Torch-free replacement of the deployed-model path: src/scripts/spkdeploy:16-40 writes a TorchScript archive
 * (+ "cutoff" metadata), interfaces/lammps/pair_schnetpack.cpp:128 loads it with torch::jit::load and :328 calls
 * model.forward on a dict of _positions / _atomic_numbers / _idx_i / _idx_j / _offsets / _idx_m / _cell and
 * reads "energy" and "forces" back (:330-350).  Here schnetpack_amd.deploy.export_potential() writes a flat
 * weight file (header + named fp32 tensors, little endian) and the functions below load it and run the same
 * model: PairwiseDistances -> SchNet / PaiNN -> Atomwise(energy, sum) -> Forces -> AddOffsets(mean / atomref).
 * A handle owns its device memory (weights, packed weight images, grow-only work buffers) and one stream; calls
 * on one handle must not overlap.  ALL pointers of this section are HOST pointers.
 *
 * spk_potential_load / _from_memory   parse + upload; *out receives the handle.
 * spk_potential_info                  host_info[0] kind (0 SchNet, 1 PaiNN), [1] n_atom_basis, [2] n_interactions,
 *                                     [3] n_rbf, [4] radial kind, [5] n_filters, [6] head hidden width,
 *                                     [7] embedding rows; *host_cutoff = the "cutoff" metadata of spkdeploy:36.
 * spk_potential_compute               explicit neighbour list, as pair_schnetpack.cpp:196-283 assembles it:
 *     z [n_atoms] int64, R [n_atoms,3] fp32, idx_i / idx_j [n_edges] int64 in ANY order (the LAMMPS list is
 *     ordered by local index, not by tag; the runtime stable-sorts by idx_i on the host when needed),
 *     offsets [n_edges,3] or NULL, idx_m [n_atoms] ascending or NULL (one system, n_mol = 1).
 *     energy [n_mol] and forces [n_atoms,3] are written.
 * spk_potential_compute_cell          the runtime builds the list itself on the device (spk_nbl_*): cell [n_mol,3,3]
 *     row vectors or NULL, pbc [n_mol,3] bytes or NULL; skin > 0 keeps a (cutoff + skin) list until an atom
 *     has moved more than skin / 2 since the list was built (md/neighborlist_md.py:80-90) or n_atoms / cell /
 *     idx_m change.  host_stats (may be NULL): [0] pairs in the list, [1] 1 if the list was rebuilt by this call.
 * Errors: SPK_ERR_ARG (bad file / unsupported head or shape / atomic number outside the embedding),
 * SPK_ERR_INDEX (neighbour index out of range), SPK_ERR_HIP. */
typedef struct spk_potential spk_potential_t;
int spk_potential_load(const char* path, spk_potential_t** out);
int spk_potential_from_memory(const void* host_blob, int64_t n_bytes, spk_potential_t** out);
void spk_potential_free(spk_potential_t* p);
int spk_potential_info(const spk_potential_t* p, int32_t* host_info, float* host_cutoff);
int spk_potential_compute(spk_potential_t* p, int64_t n_atoms, const int64_t* host_z, const float* host_R,
                          int64_t n_edges, const int64_t* host_idx_i, const int64_t* host_idx_j,
                          const float* host_offsets, int64_t n_mol, const int64_t* host_idx_m,
                          float* host_energy, float* host_forces);
int spk_potential_compute_cell(spk_potential_t* p, int64_t n_atoms, const int64_t* host_z, const float* host_R,
                               int64_t n_mol, const int64_t* host_idx_m, const float* host_cell,
                               const uint8_t* host_pbc, float skin, float* host_energy, float* host_forces,
                               int64_t* host_stats);

/* ------------------------------------------------------------------ small helpers
 * out[n, :] = table[z[n], :]  (nn.Embedding lookup, schnet.py:161 / painn.py:239) */
int spk_embedding_f32(const float* table, const int64_t* z, int64_t n, int32_t F, float* out,
                      void* stream);
/* y = a + b (n floats) */
int spk_add_f32(const float* a, const float* b, int64_t n, float* y, void* stream);

/* ------------------------------------------------------------------ training regime (force matching: every operator is
 * differentiated twice, atomistic/response.py:59-68 with create_graph = training).  A family of kernels closed under
 * differentiation -- the derivative of each member is another member -- so that the recorded backward and its backward are
 * these launches (spk_train.hip).  All tensors dense row-major fp32; `a`, `c` operands may be NULL.
 *
 * out[t] = a[t] * act^(order)(z[t]) + c[t]      (act: SPK_ACT_*, derivative order 0..3; nn/activations.py:9-22, F.silu) */
int spk_act_mul_f32(const float* a, const float* z, const float* c, int64_t n, int32_t act, int32_t order,
                    float* out, void* stream);
/* G[O,K] = U[n,O]^T X[n,K], gb[O] = column sums of U (gb may be NULL): weight / bias gradients of Dense (nn/base.py:52-55).
 * spk_gemm_tn_plan reports the slice count of the contraction; with more than one slice the caller provides `ws`
 * (ws_floats floats) and `tickets` (n_tiles zero-initialised uint32; the kernel leaves them zero).  Deterministic. */
int spk_gemm_tn_plan(int64_t n, int32_t O, int32_t K, int32_t* n_slices, int64_t* ws_floats, int32_t* n_tiles);
int spk_gemm_tn_f32(const float* U, const float* X, int64_t n, int32_t O, int32_t K, float* G, float* gb, float* ws,
                    uint32_t* tickets, void* stream);
/* the same with the bias gradient restricted to the rows [0, n_bias) of U ([value rows ; tangent rows]-stacked operands of the
 * force-matching engine below carry a bias on the value rows only) */
int spk_gemm_tn_nb_f32(const float* U, const float* X, int64_t n, int32_t O, int32_t K, float* G, float* gb, int64_t n_bias,
                       float* ws, uint32_t* tickets, void* stream);
/* The two independent products of a Dense backward in ONE launch: out = a w (trans = 1: a [m, n_out], w [n_out, k] -> [m, k]) or
 * a w^T (trans = 0: a [m, k] -> [m, n_out]), and (G, gb) as spk_gemm_tn_f32.  Widths of the first product must be multiples of 4. */
int spk_gemm_pair_f32(const float* a, const float* w, int32_t trans, int64_t m, int32_t k, int32_t n_out, float* out,
                      const float* U, const float* X, int64_t n, int32_t O, int32_t K, float* G, float* gb, float* ws,
                      uint32_t* tickets, void* stream);
/* y[idx_out[e], :] += x[idx_src[e], :] * W[e, :]   (schnet.py:64-66 on materialised filters; y [n_out, F] overwritten).
 * rowptr_out = CSR row pointers of an ascending idx_out (deterministic segmented sum) or NULL (float atomics).
 * idx_src may be NULL: identity (x has one row per pair). */
int spk_cfconv_edge_f32(const float* x, const float* W, const int64_t* idx_out, const int64_t* idx_src,
                        const int32_t* rowptr_out, int64_t n_edges, int64_t n_out, int64_t n_src, int32_t F, float* y,
                        void* stream);
/* out[e, :] = a[idx_a[e], :] * b[idx_b[e], :]   (the derivative of the above w.r.t. W); either index may be NULL: identity */
int spk_edge_mul_f32(const float* a, const float* b, const int64_t* idx_a, const int64_t* idx_b, int64_t n_edges,
                     int64_t n_a, int64_t n_b, int32_t F, float* out, void* stream);
/* out[e, r] = a[e] * d^order/dd^order phi_r(d[e])   and   out[e] = a[e] * sum_r G[e, r] d^order/dd^order phi_r(d[e]);
 * rb->kind 0 / 1 = the radial bases (nn/radial.py), 2 = the cosine cutoff (nn/cutoff.py:14-33) with one "basis function";
 * order 0..3 */
int spk_radial_d_f32(const float* d, const float* a, int64_t n, const spk_radial_t* rb, int32_t order, float* out,
                     void* stream);
int spk_radial_c_f32(const float* G, const float* d, const float* a, int64_t n, const spk_radial_t* rb, int32_t order,
                     float* out, void* stream);
/* out[r, f] = W[r, f] * s[r]   (Wij * rcut_ij[:, None], schnet.py:61)   and   out[r] = sum_f a[r, f] * b[r, f] */
int spk_rowscale_f32(const float* W, const float* s, int64_t rows, int32_t F, float* out, void* stream);
/* 3-vector algebra of PaiNN (painn.py:55-66, 99-117) on V-type [M, 3, F], s-type [M, F] and u-type [M, 3] operands; V / s operands
 * carry a row stride in floats (ldA / ldB) so the halves of a split tensor are read in place; `out` is dense.
 *   SCALE:    out[m,k,f] = A[m,k,f] B[m,f]            (A: V, B: s)      DOT:     out[m,f] = sum_k A[m,k,f] B[m,k,f]   (A, B: V)
 *   OUTER:    out[m,k,f] = A[m,f] B[m,k]              (A: s, B: u)      CONTRACT: out[m,f] = sum_k A[m,k,f] B[m,k]    (A: V, B: u)
 *   ROWDOT:   out[m,k]   = sum_f A[m,k,f] B[m,f]      (A: V, B: s)
 * The five are closed under differentiation (spk_train.hip). */
#define SPK_VEC3_SCALE 0
#define SPK_VEC3_DOT 1
#define SPK_VEC3_OUTER 2
#define SPK_VEC3_CONTRACT 3
#define SPK_VEC3_ROWDOT 4
int spk_vec3_f32(int32_t op, const float* A, int64_t ldA, const float* B, int64_t ldB, int64_t M, int32_t F, float* out,
                 void* stream);
int spk_rowdot_f32(const float* a, const float* b, int64_t rows, int32_t F, float* out, void* stream);

/* Force-matching loss of a training step, loss = wE mean((E - E_t)^2) + wF mean((F - F_t)^2) (AtomisticTask.loss_fn with two MSE
 * outputs, task.py:59-66, 142-146), and its gradients gE [M], gF [n3] w.r.t. E and F in one launch; spk_fm_loss_bwd_f32 scales them by the
 * incoming gradient g[0] (device scalar).  loss [1]. */
int spk_fm_loss_f32(const float* E, const float* E_t, int64_t M, const float* F, const float* F_t, int64_t n3, float wE, float wF,
                    float* loss, float* gE, float* gF, void* stream);
int spk_fm_loss_bwd_f32(const float* g, const float* gE, int64_t M, const float* gF, int64_t n3, float* outE, float* outF, void* stream);

/* AdamW (torch.optim.AdamW arithmetic: decoupled weight decay, bias corrections, eps outside the corrected root -- the optimizer of the
 * reference's training configs, task.py:187-199) for ALL parameters in one launch.  grads / exp_avg / exp_avg_sq are flat buffers of one
 * layout; chunk c updates param[poffset .. poffset + n) from the flat range [offset, offset + n), n <= SPK_ADAMW_CHUNK (one workgroup per
 * chunk; the table lives on the device).  step [1] is the device-side step count (float, starts at 0; the launch reads it, uses count + 1
 * and stores it); ticket [1] must be zero before the first call. */
#define SPK_ADAMW_CHUNK 2048
typedef struct {
  void* param;       /* base of the parameter tensor (float32) */
  int64_t poffset;   /* first element of the chunk inside the parameter */
  int64_t offset;    /* first element of the chunk inside the flat buffers */
  int64_t n;
} spk_adamw_chunk_t;
int spk_adamw_f32(const spk_adamw_chunk_t* chunks, int64_t n_chunks, const float* grads, float* exp_avg, float* exp_avg_sq, float* step,
                  uint32_t* ticket, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);
/* The same launch with the learning rate read from device memory (lr_dev [1]): a step captured in a HIP graph follows a schedule
 * (the reference attaches torch lr schedulers to its optimizer, task.py:253-275) by a host-to-device copy between replays. */
int spk_adamw_devlr_f32(const spk_adamw_chunk_t* chunks, int64_t n_chunks, const float* grads, float* exp_avg, float* exp_avg_sq, float* step,
                        uint32_t* ticket, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------ force-matching gradients by forward-over-reverse
 * Replaces what the reference obtains from autograd with create_graph = True: Forces (atomistic/response.py:59-68) builds the
 * graph of -dE/dR, the task differentiates loss(E, F) through it (task.py:166-185) -- reverse over reverse, several hundred
 * framework nodes per step.  For a loss L(E, F) with gE = dL/dE [n_mol] and gF = dL/dF [N,3] held fixed,
 *     dL/dtheta = d/dtheta [ sum_m gE_m E_m + D_t E_tot ],   t = -gF   (directional derivative of the total energy along t),
 * i.e. one dual-number forward pass along t and ONE reverse pass (oracle/fm_oracle.py, pinned against autograd; spk_fm_engine.h):
 *   spk_*_fm_forward_f32    values (kept in `workspace`) + reverse w.r.t. the positions -> E [n_mol], F [N,3] (F may be NULL)
 *   spk_*_fm_backward_f32   tangents + reverse of the dual graph -> `grads`, ONE flat fp32 buffer:
 *        SchNet: per interaction the nine tensors of spk_schnet_layer_t in that order | head w1, b1, w2, b2 | embedding [n_types, F]
 *        PaiNN:  per interaction the nine tensors of spk_painn_layer_t in that order (filt_* excluded) | filter_net.weight, .bias
 *                (all rows) | head w1, b1, w2, b2 | embedding
 *     (every entry is written; a weight shared by several interactions gets one slot per interaction -- the caller sums).
 * The model covers PairwiseDistances -> SchNet / PaiNN -> Atomwise (default two-layer head, summed per molecule) -> Forces without
 * stress; any F, n_filters, n_rbf, Gaussian / Bessel basis; idx_i ASCENDING (err |= 1 otherwise), idx_j in any order, the list need
 * not be symmetric (the transposed sums run over a by-neighbour CSR built on the device per call).  Only raw state_dict weights are
 * read (no transposed / packed copies: the weights change every step).  No host synchronisation: both calls can be captured in a
 * HIP graph.  The SAME workspace (and batch) must be passed to the backward call; spk_*_fm_workspace_bytes sizes it (n_types = rows of the
 * embedding table of the batch description: the workspace holds the one-hot rows of Z, the operand of the table's gradient). */
typedef struct {
  int64_t n_atoms, n_edges, n_mol;
  const int64_t* Z;        /* [N] atomic numbers (rows of `embedding`; out of range -> zero row) */
  const int64_t* idx_i;    /* [E] ascending */
  const int64_t* idx_j;    /* [E] */
  const int64_t* idx_m;    /* [N] ascending molecule index */
  const float* R;          /* [N,3] */
  const float* offsets;    /* [E,3] or NULL */
  const float* embedding;  /* [n_types, F] nuclear embedding table (representation/schnet.py:126-128) */
  int32_t n_types;
  int32_t reserved;
} spk_fm_batch_t;
int64_t spk_schnet_fm_workspace_bytes(const spk_schnet_t* m, const spk_head_t* head, const spk_radial_t* rb, int64_t n_atoms, int64_t n_edges, int64_t n_mol,
                                      int32_t n_types);
int64_t spk_schnet_fm_grad_floats(const spk_schnet_t* m, const spk_head_t* head, const spk_radial_t* rb, int32_t n_types);
int spk_schnet_fm_forward_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_radial_t* rb, const spk_fm_batch_t* batch, void* workspace,
                              float* E, float* F, int32_t* err, void* stream);
int spk_schnet_fm_backward_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_radial_t* rb, const spk_fm_batch_t* batch, void* workspace,
                               const float* gE, const float* gF, float* grads, void* stream);
int64_t spk_painn_fm_workspace_bytes(const spk_painn_t* m, const spk_head_t* head, const spk_radial_t* rb, int64_t n_atoms, int64_t n_edges, int64_t n_mol,
                                     int32_t n_types);
int64_t spk_painn_fm_grad_floats(const spk_painn_t* m, const spk_head_t* head, const spk_radial_t* rb, int32_t n_types);
int spk_painn_fm_forward_f32(const spk_painn_t* m, const spk_head_t* head, const spk_radial_t* rb, const spk_fm_batch_t* batch, void* workspace,
                             float* E, float* F, int32_t* err, void* stream);
int spk_painn_fm_backward_f32(const spk_painn_t* m, const spk_head_t* head, const spk_radial_t* rb, const spk_fm_batch_t* batch, void* workspace,
                              const float* gE, const float* gF, float* grads, void* stream);
/* EXPERIMENT (round 5, default off): row chains of the force-matching engine (csrc/spk_fm_chain.h) -- consecutive atom-local launches of a
 * pass recorded as stages of ONE launch (one workgroup per four atoms, fp32 products on v_mfma_f32_4x4x1).  Same results, 137 -> 62
 * launches per PaiNN step -- and slower (0.67 -> 0.89 ms at 8 frames): a stage of the generic chain kernel has about the fixed cost of the launch
 * it replaces (profiles/r05_row_chains.md).  mode 1: record chains; 0 or -1: launch by launch (default).  SPK_FM_CHAIN gives
 * the initial value. */
void spk_fm_set_chain(int32_t mode);

/* Up to SPK_INDEX_JOBS_MAX index jobs in ONE launch: job k derives the CSR row pointers rowptr [rows + 1] of the ascending index idx [n] with
 * entries in [0, rows) (err[0] |= 1 if not ascending, |= 2 if out of range -- like spk_segment_rowptr_i32), or, with rowptr == NULL, only checks
 * that every entry lies in [0, rows) (like spk_index_range_check).  err may be NULL when no job is a pure range check.  Device only. */
#define SPK_INDEX_JOBS_MAX 8
typedef struct {
  const int64_t* idx;
  int64_t n;
  int64_t rows;
  int32_t* rowptr;
} spk_index_job_t;
int spk_index_jobs(const spk_index_job_t* jobs, int32_t n_jobs, int32_t* err, void* stream);

/* By-neighbour CSR of a pair list on the device (the transpose permutation of SURVEY.md section 7 step 4): perm [E] = the pairs
 * ordered by idx_j (stable: ascending pair index inside a column), colptr [N + 2] (column N collects out-of-range neighbours).
 * tmp: spk_transpose_plan_bytes(E, N) bytes.  No host synchronisation. */
/* spk_transposed_build: the arrays of spk_transposed_t from the plan above (t_idx_i, t_idx_j [E] int64; rowptr [N + 2]; perm [E]);
 * tmp as for spk_transpose_plan.  No host synchronisation. */
int spk_transposed_build(const int64_t* idx_i, const int64_t* idx_j, int64_t n_edges, int64_t n_atoms, int64_t* t_idx_i, int64_t* t_idx_j,
                         int32_t* rowptr, int32_t* perm, void* tmp, void* stream);
int64_t spk_transpose_plan_bytes(int64_t n_edges, int64_t n_atoms);
int spk_transpose_plan(const int64_t* idx_j, int64_t n_edges, int64_t n_atoms, int32_t* colptr, int32_t* perm, void* tmp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPK_HIP_H */
