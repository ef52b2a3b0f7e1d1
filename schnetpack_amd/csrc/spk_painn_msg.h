// Argument block of the PaiNN message kernels (shared by the row kernels in spk_painn.hip and the MFMA tile
// kernels in spk_painn_tile.hip).
#pragma once
#include "spk_common.h"

struct MsgArgs {
  const float* c;       // [N, 3F] context-net output
  const float* q;       // [N, F]
  const float* mu;      // [N, 3, F]
  const float* gq_out;  // bwd [N, F]
  const float* gmu_out; // bwd [N, 3, F]
  const float* rij;     // [E, 3]
  const int64_t* idx_i;
  const int64_t* idx_j;
  const int32_t* rowptr;
  const float* wf;      // [3F, n_rbf] rows of this layer
  const float* bf;      // [3F]
  float* q_out;         // fwd [N, F]
  float* mu_out;        // fwd [N, 3, F]
  float* gc;            // bwd [N, 3F]
  float* gmu;           // bwd [N, 3, F]
  float* gr;            // bwd [E, 3] accumulated
  int64_t E, N;
  int F;
  int skin_list;      // the list holds a sizeable share of pairs beyond the cutoff (spk_graph_t.filter_pairs): the row kernel drops them before
                      // their rows are fetched, the tile kernel would pay the filter GEMM for them
  int mu_zero;        // mu is known to be all zeros (first interaction, painn.py:246): its rows are not gathered
  int xcd_map;        // row kernels: the workgroups of an XCD walk a contiguous eighth of the atoms (set by the launcher)
  int geom_only;      // bwd: only gr is wanted (first interaction of an eval-mode backward): gc / gmu are not formed
  RadialDev rb;
  // EXPERIMENT (spk_tabfilter.hip, opt-in): the raw filter phi(d) W_f^T + b_f of this interaction from a cubic-Hermite table
  // [n_knots][3F][2] = (value, slope * step) instead of the n_rbf FMAs per channel; null = off
  const float* tab;
  int tab_knots;
  float tab_inv_step;
  // block plan of the list (spk_painn_blk.hip; null: none) and whether its per-call edge tables are current for `rij`
  const spk_blocks_t* blocks;
  int blocks_prepared;
};

// MFMA tile kernel of the forward message (spk_painn_tile.hip): true if it should run for this shape / list
bool spk_painn_msg_tile_ok(const MsgArgs& a);
int spk_painn_msg_tile_fwd(const MsgArgs& a, hipStream_t stream);
bool spk_painn_msg_tile_bwd_ok(const MsgArgs& a);
int spk_painn_msg_tile_bwd(const MsgArgs& a, hipStream_t stream);
// row-tile backward (spk_painn_tile.hip): a wavefront per CSR row, filter and slope from the split-precision GEMM of 32-edge chunks
bool spk_painn_msg_rowtile_bwd_ok(const MsgArgs& a);
int spk_painn_msg_rowtile_bwd(const MsgArgs& a, hipStream_t stream);
bool spk_painn_msg_rowtile_fwd_ok(const MsgArgs& a);
int spk_painn_msg_rowtile_fwd(const MsgArgs& a, hipStream_t stream);
