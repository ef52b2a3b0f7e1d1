// Molecule-resident PaiNN kernels (spk_painn_mol.hip): what the general driver (spk_painn.hip) calls.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spk_common.h"
#include "spk_pack.h"

// The standard potential in the two launches (spk_painn_potential_forces_f32; the POT instances of the kernels): pair vectors
// from the positions, rows of the embedding table, the default energy head  y = w2 . act(W1 q + b1) + b2  (atomistic/atomwise.py:69-88)
// on the atom tile that is still in LDS; the backward starts from dE/dE = 1 and ends at the forces.
struct PmHeadDev {
  const float* w1;        // outnet.0.weight [H, F], H = 64
  const float* w1t;       // its transpose   [F, H]   (backward)
  const float* b1;        // [H]
  const float* w2;        // outnet.1.weight [H]
  const float* b2;        // [1] or null
  int H, act;
  const int64_t* idx_m;   // [N]
  float* E;               // [n_mol]: one atomic per (group, molecule) -- cleared by the caller -- unless direct_store
  float* pre_h;           // [N, H] pre-activation of the hidden layer (forward -> backward)
  int direct_store;       // every molecule lies inside one group: plain stores, nothing to clear
};

// Shapes / lists the molecule-resident kernels cover (everything else runs the general driver)
bool spk_painn_mol_eligible(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb);
bool spk_painn_mol_bwd_eligible(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb);
int spk_painn_mol_forward(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab, const float* q0,
                          const float* r_ij, float* q_out, float* mu_out, float* saved, hipStream_t stream);
int spk_painn_mol_backward(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab, const float* gq_out,
                           const float* gmu_out, const float* r_ij, const float* saved, float* gc_scratch, float* gr, float* gq0, hipStream_t stream);
int spk_painn_mol_forward_ex(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab, const float* q0,
                             const float* r_ij, const float* R, const float* offsets, const float* emb, const int64_t* Z, int n_types,
                             const PmHeadDev* head, float* rij_out, float* gq_head_out, float* q_out, float* mu_out, float* saved, hipStream_t stream);
int spk_painn_mol_backward_ex(const spk_painn_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab, const float* gq_out,
                              const float* gmu_out, const float* r_ij, const float* saved,
                              float* gc_scratch, float* gr, float* gq0, float* forces, hipStream_t stream);
