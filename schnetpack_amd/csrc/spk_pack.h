// Packed weight images of a whole model (see spk_pack_weight_f32 in spk_chain.hip): the drivers of the
// fused representations keep, next to every Linear weight W [n_out, k_in], the packed image of the
// forward layer (A = W) and of the input-gradient layer (A = W^T) in ONE caller-owned buffer `wpack`,
// laid out in the order of the table built here.  spk_apply_pack() rewrites the layers of a chain to the
// packed images when every layer has one.
#pragma once
#include <vector>
#include "spk_common.h"

struct SpkPackEntry {
  const float* raw;    // W [n_out, k_in]
  const float* rawT;   // optional transposed copy [k_in, n_out] handed in by the caller
  int n_out, k_in;
  int64_t off_fwd, off_bwd;
};

struct SpkPackTable {
  std::vector<SpkPackEntry> e;
  const float* base = nullptr;
  int64_t total = 0;
  void add(const float* raw, const float* rawT, int n_out, int k_in) {
    SpkPackEntry x;
    x.raw = raw; x.rawT = rawT; x.n_out = n_out; x.k_in = k_in;
    x.off_fwd = total; x.off_bwd = total + (int64_t)n_out * k_in;
    total += 2 * (int64_t)n_out * k_in;
    e.push_back(x);
  }
};

int spk_pack_weight_internal(const float* w, int n_out, int k_in, int transposed, float* packed, hipStream_t stream);

static inline int spk_pack_all(const SpkPackTable& T, float* wpack, hipStream_t stream) {
  for (const SpkPackEntry& x : T.e) {
    int rc = spk_pack_weight_internal(x.raw, x.n_out, x.k_in, 0, wpack + x.off_fwd, stream);
    if (rc) return rc;
    rc = spk_pack_weight_internal(x.raw, x.n_out, x.k_in, 1, wpack + x.off_bwd, stream);
    if (rc) return rc;
  }
  return SPK_OK;
}

// packed image of the forward (transposed == 0) or input-gradient (== 1) layer of the weight `raw`, or NULL
static inline const float* spk_packed_of(const SpkPackTable& T, const float* raw, int transposed) {
  if (!T.base || spk_get_variant() == SPK_VARIANT_SIMPLE) return nullptr;
  for (const SpkPackEntry& x : T.e)
    if (x.raw == raw) return T.base + (transposed ? x.off_bwd : x.off_fwd);
  return nullptr;
}

static inline void spk_apply_pack(spk_chain_t& c, const SpkPackTable& T) {
  if (!T.base || spk_get_variant() == SPK_VARIANT_SIMPLE) return;
  const float* repl[3] = {nullptr, nullptr, nullptr};
  for (int l = 0; l < c.n_layers; ++l) {
    const spk_chain_layer_t& L = c.layers[l];
    for (const SpkPackEntry& x : T.e) {
      if (L.trans == 0 && L.w == x.raw && L.k == x.k_in && L.n_out == x.n_out) repl[l] = T.base + x.off_fwd;
      else if (L.trans == 1 && x.rawT && L.w == x.rawT && L.k == x.k_in && L.n_out == x.n_out) repl[l] = T.base + x.off_fwd;
      else if (L.trans == 1 && L.w == x.raw && L.k == x.n_out && L.n_out == x.k_in) repl[l] = T.base + x.off_bwd;
      if (repl[l]) break;
    }
    if (!repl[l]) return;
  }
  for (int l = 0; l < c.n_layers; ++l) { c.layers[l].w = repl[l]; c.layers[l].trans = 2; }
}
