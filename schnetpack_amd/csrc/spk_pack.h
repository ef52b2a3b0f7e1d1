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
  int64_t off_fwd_s, off_bwd_s;   // split-precision images (spk_split.h; -1: contraction length not a multiple of 16)
};

// further images kept in the same buffer, found by the raw tensor they were made from (e.g. the LDS image of a filter-network weight)
struct SpkPackExtra { const float* raw; int64_t off, floats; };

struct SpkPackTable {
  std::vector<SpkPackEntry> e;
  std::vector<SpkPackExtra> x;
  const float* base = nullptr;
  int64_t total = 0;
  void add_extra(const float* raw, int64_t floats) { x.push_back(SpkPackExtra{raw, total, floats}); total += floats; }
  const float* extra_of(const float* raw) const {
    if (!base) return nullptr;
    for (const SpkPackExtra& q : x) if (q.raw == raw) return base + q.off;
    return nullptr;
  }
  void add(const float* raw, const float* rawT, int n_out, int k_in) {
    SpkPackEntry x;
    x.raw = raw; x.rawT = rawT; x.n_out = n_out; x.k_in = k_in;
    x.off_fwd = total; x.off_bwd = total + (int64_t)n_out * k_in;
    total += 2 * (int64_t)n_out * k_in;
    // the (high, low) fp16 images have the size and the chunk geometry of the fp32 ones (8 k = 1024 bytes per tile row block)
    x.off_fwd_s = (k_in % 16 == 0) ? total : -1;
    if (x.off_fwd_s >= 0) total += (int64_t)n_out * k_in;
    x.off_bwd_s = (n_out % 16 == 0) ? total : -1;
    if (x.off_bwd_s >= 0) total += (int64_t)n_out * k_in;
    e.push_back(x);
  }
};

int spk_pack_weight_internal(const float* w, int n_out, int k_in, int transposed, float* packed, hipStream_t stream);
int spk_pack_weight_split_internal(const float* w, int n_out, int k_in, int transposed, float* packed, hipStream_t stream);

static inline int spk_pack_all(const SpkPackTable& T, float* wpack, hipStream_t stream) {
  for (const SpkPackEntry& x : T.e) {
    int rc = spk_pack_weight_internal(x.raw, x.n_out, x.k_in, 0, wpack + x.off_fwd, stream);
    if (rc) return rc;
    rc = spk_pack_weight_internal(x.raw, x.n_out, x.k_in, 1, wpack + x.off_bwd, stream);
    if (rc) return rc;
    if (x.off_fwd_s >= 0) { rc = spk_pack_weight_split_internal(x.raw, x.n_out, x.k_in, 0, wpack + x.off_fwd_s, stream); if (rc) return rc; }
    if (x.off_bwd_s >= 0) { rc = spk_pack_weight_split_internal(x.raw, x.n_out, x.k_in, 1, wpack + x.off_bwd_s, stream); if (rc) return rc; }
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

// the split-precision image of the same layer (NULL: none, or the split path is switched off -- spk_set_split)
static inline const float* spk_packed_split_of(const SpkPackTable& T, const float* raw, int transposed) {
  if (!T.base || spk_get_variant() == SPK_VARIANT_SIMPLE || !spk_get_split()) return nullptr;
  for (const SpkPackEntry& x : T.e)
    if (x.raw == raw) {
      const int64_t off = transposed ? x.off_bwd_s : x.off_fwd_s;
      return off >= 0 ? T.base + off : nullptr;
    }
  return nullptr;
}

// (spk_chain.hip) split images of the chain that is launched next on this host thread: the launcher takes the split-precision kernel when every layer has one
void spk_note_split_images(const float* const* packed, const float* const* split, int n);

static inline void spk_apply_pack(spk_chain_t& c, const SpkPackTable& T) {
  if (!T.base || spk_get_variant() == SPK_VARIANT_SIMPLE) return;
  const float* repl[3] = {nullptr, nullptr, nullptr};
  const float* repl_s[3] = {nullptr, nullptr, nullptr};
  for (int l = 0; l < c.n_layers; ++l) {
    const spk_chain_layer_t& L = c.layers[l];
    for (const SpkPackEntry& x : T.e) {
      if (L.trans == 0 && L.w == x.raw && L.k == x.k_in && L.n_out == x.n_out) { repl[l] = T.base + x.off_fwd; repl_s[l] = x.off_fwd_s >= 0 ? T.base + x.off_fwd_s : nullptr; }
      else if (L.trans == 1 && x.rawT && L.w == x.rawT && L.k == x.k_in && L.n_out == x.n_out) { repl[l] = T.base + x.off_fwd; repl_s[l] = x.off_fwd_s >= 0 ? T.base + x.off_fwd_s : nullptr; }
      else if (L.trans == 1 && L.w == x.raw && L.k == x.n_out && L.n_out == x.k_in) { repl[l] = T.base + x.off_bwd; repl_s[l] = x.off_bwd_s >= 0 ? T.base + x.off_bwd_s : nullptr; }
      if (repl[l]) break;
    }
    if (!repl[l]) return;
  }
  bool all_split = true;
  for (int l = 0; l < c.n_layers; ++l) { c.layers[l].w = repl[l]; c.layers[l].trans = 2; if (!repl_s[l]) all_split = false; }
  if (all_split) spk_note_split_images(repl, repl_s, c.n_layers);
}
