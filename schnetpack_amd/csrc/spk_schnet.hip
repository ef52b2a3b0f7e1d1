// Whole-representation drivers for SchNet (representation/schnet.py:147-173): every kernel of one
// forward (or one first-order backward) is enqueued on the caller's stream from a single C call.
// Per interaction: ONE fused edge kernel (cfconv) and ONE fused atom-wise chain kernel
// (f2out.0 -> f2out.1 + residual -> next in2f; resp. their transposes in the backward), which also
// clears the accumulation buffer of the next edge kernel.
#include "spk_common.h"

int spk_cfconv_fwd_internal(const spk_graph_t* g, const spk_radial_t* rb, const float* h,
                            const float* r_ij, const float* w1, const float* b1, const float* w2,
                            const float* b2, int nf, float* y, hipStream_t stream, bool pre_zeroed,
                            float* gsave);
int64_t spk_cfconv_gsave_floats(const spk_graph_t* g, const spk_radial_t* rb, int nf);
int spk_cfconv_bwd_internal(const spk_graph_t* g, const spk_radial_t* rb, const float* h,
                            const float* gy, const float* r_ij, const float* w1, const float* b1,
                            const float* w2, const float* b2, int nf, float* gh, float* gr,
                            hipStream_t stream, bool pre_zeroed, const float* gload, bool gr_assign, bool want_gh);

// EXPERIMENT (spk_tabfilter.hip): a model whose first interaction has a registered filter table runs the general driver with the
// table-driven cfconv kernels (the molecule-resident launches evaluate the filter network themselves)
bool spk_filter_table_lookup(const float* key, const float** table, int* n_knots, float* d_max);
static bool schnet_tabulated(const spk_schnet_t* m) {
  const float* tab; int nk; float dmax;
  return m && m->layers && m->n_interactions > 0 && spk_filter_table_lookup(m->layers[0].fn_w2, &tab, &nk, &dmax);
}

static int check_model(const spk_schnet_t* m, const char* who) {
  SPK_CHECK_ARG(m != nullptr && m->layers != nullptr, "%s: null model", who);
  SPK_CHECK_ARG(m->n_atom_basis > 0 && m->n_filters > 0 && m->n_interactions >= 0, "%s: bad model sizes", who);
  return SPK_OK;
}

// saved: per interaction h [N,nf] | pre3 [N,F], followed (optionally) by the raw filter outputs
// g_e of every interaction ([n_half_padded, nf] each) when `n_edges_hint` > 0 was given
extern "C" int64_t spk_schnet_saved_floats(const spk_schnet_t* m, int64_t n_atoms) {
  if (!m) return 0;
  return (int64_t)m->n_interactions * n_atoms * (m->n_filters + m->n_atom_basis);
}
int64_t spk_active_pairs_floats(int64_t n_half);
int spk_active_pairs_internal(const spk_graph_t* g, const float* r_ij, float cutoff, float* ws, hipStream_t stream,
                              const int32_t** half_out, const int32_t** count_out);
// the per-call compacted pair list lives behind the saved filters (only when the pair kernels run)
static bool schnet_filter_on(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb) {
  // (the experimental group-local kernels address `half` through per-group offsets: no compaction there)
  return g->filter_pairs && g->n_half > 0 && spk_get_variant() != SPK_VARIANT_MFMA_MOL &&
         spk_cfconv_gsave_floats(g, rb, m->n_filters) > 0;
}
extern "C" int64_t spk_schnet_saved_floats_graph(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb) {
  if (!m || !g || !rb) return 0;
  return spk_schnet_saved_floats(m, g->n_atoms) + (int64_t)m->n_interactions * spk_cfconv_gsave_floats(g, rb, m->n_filters) +
         (schnet_filter_on(m, g, rb) ? spk_active_pairs_floats(g->n_half) : 0);
}

// scratch: forward  y0 | y1 | t0 | t1            (2 nf + 2 max(F, nf))
//          backward gh0 | gh1 | gy | gxb | t0 | t1
extern "C" int64_t spk_schnet_scratch_floats(const spk_schnet_t* m, int64_t n_atoms) {
  if (!m) return 0;
  const int64_t mx = m->n_filters > m->n_atom_basis ? m->n_filters : m->n_atom_basis;
  return n_atoms * (3 * (int64_t)m->n_filters + (int64_t)m->n_atom_basis + 2 * mx);
}

#define SPK_TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

#include "spk_pack.h"
// molecule-resident path (spk_schnet_mol.hip): block-diagonal lists with <= 32 atoms per group
bool spk_schnet_mol_eligible(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb);
int spk_schnet_mol_forward(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                           const float* x0, const float* r_ij, float* x_out, float* saved, int64_t gsz, hipStream_t stream);
bool spk_schnet_mol_bwd_eligible(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb);
int spk_schnet_mol_backward(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                            const float* gx_out, const float* r_ij, const float* saved, int64_t gsz, float* gr, float* gx0,
                            hipStream_t stream);
struct MolHeadDev {      // (layout shared with spk_schnet_mol.hip)
  const float *w1, *w1t, *b1, *w2, *b2;
  int H, act;
  const int64_t* idx_m;
  float* E;
  float* pre_h;
  const float* gE;
  int direct_store, negate;
};
int spk_schnet_mol_forward_ex(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                              const float* x0, const float* r_ij, const float* R, const float* offsets, const MolHeadDev* head, float* x_out,
                              float* saved, int64_t gsz, hipStream_t stream, const float* emb = nullptr, const int64_t* Z = nullptr, int n_types = 0);
int spk_schnet_mol_backward_ex(const spk_schnet_t* m, const spk_graph_t* g, const spk_radial_t* rb, const SpkPackTable& ptab,
                               const float* gx_out, const float* r_ij, const float* R, const float* offsets, const MolHeadDev* head,
                               const float* saved, int64_t gsz, float* gr, float* gR, float* gx0, hipStream_t stream);
// order of the packed images in spk_schnet_t::wpack: per interaction in2f, f2out.0, f2out.1 (forward, transposed each)
static bool schnet_pack_shapes_ok(const spk_schnet_t* m) { return m->n_atom_basis % 128 == 0 && m->n_filters % 128 == 0 && m->n_atom_basis <= 384 && m->n_filters <= 384; }
static SpkPackTable schnet_pack_table(const spk_schnet_t* m) {
  SpkPackTable T;
  const int F = m->n_atom_basis, NF = m->n_filters;
  for (int l = 0; l < m->n_interactions; ++l) {
    const spk_schnet_layer_t& P = m->layers[l];
    T.add(P.in2f_w, P.in2f_wT, NF, F);
    T.add(P.f2out_w1, P.f2out_w1T, F, NF);
    T.add(P.f2out_w2, P.f2out_w2T, F, F);
  }
  // the split-precision LDS image of filter_network.1.weight of every interaction (spk_schnet_mol.hip: staged by a plain copy)
  if (NF == 128)
    for (int l = 0; l < m->n_interactions; ++l) T.add_extra(m->layers[l].fn_w2, 128 * 128);
  T.base = m->wpack;
  return T;
}
int spk_schnet_mol_pack_w2(const float* w2, float* image, hipStream_t stream);
extern "C" int64_t spk_schnet_packed_floats(const spk_schnet_t* m) {
  if (!m || !m->layers || m->n_interactions <= 0 || !schnet_pack_shapes_ok(m)) return 0;
  return schnet_pack_table(m).total;
}
extern "C" int spk_schnet_pack_weights_f32(const spk_schnet_t* m, float* wpack, void* stream) {
  SPK_CHECK_ARG(m && wpack && spk_schnet_packed_floats(m) > 0, "spk_schnet_pack_weights_f32: model shapes have no packed form (see spk_schnet_packed_floats)");
  const SpkPackTable T = schnet_pack_table(m);
  int rc = spk_pack_all(T, wpack, (hipStream_t)stream);
  for (const SpkPackExtra& q : T.x) {
    if (rc) break;
    rc = spk_schnet_mol_pack_w2(q.raw, wpack + q.off, (hipStream_t)stream);
  }
  return rc;
}

// forward layer from either the [out,in] weight or its transposed copy (coalesced reads)
static spk_chain_layer_t mk_fwd(const float* w, const float* wT, const float* b, const float* res, float* out,
                                float* pre_out, int k, int n_out, int act);

static spk_chain_layer_t mk_layer(const float* w, const float* b, const float* res, float* out, float* pre_out,
                                  const float* post_pre, int k, int n_out, int act, int trans, int post_act) {
  spk_chain_layer_t L;
  L.w = w; L.b = b; L.res = res; L.out = out; L.pre_out = pre_out; L.post_pre = post_pre;
  L.k = k; L.n_out = n_out; L.act = act; L.trans = trans; L.post_act = post_act;
  return L;
}

static spk_chain_layer_t mk_fwd(const float* w, const float* wT, const float* b, const float* res, float* out,
                                float* pre_out, int k, int n_out, int act) {
  return wT ? mk_layer(wT, b, res, out, pre_out, nullptr, k, n_out, act, 1, 0)
            : mk_layer(w, b, res, out, pre_out, nullptr, k, n_out, act, 0, 0);
}

extern "C" int spk_schnet_forward_f32(const spk_schnet_t* m, const spk_graph_t* g,
                                      const spk_radial_t* rb, const float* x0, const float* r_ij,
                                      float* x_out, float* saved, float* scratch, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_schnet_forward_f32";
  SPK_TRY(check_model(m, who));
  const SpkPackTable ptab = (m->wpack && schnet_pack_shapes_ok(m)) ? schnet_pack_table(m) : SpkPackTable();
  SPK_CHECK_ARG(g != nullptr && rb != nullptr, "%s: null graph/radial", who);
  const int64_t N = g->n_atoms;
  const int F = m->n_atom_basis, NF = m->n_filters, L = m->n_interactions;
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(x0 && x_out && (L == 0 || (saved && scratch)), "%s: null buffer", who);
  if (L == 0) {
    SPK_HIP_TRY(hipMemcpyAsync(x_out, x0, (size_t)N * F * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return SPK_OK;
  }
  const int64_t mx = NF > F ? NF : F;
  float* ybuf[2] = {scratch, scratch + N * (int64_t)NF};
  float* tmp0 = scratch + 2 * N * (int64_t)NF;
  float* tmp1 = tmp0 + N * mx;
  auto hbuf = [&](int l) { return saved + (int64_t)l * N * (NF + F); };
  const int64_t gsz = (m->reserved & 1) ? spk_cfconv_gsave_floats(g, rb, NF) : 0;  // bit 0: saved has filter space
  float* gbase = saved + (int64_t)L * N * (NF + F);
  // batches of small molecules with the filters saved for the backward: the whole forward is one molecule-resident launch
  if (gsz > 0 && ptab.base && spk_schnet_mol_eligible(m, g, rb) && !schnet_tabulated(m))
    return spk_schnet_mol_forward(m, g, rb, ptab, x0, r_ij, x_out, saved, gsz, stream);
  // skin lists: compact the pair list of THIS call (pairs inside the cutoff, order kept) behind the saved filters
  spk_graph_t gact = *g;
  if (gsz > 0 && schnet_filter_on(m, g, rb)) {
    const int32_t *ah = nullptr, *ac = nullptr;
    SPK_TRY(spk_active_pairs_internal(g, r_ij, rb->cutoff, gbase + (int64_t)L * gsz, stream, &ah, &ac));
    gact.half = ah; gact.n_half_dev = ac;
  }
  g = &gact;
  // prelude: h_0 = in2f_0(x0); clear y of the first edge kernel
  {
    spk_chain_t c = {};
    c.n_layers = 1; c.m = N; c.in = x0; c.zero_ptr = ybuf[0]; c.zero_count = N * (int64_t)NF;
    c.tmp[0] = tmp0; c.tmp[1] = tmp1;
    c.layers[0] = mk_fwd(m->layers[0].in2f_w, m->layers[0].in2f_wT, nullptr, nullptr, hbuf(0), nullptr, F, NF, SPK_ACT_NONE);
    spk_apply_pack(c, ptab);
    SPK_TRY(spk_dense_chain_f32(&c, stream));
  }
  for (int l = 0; l < L; ++l) {
    const spk_schnet_layer_t& P = m->layers[l];
    float* h = hbuf(l);
    float* pre3 = h + N * (int64_t)NF;
    float* y = ybuf[l & 1];
    const float* xin = (l == 0) ? x0 : x_out;
    SPK_TRY(spk_cfconv_fwd_internal(g, rb, h, r_ij, P.fn_w1, P.fn_b1, P.fn_w2, P.fn_b2, NF, y, stream, true,
                                    gsz > 0 ? gbase + l * gsz : nullptr));
    spk_chain_t c = {};
    c.m = N; c.in = y; c.tmp[0] = tmp0; c.tmp[1] = tmp1;
    c.layers[0] = mk_fwd(P.f2out_w1, P.f2out_w1T, P.f2out_b1, nullptr, nullptr, pre3, NF, F, SPK_ACT_SSP);
    c.layers[1] = mk_fwd(P.f2out_w2, P.f2out_w2T, P.f2out_b2, xin, x_out, nullptr, F, F, SPK_ACT_NONE);
    c.n_layers = 2;
    if (l + 1 < L) {
      c.layers[2] = mk_fwd(m->layers[l + 1].in2f_w, m->layers[l + 1].in2f_wT, nullptr, nullptr, hbuf(l + 1), nullptr, F, NF, SPK_ACT_NONE);
      c.n_layers = 3;
      c.zero_ptr = ybuf[(l + 1) & 1]; c.zero_count = N * (int64_t)NF;
    }
    spk_apply_pack(c, ptab);
    SPK_TRY(spk_dense_chain_f32(&c, stream));
  }
  return SPK_OK;
}

extern "C" int spk_schnet_backward_f32(const spk_schnet_t* m, const spk_graph_t* g,
                                       const spk_radial_t* rb, const float* gx_out,
                                       const float* r_ij, const float* saved, float* scratch,
                                       float* gr, float* gx0, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_schnet_backward_f32";
  SPK_TRY(check_model(m, who));
  const SpkPackTable ptab = (m->wpack && schnet_pack_shapes_ok(m)) ? schnet_pack_table(m) : SpkPackTable();
  SPK_CHECK_ARG(g != nullptr && rb != nullptr, "%s: null graph/radial", who);
  const int64_t N = g->n_atoms, E = g->n_edges;
  const int F = m->n_atom_basis, NF = m->n_filters, L = m->n_interactions;
  // batches of small molecules: the whole backward is one molecule-resident launch (it assigns every entry of gr)
  if (L > 0 && N > 0 && E > 0 && (m->reserved & 1) && ptab.base && gx_out && saved && gr && spk_schnet_mol_bwd_eligible(m, g, rb) && !schnet_tabulated(m)) {
    const int64_t gsz_m = spk_cfconv_gsave_floats(g, rb, NF);
    if (gsz_m > 0) return spk_schnet_mol_backward(m, g, rb, ptab, gx_out, r_ij, saved, gsz_m, gr, gx0, stream);
  }
  // with saved filters the pair kernel runs and writes every entry of gr exactly once per interaction: the
  // first interaction of the backward assigns, the others accumulate -- no clearing pass
  const bool gr_assign = L > 0 && (m->reserved & 1) && E > 0 && spk_cfconv_gsave_floats(g, rb, NF) > 0 &&
                         (!schnet_filter_on(m, g, rb) || (ptab.base && spk_schnet_mol_eligible(m, g, rb)));
  if (E > 0) {
    SPK_CHECK_ARG(gr != nullptr, "%s: null gr", who);
    if (!gr_assign) { int _zr = spk_zero_async(gr, (size_t)E * 3 * sizeof(float), stream); if (_zr) return _zr; }
  }
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(gx_out && (L == 0 || (saved && scratch)), "%s: null buffer", who);
  if (L == 0) {
    if (gx0) SPK_HIP_TRY(hipMemcpyAsync(gx0, gx_out, (size_t)N * F * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return SPK_OK;
  }
  const int64_t mx = NF > F ? NF : F;
  float* ghbuf[2] = {scratch, scratch + N * (int64_t)NF};
  float* gy = scratch + 2 * N * (int64_t)NF;
  float* gxb = gy + N * (int64_t)NF;
  float* tmp0 = gxb + N * (int64_t)F;
  float* tmp1 = tmp0 + N * mx;
  auto hbuf = [&](int l) { return saved + (int64_t)l * N * (NF + F); };
  auto pre3 = [&](int l) { return saved + (int64_t)l * N * (NF + F) + N * (int64_t)NF; };
  const int64_t gsz = (m->reserved & 1) ? spk_cfconv_gsave_floats(g, rb, NF) : 0;
  const float* gbase = saved + (int64_t)L * N * (NF + F);
  // the compacted pair list written by the forward of this call (the molecule-resident forward writes none: it keeps
  // the full pair list, and so does this backward then -- saved filters are addressed by pair position)
  const bool mol_fwd = gsz > 0 && ptab.base && spk_schnet_mol_eligible(m, g, rb) && !schnet_tabulated(m);
  spk_graph_t gact = *g;
  if (gsz > 0 && !mol_fwd && schnet_filter_on(m, g, rb)) {
    const int32_t* ah = (const int32_t*)(gbase + (int64_t)L * gsz);
    gact.half = ah; gact.n_half_dev = ah + g->n_half;
  }
  const bool filtered = gact.n_half_dev != nullptr;
  // prelude: gy_{L-1} = ((gx W4) * ssp'(pre3)) W3 ; clear gh of the first edge kernel
  {
    const spk_schnet_layer_t& P = m->layers[L - 1];
    spk_chain_t c = {};
    c.n_layers = 2; c.m = N; c.in = gx_out; c.zero_ptr = ghbuf[(L - 1) & 1]; c.zero_count = N * (int64_t)NF;
    c.tmp[0] = tmp0; c.tmp[1] = tmp1;
    c.layers[0] = mk_layer(P.f2out_w2, nullptr, nullptr, nullptr, nullptr, pre3(L - 1), F, F, SPK_ACT_NONE, 1, SPK_ACT_SSP);
    c.layers[1] = mk_layer(P.f2out_w1, nullptr, nullptr, gy, nullptr, nullptr, F, NF, SPK_ACT_NONE, 1, 0);
    spk_apply_pack(c, ptab);
    SPK_TRY(spk_dense_chain_f32(&c, stream));
  }
  const float* gx = gx_out;
  for (int l = L - 1; l >= 0; --l) {
    const spk_schnet_layer_t& P = m->layers[l];
    float* gh = ghbuf[l & 1];
    SPK_TRY(spk_cfconv_bwd_internal(filtered ? &gact : g, rb, hbuf(l), gy, r_ij, P.fn_w1, P.fn_b1, P.fn_w2, P.fn_b2, NF, gh, gr, stream, true,
                                    gsz > 0 ? gbase + l * gsz : nullptr, gr_assign && l == L - 1, !(l == 0 && !gx0)));
    if (l == 0 && !gx0) break;  // dL/dx0 not requested (eval path): nothing below feeds dL/dr_ij, and the edge kernel
                                // above was told not to form dL/dh (no transposed sum, no atomics)
    float* out = (l == 0) ? gx0 : gxb;
    spk_chain_t c = {};
    c.m = N; c.in = gh; c.tmp[0] = tmp0; c.tmp[1] = tmp1;
    // in2f: h = x W_in^T ; residual path adds gx
    c.layers[0] = mk_layer(P.in2f_w, nullptr, gx, out, nullptr, nullptr, NF, F, SPK_ACT_NONE, 1, 0);
    c.n_layers = 1;
    if (l > 0) {
      const spk_schnet_layer_t& Q = m->layers[l - 1];
      c.layers[0].post_pre = nullptr;
      c.layers[1] = mk_layer(Q.f2out_w2, nullptr, nullptr, nullptr, nullptr, pre3(l - 1), F, F, SPK_ACT_NONE, 1, SPK_ACT_SSP);
      c.layers[2] = mk_layer(Q.f2out_w1, nullptr, nullptr, gy, nullptr, nullptr, F, NF, SPK_ACT_NONE, 1, 0);
      c.n_layers = 3;
      c.zero_ptr = ghbuf[(l - 1) & 1]; c.zero_count = N * (int64_t)NF;
    }
    spk_apply_pack(c, ptab);
    SPK_TRY(spk_dense_chain_f32(&c, stream));
    gx = out;
  }
  return SPK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The standard potential  PairwiseDistances -> SchNet -> Atomwise(sum) -> Forces  on batches of small molecules: TWO launches
// (atomistic/distances.py:14-26, representation/schnet.py:147-173, atomistic/atomwise.py:69-88, atomistic/response.py:59-76).
static bool potential_ok(const spk_schnet_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb) {
  if (!m || !head || !g || !rb || !m->layers || m->n_interactions <= 0 || !(m->reserved & 1)) return false;
  if (schnet_tabulated(m)) return false;          // (experiment) registered filter tables: the stage-by-stage path runs the table kernels
  if (!m->wpack || !schnet_pack_shapes_ok(m)) return false;
  if (!head->w1 || !head->w1t || !head->b1 || !head->w2 || head->n_hidden < 32 || head->n_hidden > 128 || head->n_hidden % 32) return false;
  if (head->act != SPK_ACT_SSP && head->act != SPK_ACT_SILU) return false;
  if (getenv("SPK_NO_POTENTIAL")) return false;
  return spk_schnet_mol_eligible(m, g, rb) && spk_schnet_mol_bwd_eligible(m, g, rb) && spk_cfconv_gsave_floats(g, rb, m->n_filters) > 0;
}
extern "C" int spk_schnet_potential_supported(const spk_schnet_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb) {
  return potential_ok(m, head, g, rb) ? 1 : 0;
}

extern "C" int spk_schnet_potential_forward_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb,
                                                const float* x0, const float* R, const float* offsets, const int64_t* idx_m, int64_t n_mol,
                                                float* x_out, float* E, float* pre_h, float* saved, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_schnet_potential_forward_f32";
  SPK_TRY(check_model(m, who));
  SPK_CHECK_ARG(potential_ok(m, head, g, rb), "%s: model / list not covered by the fused potential (see spk_schnet_potential_supported)", who);
  SPK_CHECK_ARG(n_mol >= 0 && (n_mol == 0 || E), "%s: null energy buffer", who);
  if (n_mol > 0) { int zr = spk_zero_async(E, (size_t)n_mol * sizeof(float), stream); if (zr) return zr; }
  if (g->n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(x0 && R && idx_m && x_out && pre_h && saved, "%s: null buffer", who);
  MolHeadDev h;
  h.w1 = head->w1; h.w1t = head->w1t; h.b1 = head->b1; h.w2 = head->w2; h.b2 = head->b2; h.H = head->n_hidden; h.act = head->act;
  h.idx_m = idx_m; h.E = E; h.pre_h = pre_h; h.gE = nullptr; h.direct_store = 0; h.negate = 0;
  return spk_schnet_mol_forward_ex(m, g, rb, schnet_pack_table(m), x0, nullptr, R, offsets, &h, x_out, saved,
                                   spk_cfconv_gsave_floats(g, rb, m->n_filters), stream);
}

extern "C" int spk_schnet_potential_backward_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb,
                                                 const float* gE, const float* gx_out, const float* R, const float* offsets, const int64_t* idx_m,
                                                 const float* pre_h, const float* saved, float* gR, float* gx0, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_schnet_potential_backward_f32";
  SPK_TRY(check_model(m, who));
  SPK_CHECK_ARG(potential_ok(m, head, g, rb), "%s: model / list not covered by the fused potential (see spk_schnet_potential_supported)", who);
  if (g->n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(gE && R && idx_m && pre_h && saved && gR, "%s: null buffer", who);
  MolHeadDev h;
  h.w1 = head->w1; h.w1t = head->w1t; h.b1 = head->b1; h.w2 = head->w2; h.b2 = head->b2; h.H = head->n_hidden; h.act = head->act;
  h.idx_m = idx_m; h.E = nullptr; h.pre_h = const_cast<float*>(pre_h); h.gE = gE; h.direct_store = 0; h.negate = 0;
  return spk_schnet_mol_backward_ex(m, g, rb, schnet_pack_table(m), gx_out, nullptr, R, offsets, &h, saved,
                                    spk_cfconv_gsave_floats(g, rb, m->n_filters), nullptr, gR, gx0, stream);
}

// Energies and FORCES of the standard potential in the two launches, nothing else: x0 may be NULL (the rows of the nuclear
// embedding table `emb` [n_types, F] are looked up by Z inside the forward launch), dL/dE is 1 (forces of the summed energy), the
// backward writes -dE/dR.  all_inside != 0: the caller guarantees that every molecule's atoms lie inside ONE group of the plan and
// that every molecule has an atom -- the energies are then stored, not accumulated, and E needs no clearing launch.
extern "C" int spk_schnet_potential_forces_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_graph_t* g, const spk_radial_t* rb,
                                               const float* x0, const float* emb, const int64_t* Z, int32_t n_types, const float* R,
                                               const float* offsets, const int64_t* idx_m, int64_t n_mol, int32_t all_inside, float* x_out, float* E,
                                               float* F, float* pre_h, float* saved, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_schnet_potential_forces_f32";
  SPK_TRY(check_model(m, who));
  SPK_CHECK_ARG(potential_ok(m, head, g, rb), "%s: model / list not covered by the fused potential (see spk_schnet_potential_supported)", who);
  SPK_CHECK_ARG(n_mol >= 0 && (n_mol == 0 || E), "%s: null energy buffer", who);
  SPK_CHECK_ARG(x0 || (emb && Z && n_types > 0), "%s: neither features nor an embedding table", who);
  if (n_mol > 0 && !all_inside) { int zr = spk_zero_async(E, (size_t)n_mol * sizeof(float), stream); if (zr) return zr; }
  if (g->n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(R && idx_m && x_out && F && pre_h && saved, "%s: null buffer", who);
  MolHeadDev h;
  h.w1 = head->w1; h.w1t = head->w1t; h.b1 = head->b1; h.w2 = head->w2; h.b2 = head->b2; h.H = head->n_hidden; h.act = head->act;
  h.idx_m = idx_m; h.E = E; h.pre_h = pre_h; h.gE = nullptr; h.direct_store = all_inside ? 1 : 0; h.negate = 1;
  const SpkPackTable ptab = schnet_pack_table(m);
  const int64_t gsz = spk_cfconv_gsave_floats(g, rb, m->n_filters);
  SPK_TRY(spk_schnet_mol_forward_ex(m, g, rb, ptab, x0, nullptr, R, offsets, &h, x_out, saved, gsz, stream, x0 ? nullptr : emb, Z, n_types));
  return spk_schnet_mol_backward_ex(m, g, rb, ptab, nullptr, nullptr, R, offsets, &h, saved, gsz, nullptr, F, nullptr, stream);
}
