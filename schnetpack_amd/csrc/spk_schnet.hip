// Whole-representation drivers for SchNet (representation/schnet.py:147-173): every kernel of one
// forward (or one first-order backward) is enqueued on the caller's stream from a single C call.
#include "spk_common.h"

int spk_dense_internal(const float* in, const float* pre_in, const float* w, const float* b,
                       const float* res, float* out, float* pre_out, int64_t M, int KC, int NW,
                       int act, bool trans, int pro, hipStream_t stream);
int spk_cfconv_fwd_internal(const spk_graph_t* g, const spk_radial_t* rb, const float* h,
                            const float* r_ij, const float* w1, const float* b1, const float* w2,
                            const float* b2, int nf, float* y, hipStream_t stream);
int spk_cfconv_bwd_internal(const spk_graph_t* g, const spk_radial_t* rb, const float* h,
                            const float* gy, const float* r_ij, const float* w1, const float* b1,
                            const float* w2, const float* b2, int nf, float* gh, float* gr,
                            hipStream_t stream);

static int check_model(const spk_schnet_t* m, const char* who) {
  SPK_CHECK_ARG(m != nullptr && m->layers != nullptr, "%s: null model", who);
  SPK_CHECK_ARG(m->n_atom_basis > 0 && m->n_filters > 0 && m->n_interactions >= 0, "%s: bad model sizes", who);
  return SPK_OK;
}

extern "C" int64_t spk_schnet_saved_floats(const spk_schnet_t* m, int64_t n_atoms) {
  if (!m) return 0;
  return (int64_t)m->n_interactions * n_atoms * (m->n_filters + m->n_atom_basis);
}

extern "C" int64_t spk_schnet_scratch_floats(const spk_schnet_t* m, int64_t n_atoms) {
  if (!m) return 0;
  return n_atoms * (2 * (int64_t)m->n_filters + 2 * (int64_t)m->n_atom_basis);
}

#define SPK_TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

extern "C" int spk_schnet_forward_f32(const spk_schnet_t* m, const spk_graph_t* g,
                                      const spk_radial_t* rb, const float* x0, const float* r_ij,
                                      float* x_out, float* saved, float* scratch, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char* who = "spk_schnet_forward_f32";
  SPK_TRY(check_model(m, who));
  SPK_CHECK_ARG(g != nullptr && rb != nullptr, "%s: null graph/radial", who);
  const int64_t N = g->n_atoms;
  const int F = m->n_atom_basis, NF = m->n_filters, L = m->n_interactions;
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(x0 && x_out && (L == 0 || (saved && scratch)), "%s: null buffer", who);
  if (L == 0) {
    SPK_HIP_TRY(hipMemcpyAsync(x_out, x0, (size_t)N * F * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return SPK_OK;
  }
  float* y = scratch;                       // [N, NF]
  float* t = scratch + N * (int64_t)NF;     // [N, F]
  for (int l = 0; l < L; ++l) {
    const spk_schnet_layer_t& P = m->layers[l];
    float* h = saved + (int64_t)l * N * (NF + F);
    float* pre3 = h + N * (int64_t)NF;
    const float* xin = (l == 0) ? x0 : x_out;
    SPK_TRY(spk_dense_internal(xin, nullptr, P.in2f_w, nullptr, nullptr, h, nullptr, N, F, NF, SPK_ACT_NONE, false, SPK_ACT_NONE, stream));
    SPK_TRY(spk_cfconv_fwd_internal(g, rb, h, r_ij, P.fn_w1, P.fn_b1, P.fn_w2, P.fn_b2, NF, y, stream));
    SPK_TRY(spk_dense_internal(y, nullptr, P.f2out_w1, P.f2out_b1, nullptr, t, pre3, N, NF, F, SPK_ACT_SSP, false, SPK_ACT_NONE, stream));
    SPK_TRY(spk_dense_internal(t, nullptr, P.f2out_w2, P.f2out_b2, xin, x_out, nullptr, N, F, F, SPK_ACT_NONE, false, SPK_ACT_NONE, stream));
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
  SPK_CHECK_ARG(g != nullptr && rb != nullptr, "%s: null graph/radial", who);
  const int64_t N = g->n_atoms, E = g->n_edges;
  const int F = m->n_atom_basis, NF = m->n_filters, L = m->n_interactions;
  if (E > 0) {
    SPK_CHECK_ARG(gr != nullptr, "%s: null gr", who);
    SPK_HIP_TRY(hipMemsetAsync(gr, 0, (size_t)E * 3 * sizeof(float), stream));
  }
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(gx_out && (L == 0 || (saved && scratch)), "%s: null buffer", who);
  if (L == 0) {
    if (gx0) SPK_HIP_TRY(hipMemcpyAsync(gx0, gx_out, (size_t)N * F * sizeof(float), hipMemcpyDeviceToDevice, stream));
    return SPK_OK;
  }
  float* gt = scratch;                              // [N, F]
  float* gy = gt + N * (int64_t)F;                  // [N, NF]
  float* gh = gy + N * (int64_t)NF;                 // [N, NF]
  float* gxb = gh + N * (int64_t)NF;                // [N, F]
  const float* gx = gx_out;
  for (int l = L - 1; l >= 0; --l) {
    const spk_schnet_layer_t& P = m->layers[l];
    const float* h = saved + (int64_t)l * N * (NF + F);
    const float* pre3 = h + N * (int64_t)NF;
    // f2out.1: x_new = x + t W4^T + b4
    SPK_TRY(spk_dense_internal(gx, nullptr, P.f2out_w2, nullptr, nullptr, gt, nullptr, N, F, F, SPK_ACT_NONE, true, SPK_ACT_NONE, stream));
    // f2out.0: t = ssp(y W3^T + b3)
    SPK_TRY(spk_dense_internal(gt, pre3, P.f2out_w1, nullptr, nullptr, gy, nullptr, N, F, NF, SPK_ACT_NONE, true, SPK_ACT_SSP, stream));
    SPK_TRY(spk_cfconv_bwd_internal(g, rb, h, gy, r_ij, P.fn_w1, P.fn_b1, P.fn_w2, P.fn_b2, NF, gh, gr, stream));
    // in2f: h = x W_in^T ; residual path adds gx
    float* out = (l == 0 && gx0) ? gx0 : gxb;
    SPK_TRY(spk_dense_internal(gh, nullptr, P.in2f_w, nullptr, gx, out, nullptr, N, NF, F, SPK_ACT_NONE, true, SPK_ACT_NONE, stream));
    gx = out;
  }
  return SPK_OK;
}
