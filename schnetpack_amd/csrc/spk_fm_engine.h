// Force-matching gradient engine: host orchestration of the four passes (see spk_fm_kernels.h and oracle/fm_oracle.py)
//   forward  = A (values) + B (reverse w.r.t. the positions)                    -> energies, forces
//   backward = C (tangents along t = -dL/dF) + D (reverse of the dual graph)    -> the gradient of L w.r.t. every weight
// for a loss L(E, F) given gE = dL/dE, gF = dL/dF: what the reference gets from autograd with create_graph=True
// (atomistic/response.py:59-68, task.py:166-185) as a second-order graph of several hundred nodes is here ~100 launches of
// Dense / weight-gradient GEMMs on [value ; tangent]-stacked rows and of the element-wise / row kernels of spk_fm_kernels.h.
//
// The code is a template over a BACKEND that provides the Dense GEMMs, the transposed (weight-gradient) GEMM, CSR helpers and
// the kernel launcher: spk_fm.hip instantiates it on the device (T = float; spk_dense_f32 / spk_gemm_tn on the fp32 MFMA,
// rocPRIM for the by-neighbour CSR), tests/fm_emu instantiates it with serial loops in float64 / float32 on the build box
// (test infrastructure: the orchestration is identical, so a wrong buffer, stride or formula is caught without a GPU).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include <initializer_list>
#include "spk_fm_kernels.h"

template <class T>
struct FmSchnetLayer { const T *in2f_w, *fn_w1, *fn_b1, *fn_w2, *fn_b2, *f2out_w1, *f2out_b1, *f2out_w2, *f2out_b2; };
template <class T>
struct FmPainnLayer { const T *ctx_w1, *ctx_b1, *ctx_w2, *ctx_b2, *mix_w, *ictx_w1, *ictx_b1, *ictx_w2, *ictx_b2; };
template <class T>
struct FmHead { const T *w1, *b1, *w2, *b2; int n_hidden, act; };
template <class T>
struct FmBatch {
  int64_t N, E, M;
  const int64_t *Z, *ii, *jj, *idx_m;   // atomic numbers, pair indices (ii ascending), molecule index (ascending)
  const T *R, *off;                     // positions [N,3], pair offsets [E,3] or NULL
  const T* emb;                         // nuclear embedding table [n_types, F]
  int n_types;
};
template <class T>
struct FmSchnetModel { int F, nf, L; const FmSchnetLayer<T>* layers; };
template <class T>
struct FmPainnModel { int F, L, shared_filters; T eps; const FmPainnLayer<T>* layers; const T *filt_w, *filt_b; };   // filter_net.{weight, bias}, all rows

// ---------------------------------------------------------------------------------------------------------------- workspace arena
// One pass over the same carving code computes the size (base == NULL) or binds the pointers.
struct FmArena {
  char* base;
  size_t off;
  explicit FmArena(void* b) : base((char*)b), off(0) {}
  template <class U>
  U* take(int64_t n) {
    U* r = base ? (U*)(base + off) : nullptr;
    off += ((size_t)(n > 0 ? n : 1) * sizeof(U) + 255) & ~(size_t)255;
    return r;
  }
};

template <class T>
struct FmCommonWs {
  int32_t *rowptr, *rowptr_m, *colptr, *perm, *csrc, *e_act;
  void* plan_tmp;
  T *d, *u, *fc, *fc1, *phi2, *dt, *ut, *gd, *gu, *gr;
  T *preh2, *th2, *e_atom, *gt2, *gpre2, *gEa2;
  T* onehot;      // [N, n_types] one-hot rows of Z: operand of the embedding table's gradient
  T* gemm_ws;
  uint32_t* tickets;
  int64_t zero_words;
};

// flat gradient layouts (floats): SchNet  [per interaction: the nine tensors of spk_schnet_layer_t] | head w1 b1 w2 b2 | embedding
//                                 PaiNN   [per interaction: the nine tensors of spk_painn_layer_t] | filter_net w, b | head | embedding
static inline int64_t fm_schnet_layer_grad_floats(int F, int nf, int K) { return (int64_t)nf * F + (int64_t)nf * K + nf + (int64_t)nf * nf + nf + (int64_t)F * nf + F + (int64_t)F * F + F; }
static inline int64_t fm_painn_layer_grad_floats(int F) { return (int64_t)F * F + F + 3ll * F * F + 3 * F + 2ll * F * F + 2ll * F * F + F + 3ll * F * F + 3 * F; }
static inline int64_t fm_head_grad_floats(int F, int H) { return (int64_t)H * F + H + H + 1; }
static inline int64_t fm_schnet_grad_floats(int F, int nf, int L, int K, int H, int n_types) {
  return L * fm_schnet_layer_grad_floats(F, nf, K) + fm_head_grad_floats(F, H) + (int64_t)n_types * F;
}
static inline int64_t fm_painn_grad_floats(int F, int L, int K, int H, int n_types, int shared) {
  const int Lf = shared ? 1 : L;
  return L * fm_painn_layer_grad_floats(F) + 3ll * F * Lf * K + 3ll * F * Lf + fm_head_grad_floats(F, H) + (int64_t)n_types * F;
}

template <class T, class B>
struct FmEngine {
  B& be;
  explicit FmEngine(B& b) : be(b) {}

  // the atom-local element-wise kernels of PaiNN's mixing block through one descriptor (a stage of a row chain on the device, spk_fm_chain.h)
  void ew(int kind, int64_t N, int F, T eps, std::initializer_list<const T*> in, std::initializer_list<T*> out) {
    FmEwArgs<T> a;
    a.kind = kind; a.F = F; a.N = N; a.eps = eps;
    for (int q = 0; q < 8; ++q) a.in[q] = nullptr;
    for (int q = 0; q < 3; ++q) a.out[q] = nullptr;
    int q = 0;
    for (const T* p : in) a.in[q++] = p;
    q = 0;
    for (T* p : out) a.out[q++] = p;
    be.ew(a);
  }

  // ------------------------------------------------------------------------------------------------ shared pieces
  void carve_common(FmArena& a, FmCommonWs<T>& w, int64_t N, int64_t E, int64_t M, int K, int H, int n_types, bool painn, int64_t gemm_ws_floats) {
    // zeroed together by ONE launch in prepare(): [4096] ticket counters of the weight-gradient GEMM | [64] e_act | gd [E] | gu [3 E] (pass B accumulates into them)
    const size_t z0 = a.off;
    w.tickets = a.take<uint32_t>(4096 + 64);
    w.e_act = (int32_t*)(w.tickets ? w.tickets + 4096 : nullptr);
    w.gd = a.take<T>(E);
    w.gu = painn ? a.take<T>(3 * E) : nullptr;
    w.zero_words = (int64_t)((a.off - z0) / 4);
    w.rowptr = a.take<int32_t>(N + 1);
    w.rowptr_m = a.take<int32_t>(M + 1);
    w.colptr = a.take<int32_t>(N + 2);
    w.perm = a.take<int32_t>(E);
    w.csrc = a.take<int32_t>(E);
    w.gr = a.take<T>(3 * E);
    w.plan_tmp = a.take<char>((int64_t)be.transpose_tmp_bytes(E, N));
    w.d = a.take<T>(E); w.u = a.take<T>(3 * E); w.fc = a.take<T>(E); w.fc1 = a.take<T>(E);
    w.phi2 = a.take<T>(2 * E * K);
    w.dt = a.take<T>(E);
    w.ut = painn ? a.take<T>(3 * E) : nullptr;
    w.preh2 = a.take<T>(2 * N * H); w.th2 = a.take<T>(2 * N * H); w.e_atom = a.take<T>(N);
    w.gt2 = a.take<T>(2 * N * H); w.gpre2 = a.take<T>(2 * N * H); w.gEa2 = a.take<T>(2 * N);
    w.onehot = a.take<T>(N * (int64_t)n_types);
    w.gemm_ws = a.take<T>(gemm_ws_floats);
  }

  int prepare(const FmBatch<T>& b, const FmRadial<T>& rb, FmCommonWs<T>& w, int32_t* err) {
    int rc;
    if ((rc = be.zero_u32(w.tickets, w.zero_words))) return rc;
    if ((rc = be.rowptr2(b.ii, b.E, b.N, w.rowptr, b.idx_m, b.N, b.M, w.rowptr_m, err))) return rc;      // (one launch on the device)
    if ((rc = be.transpose_plan(b.jj, b.E, b.N, w.colptr, w.perm, w.plan_tmp))) return rc;
    be.flat("fm_colsrc", k_fm_colsrc<T>, b.E, w.perm, b.ii, b.E, b.N, w.csrc, (T*)nullptr);
    be.flat("fm_geom", k_fm_geom<T>, b.E * rb.n_rbf, b.R, b.off, b.ii, b.jj, b.E, b.N, rb, w.d, w.u, w.fc, w.fc1, w.phi2, w.e_act);
    return 0;
  }

  // energy head on x [N, F] (atomistic/atomwise.py:69-88): pre_h, th = act(pre_h), e_i, E_m
  int head_forward(const FmBatch<T>& b, const FmHead<T>& hd, int F, const T* x, FmCommonWs<T>& w, T* E_out, const int32_t* err) {
    const int H = hd.n_hidden;
    int rc;
    if ((rc = be.dense(x, hd.w1, hd.b1, nullptr, w.th2, w.preh2, b.N, F, H, hd.act))) return rc;
    be.rows("fm_head_e", k_fm_rowdot_bias<T>, b.N, w.th2, hd.w2, hd.b2, b.N, H, w.e_atom);
    be.rows("fm_head_E", k_fm_segsum1<T>, b.M, w.e_atom, w.rowptr_m, b.M, err, E_out);
    return 0;
  }
  // dE_tot/dx -> gx [N, F]
  int head_backward_R(const FmBatch<T>& b, const FmHead<T>& hd, int F, FmCommonWs<T>& w, T* gx) {
    const int H = hd.n_hidden;
    be.flat("fm_bcast", k_fm_bcast_rows<T>, b.N * H, hd.w2, (const T*)nullptr, (const int64_t*)nullptr, b.N, H, (int64_t)0, w.gt2);
    return be.dense_bwd_input(w.gt2, w.preh2, hd.w1, nullptr, gx, b.N, F, H, hd.act);
  }
  int head_tangent(const FmBatch<T>& b, const FmHead<T>& hd, int F, const T* xt, FmCommonWs<T>& w) {
    const int H = hd.n_hidden;
    return be.dense_tangent(xt, hd.w1, w.preh2, w.th2 + b.N * H, w.preh2 + b.N * H, b.N, F, H, hd.act, false);
  }
  // reverse of the dual head: S = sum_i gE_i e_i + sum_i et_i; x2 = [x ; xt]; -> gx2 = [gx ; hx] [2N, F] and the head's gradients
  int head_dual_backward(const FmBatch<T>& b, const FmHead<T>& hd, int F, const T* x2, const T* gE, FmCommonWs<T>& w, T* gx2, T* g_head) {
    const int H = hd.n_hidden;
    const int64_t N = b.N;
    T* g_w1 = g_head;
    T* g_b1 = g_w1 + (int64_t)H * F;
    T* g_w2 = g_b1 + H;
    T* g_b2 = g_w2 + H;
    int rc;
    be.flat("fm_gather1", k_fm_gather1<T>, N, gE, b.idx_m, N, b.M, w.gEa2, w.gEa2 + N);
    if ((rc = be.gemm_tn(w.gEa2, w.th2, 2 * N, 1, H, g_w2, g_b2, N))) return rc;
    be.flat("fm_head_dual_cot", k_fm_head_dual_cot<T>, N * H, hd.w2, (const T*)w.gEa2, (const T*)w.preh2, N, H, hd.act, w.gpre2);
    if ((rc = be.gemm_tn(w.gpre2, x2, 2 * N, H, F, g_w1, g_b1, N))) return rc;
    return be.dense_bwd_input(w.gpre2, nullptr, hd.w1, nullptr, gx2, 2 * N, F, H, FM_ACT_NONE);
  }

  // ================================================================================================ SchNet
  struct SchnetWs {
    FmCommonWs<T> c;
    std::vector<T*> X2, h2, a2, z2, Wf2, y2, p32, s2;
    std::vector<T*> GX, gp2, gh2, gg2, ga2;     // per interaction: operands of the weight-gradient GEMMs, which run as ONE batched launch at the end of pass D
    T *gs2, *gy2, *gz2;
    size_t bytes;
  };
  void schnet_carve(void* base, const FmSchnetModel<T>& m, int K, int H, int64_t N, int64_t E, int64_t M, int n_types, SchnetWs& w) {
    FmArena a(base);
    const int F = m.F, nf = m.nf, L = m.L;
    int64_t gw = 0;      // the weight-gradient GEMMs of a pass run concurrently (one batched launch): their slice workspaces add up
    auto need = [&](int64_t n, int O, int Kk, int times) { gw += times * be.gemm_tn_ws_floats(n, O, Kk); };
    need(N, n_types, F, 1); need(2 * N, 1, H, 1); need(2 * N, H, F, 1); need(2 * N, F, F, L); need(2 * N, F, nf, L); need(2 * E, nf, nf, L); need(2 * E, nf, K, L); need(2 * N, nf, F, L);
    carve_common(a, w.c, N, E, M, K, H, n_types, false, gw);
    w.X2.resize(L + 1);
    for (int l = 0; l <= L; ++l) w.X2[l] = a.take<T>(2 * N * F);
    w.h2.resize(L); w.a2.resize(L); w.z2.resize(L); w.Wf2.resize(L); w.y2.resize(L); w.p32.resize(L); w.s2.resize(L);
    for (int l = 0; l < L; ++l) {
      w.h2[l] = a.take<T>(2 * N * nf); w.a2[l] = a.take<T>(2 * E * nf); w.z2[l] = a.take<T>(2 * E * nf); w.Wf2[l] = a.take<T>(2 * E * nf);
      w.y2[l] = a.take<T>(2 * N * nf); w.p32[l] = a.take<T>(2 * N * F); w.s2[l] = a.take<T>(2 * N * F);
    }
    w.GX.resize(L + 1);
    for (int l = 0; l <= L; ++l) w.GX[l] = a.take<T>(2 * N * F);
    w.gp2.resize(L); w.gh2.resize(L); w.gg2.resize(L); w.ga2.resize(L);
    for (int l = 0; l < L; ++l) {
      w.gp2[l] = a.take<T>(2 * N * F); w.gh2[l] = a.take<T>(2 * N * nf); w.gg2[l] = a.take<T>(2 * E * nf); w.ga2[l] = a.take<T>(2 * E * nf);
    }
    w.gs2 = a.take<T>(2 * N * F); w.gy2 = a.take<T>(2 * N * nf); w.gz2 = a.take<T>(2 * E * nf);
    w.bytes = a.off;
  }

  // passes A + B
  int schnet_forward(const FmSchnetModel<T>& m, const FmHead<T>& hd, const FmRadial<T>& rb, const FmBatch<T>& b, void* ws, T* E_out, T* F_out, int32_t* err) {
    SchnetWs w;
    const int F = m.F, nf = m.nf, L = m.L, K = rb.n_rbf;
    const int64_t N = b.N, E = b.E;
    schnet_carve(ws, m, K, hd.n_hidden, N, E, b.M, b.n_types, w);
    be.set_gemm_ws(w.c.gemm_ws, w.c.tickets);
    int rc;
    if ((rc = prepare(b, rb, w.c, err))) return rc;
    be.chain_begin(N);      // atom-local launches from here on may leave as row chains (device backend, small batches)
    be.flat("fm_embed", k_fm_embed<T>, N * F, b.emb, b.Z, N, F, b.n_types, w.X2[0], w.c.onehot, err);
    // ---- pass A.  The filter networks depend on the geometry only: they are issued first, on two side streams (interactions alternate),
    // and the atom chain on the main stream waits for interaction l's filters in front of its convolution.
    const bool par = be.can_fork(L);
    for (int l = 0; l < L; ++l) {
      const FmSchnetLayer<T>& P = m.layers[l];
      T *a2 = w.a2[l], *z2 = w.z2[l], *Wf2 = w.Wf2[l];
      if (par) be.fork(l & 1);
      // filter network, value and d-derivative (schnet.py:61): a = phi W1^T + b1, a1 = phi1 W1^T; z = ssp(a), z1 = ssp'(a) a1; g = z W2^T + b2, g1 = z1 W2^T
      // (each a Dense layer on the (value, d-derivative) pair: one launch, the activation / cutoff product in its epilogue)
      if ((rc = be.dense_dual(w.c.phi2, P.fn_w1, P.fn_b1, z2, a2, E, K, nf, FM_ACT_SSP, nullptr, nullptr))) return rc;
      if ((rc = be.dense_dual(z2, P.fn_w2, P.fn_b2, Wf2, nullptr, E, nf, nf, FM_ACT_NONE, w.c.fc, w.c.fc1))) return rc;
      if (par) be.back(l);
    }
    for (int l = 0; l < L; ++l) {
      const FmSchnetLayer<T>& P = m.layers[l];
      T* Wf2 = w.Wf2[l];
      if ((rc = be.dense(w.X2[l], P.in2f_w, nullptr, nullptr, w.h2[l], nullptr, N, F, nf, FM_ACT_NONE))) return rc;
      if (par) be.wait(l);
      be.slotted("fm_cfconv", k_fm_cfconv<T>, N * nf, w.h2[l], Wf2, w.c.rowptr, b.jj, w.c.e_act, N, nf, w.y2[l]);
      if ((rc = be.dense(w.y2[l], P.f2out_w1, P.f2out_b1, nullptr, w.s2[l], w.p32[l], N, nf, F, FM_ACT_SSP))) return rc;
      if ((rc = be.dense(w.s2[l], P.f2out_w2, P.f2out_b2, w.X2[l], w.X2[l + 1], nullptr, N, F, F, FM_ACT_NONE))) return rc;
    }
    if ((rc = head_forward(b, hd, F, w.X2[L], w.c, E_out, err))) return rc;
    if (!F_out) return be.chain_end();
    T *gxa = w.GX[L], *gxb = w.GX[0], *ghb = w.gh2[0];                      // pass B borrows pass-D buffers (it ends before D starts)
    if ((rc = head_backward_R(b, hd, F, w.c, gxa))) return rc;              // ---- pass B
    for (int l = L - 1; l >= 0; --l) {
      const FmSchnetLayer<T>& P = m.layers[l];
      if ((rc = be.dense_bwd_input(gxa, nullptr, P.f2out_w2, nullptr, w.gs2, N, F, F, FM_ACT_NONE))) return rc;
      if ((rc = be.dense_bwd_input(w.gs2, w.p32[l], P.f2out_w1, nullptr, w.gy2, N, nf, F, FM_ACT_SSP))) return rc;
      be.rows("fm_cfconv_gd", k_fm_cfconv_gd<T>, E, w.gy2, w.h2[l], w.Wf2[l] + E * nf, b.ii, b.jj, E, N, nf, w.c.gd);
      if (l > 0) {
        be.slotted("fm_cfconv_T", k_fm_cfconv_T<T>, N * nf, w.gy2, w.Wf2[l], w.c.colptr, w.c.perm, w.c.csrc, w.c.e_act, N, nf, ghb);
        if ((rc = be.dense_bwd_input(ghb, nullptr, P.in2f_w, gxa, gxb, N, F, nf, FM_ACT_NONE))) return rc;
        T* t = gxa; gxa = gxb; gxb = t;
      }
    }
    be.flat("fm_gr", k_fm_gr<T>, E, w.c.gd, (const T*)nullptr, w.c.u, w.c.d, E, w.c.gr);
    be.flat("fm_force", k_fm_force<T>, N * 3, w.c.gr, w.c.rowptr, w.c.colptr, w.c.perm, w.c.e_act, N, F_out);
    return be.chain_end();
  }

  // passes C + D; `grads` in the flat layout of fm_schnet_grad_floats(); the workspace must be the one the forward call filled
  int schnet_backward(const FmSchnetModel<T>& m, const FmHead<T>& hd, const FmRadial<T>& rb, const FmBatch<T>& b, void* ws, const T* gE, const T* gF, T* grads) {
    SchnetWs w;
    const int F = m.F, nf = m.nf, L = m.L, K = rb.n_rbf, H = hd.n_hidden;
    const int64_t N = b.N, E = b.E;
    schnet_carve(ws, m, K, H, N, E, b.M, b.n_types, w);
    be.set_gemm_ws(w.c.gemm_ws, w.c.tickets);
    int rc;
    be.chain_begin(N);
    be.flat("fm_tgeom", k_fm_tgeom<T>, E, gF, b.ii, b.jj, w.c.d, w.c.u, E, N, w.c.dt, (T*)nullptr);
    for (int l = 0; l < L; ++l) {                                            // ---- pass C
      const FmSchnetLayer<T>& P = m.layers[l];
      T* ht = l > 0 ? w.h2[l] + N * nf : nullptr;
      if (l > 0 && (rc = be.dense(w.X2[l] + N * F, P.in2f_w, nullptr, nullptr, ht, nullptr, N, F, nf, FM_ACT_NONE))) return rc;
      be.slotted("fm_cfconv_t", k_fm_cfconv_t<T>, N * nf, w.h2[l], (const T*)ht, w.Wf2[l], w.Wf2[l] + E * nf, w.c.dt, w.c.rowptr, b.jj, w.c.e_act, N, nf, w.y2[l] + N * nf);
      if ((rc = be.dense_tangent(w.y2[l] + N * nf, P.f2out_w1, w.p32[l], w.s2[l] + N * F, w.p32[l] + N * F, N, nf, F, FM_ACT_SSP, false))) return rc;
      if ((rc = be.dense(w.s2[l] + N * F, P.f2out_w2, nullptr, l > 0 ? w.X2[l] + N * F : nullptr, w.X2[l + 1] + N * F, nullptr, N, F, F, FM_ACT_NONE))) return rc;
    }
    if ((rc = head_tangent(b, hd, F, w.X2[L] + N * F, w.c))) return rc;
    const int64_t lg = fm_schnet_layer_grad_floats(F, nf, K);
    T* g_head = grads + L * lg;
    T* g_emb = g_head + fm_head_grad_floats(F, H);
    if ((rc = head_dual_backward(b, hd, F, w.X2[L], gE, w.c, w.GX[L], g_head))) return rc;
    const bool parD = be.can_fork(L);
    for (int l = L - 1; l >= 0; --l) {                                       // ---- pass D
      const FmSchnetLayer<T>& P = m.layers[l];
      T* g = grads + l * lg;
      T* g_in2f = g; g += (int64_t)nf * F;
      T* g_w1 = g; g += (int64_t)nf * K;
      T* g_b1 = g; g += nf;
      T* g_w2 = g; g += (int64_t)nf * nf;
      T* g_b2 = g; g += nf;
      T* g_o1 = g; g += (int64_t)F * nf;
      T* g_ob1 = g; g += F;
      T* g_o2 = g; g += (int64_t)F * F;
      T* g_ob2 = g;
      T *gx = w.GX[l + 1], *gp2 = w.gp2[l], *gh2 = w.gh2[l], *gg2 = w.gg2[l], *ga2 = w.ga2[l];
      const int64_t nr = l > 0 ? 2 * N : N;      // rows that carry a tangent partner at the INPUT of this interaction (xt_0 = 0)
      if ((rc = be.gemm_tn(gx, w.s2[l], 2 * N, F, F, g_o2, g_ob2, N))) return rc;
      if ((rc = be.dense_dual_bwd(gx, P.f2out_w2, w.p32[l], gp2, w.gs2, N, F, F, FM_ACT_SSP))) return rc;
      if ((rc = be.gemm_tn(gp2, w.y2[l], 2 * N, F, nf, g_o1, g_ob1, N))) return rc;
      if ((rc = be.dense_bwd_input(gp2, nullptr, P.f2out_w1, nullptr, w.gy2, 2 * N, nf, F, FM_ACT_NONE))) return rc;
      be.slotted("fm_cfconv_T_dual", k_fm_cfconv_T_dual<T>, N * nf, w.gy2, w.Wf2[l], w.c.dt, w.c.colptr, w.c.perm, w.c.csrc, w.c.e_act, N, E, nf, gh2);
      be.flat("fm_filter_cot", k_fm_filter_cot<T>, E * nf, w.gy2, w.h2[l], (const T*)(l > 0 ? w.h2[l] + N * nf : nullptr), w.c.dt, w.c.fc, w.c.fc1, b.ii, b.jj, N, E,
              nf, gg2);
      if ((rc = be.gemm_tn(gg2, w.z2[l], 2 * E, nf, nf, g_w2, g_b2, E))) return rc;
      // the reverse of the filter network feeds only weight gradients (deferred to the batched launch): off the atom chain, on ONE side
      // stream for all interactions (they share the scratch gz2)
      if (parD) be.fork(0);
      if ((rc = be.dense_dual_bwd(gg2, P.fn_w2, w.a2[l], ga2, w.gz2, E, nf, nf, FM_ACT_SSP))) return rc;
      if (parD) be.back(l);
      if ((rc = be.gemm_tn(ga2, w.c.phi2, 2 * E, nf, K, g_w1, g_b1, E))) return rc;
      if ((rc = be.gemm_tn(gh2, w.X2[l], nr, nf, F, g_in2f, nullptr, nr))) return rc;
      if ((rc = be.dense_bwd_input(gh2, nullptr, P.in2f_w, gx, w.GX[l], nr, F, nf, FM_ACT_NONE))) return rc;
    }
    if (parD) for (int l = 0; l < L; ++l) be.wait(l);
    if ((rc = be.gemm_tn(w.c.onehot, w.GX[0], N, b.n_types, F, g_emb, nullptr, N))) return rc;      // embedding table: onehot(Z)^T gx_0
    if ((rc = be.chain_end())) return rc;
    return be.gemm_flush();
  }

  // ================================================================================================ PaiNN
  struct PainnWs {
    FmCommonWs<T> c;
    T* Phi2;
    std::vector<T*> Q2, MU2, pa2, sa2, c2, q1_2, mu1_2, VW2, n2, svw2, ctx2, pb2, sb2, a2;
    std::vector<T*> ga2, gpb2, gVW2, gP2, gc2, gpa2;     // per interaction: operands of the weight-gradient GEMMs (one batched launch at the end of pass D)
    T *gq_a, *gq_b, *gmu_a, *gmu_b, *gsb2, *gctx2, *gq1_2, *gmu1_2, *gsa2, *gtmp;
    size_t bytes;
  };
  void painn_carve(void* base, const FmPainnModel<T>& m, int K, int H, int64_t N, int64_t E, int64_t M, int n_types, PainnWs& w) {
    FmArena a(base);
    const int F = m.F, L = m.L;
    const int64_t ld = 3ll * F * (m.shared_filters ? 1 : L);
    int64_t gw = 0;
    auto need = [&](int64_t n, int O, int Kk, int times) { gw += times * be.gemm_tn_ws_floats(n, O, Kk); };
    need(N, n_types, F, 1); need(2 * N, 1, H, 1); need(2 * N, H, F, 1); need(2 * N, 3 * F, F, 2 * L); need(2 * N, F, 2 * F, L); need(6 * N, 2 * F, F, L); need(2 * E, 3 * F, K, L); need(2 * N, F, F, L);
    carve_common(a, w.c, N, E, M, K, H, n_types, true, gw);
    w.Phi2 = a.take<T>(2 * E * ld);
    auto vec = [&](std::vector<T*>& v, int n, int64_t floats) { v.resize(n); for (int l = 0; l < n; ++l) v[l] = a.take<T>(floats); };
    vec(w.Q2, L + 1, 2 * N * F);
    vec(w.MU2, L + 1, 6 * N * F);
    vec(w.pa2, L, 2 * N * F); vec(w.sa2, L, 2 * N * F); vec(w.c2, L, 6 * N * F); vec(w.q1_2, L, 2 * N * F); vec(w.mu1_2, L, 6 * N * F);
    vec(w.VW2, L, 12 * N * F); vec(w.n2, L, 2 * N * F); vec(w.svw2, L, 2 * N * F); vec(w.ctx2, L, 4 * N * F); vec(w.pb2, L, 2 * N * F);
    vec(w.sb2, L, 2 * N * F); vec(w.a2, L, 6 * N * F);
    w.gq_a = a.take<T>(2 * N * F); w.gq_b = a.take<T>(2 * N * F); w.gmu_a = a.take<T>(6 * N * F); w.gmu_b = a.take<T>(6 * N * F);
    vec(w.ga2, L, 6 * N * F); vec(w.gpb2, L, 2 * N * F); vec(w.gVW2, L, 12 * N * F); vec(w.gP2, L, 6 * E * F); vec(w.gc2, L, 6 * N * F); vec(w.gpa2, L, 2 * N * F);
    w.gsb2 = a.take<T>(2 * N * F); w.gctx2 = a.take<T>(4 * N * F); w.gq1_2 = a.take<T>(2 * N * F); w.gmu1_2 = a.take<T>(6 * N * F);
    w.gsa2 = a.take<T>(2 * N * F);
    w.gtmp = a.take<T>(m.shared_filters ? (int64_t)L * (3ll * F * K + 3 * F) : 1);
    w.bytes = a.off;
  }

  int painn_forward(const FmPainnModel<T>& m, const FmHead<T>& hd, const FmRadial<T>& rb, const FmBatch<T>& b, void* ws, T* E_out, T* F_out, int32_t* err) {
    PainnWs w;
    const int F = m.F, L = m.L, K = rb.n_rbf;
    const int64_t N = b.N, E = b.E;
    const int ld = 3 * F * (m.shared_filters ? 1 : L);
    painn_carve(ws, m, K, hd.n_hidden, N, E, b.M, b.n_types, w);
    be.set_gemm_ws(w.c.gemm_ws, w.c.tickets);
    int rc;
    if ((rc = prepare(b, rb, w.c, err))) return rc;
    be.chain_begin(N);
    // every interaction's filter rows at once (painn.py:232-236): Phi = (phi Wf^T + bf) f_c, with the d-derivative beside it -- on a side
    // stream, beside the embedding and the first context net (they meet in front of the first message)
    const bool par = be.can_fork(1);
    if (par) be.fork(0);
    if ((rc = be.dense_dual(w.c.phi2, m.filt_w, m.filt_b, w.Phi2, nullptr, E, K, ld, FM_ACT_NONE, w.c.fc, w.c.fc1))) return rc;
    if (par) be.back(0);
    be.flat("fm_embed", k_fm_embed<T>, N * F, b.emb, b.Z, N, F, b.n_types, w.Q2[0], w.c.onehot, err);
    for (int l = 0; l < L; ++l) {                                            // ---- pass A
      const FmPainnLayer<T>& P = m.layers[l];
      const T* Phi = w.Phi2 + (m.shared_filters ? 0 : 3 * F * l);
      const T* mu = l > 0 ? w.MU2[l] : nullptr;
      if ((rc = be.dense(w.Q2[l], P.ctx_w1, P.ctx_b1, nullptr, w.sa2[l], w.pa2[l], N, F, F, FM_ACT_SILU))) return rc;
      if ((rc = be.dense(w.sa2[l], P.ctx_w2, P.ctx_b2, nullptr, w.c2[l], nullptr, N, F, 3 * F, FM_ACT_NONE))) return rc;
      if (par && l == 0) be.wait(0);
      be.slotted("fm_painn_msg", k_fm_painn_msg<T>, N * F, w.Q2[l], mu, w.c2[l], Phi, ld, w.c.u, w.c.rowptr, b.jj, w.c.e_act, N, F, w.q1_2[l], w.mu1_2[l]);
      if ((rc = be.dense(w.mu1_2[l], P.mix_w, nullptr, nullptr, w.VW2[l], nullptr, 3 * N, F, 2 * F, FM_ACT_NONE))) return rc;
      ew(FM_EW_MIX, N, F, m.eps, {w.q1_2[l], w.VW2[l]}, {w.n2[l], w.svw2[l], w.ctx2[l]});
      if ((rc = be.dense(w.ctx2[l], P.ictx_w1, P.ictx_b1, nullptr, w.sb2[l], w.pb2[l], N, 2 * F, F, FM_ACT_SILU))) return rc;
      if ((rc = be.dense(w.sb2[l], P.ictx_w2, P.ictx_b2, nullptr, w.a2[l], nullptr, N, F, 3 * F, FM_ACT_NONE))) return rc;
      ew(FM_EW_UPDATE, N, F, T(0), {w.q1_2[l], w.mu1_2[l], w.VW2[l], w.a2[l], w.svw2[l]}, {w.Q2[l + 1], w.MU2[l + 1]});
    }
    if ((rc = head_forward(b, hd, F, w.Q2[L], w.c, E_out, err))) return rc;
    if (!F_out) return be.chain_end();
    if ((rc = head_backward_R(b, hd, F, w.c, w.gq_a))) return rc;           // ---- pass B
    const T* gmu = nullptr;                                                  // the head does not read the vector representation
    T *ga = w.ga2[0], *gVW = w.gVW2[0], *gc = w.gc2[0];                      // pass B borrows pass-D buffers (it ends before D starts)
    for (int l = L - 1; l >= 0; --l) {
      const FmPainnLayer<T>& P = m.layers[l];
      const T* Phi = w.Phi2 + (m.shared_filters ? 0 : 3 * F * l);
      const T* mu = l > 0 ? w.MU2[l] : nullptr;
      ew(FM_EW_UPDATE_BWD, N, F, T(0), {w.gq_a, gmu, w.VW2[l], w.a2[l], w.svw2[l]}, {ga, gVW});
      if ((rc = be.dense_bwd_input(ga, nullptr, P.ictx_w2, nullptr, w.gsb2, N, F, 3 * F, FM_ACT_NONE))) return rc;
      if ((rc = be.dense_bwd_input(w.gsb2, w.pb2[l], P.ictx_w1, nullptr, w.gctx2, N, 2 * F, F, FM_ACT_SILU))) return rc;
      ew(FM_EW_MIX_BWD, N, F, T(0), {w.gq_a, w.gctx2, w.VW2[l], w.n2[l]}, {w.gq1_2, gVW});
      if ((rc = be.dense_bwd_input(gVW, nullptr, P.mix_w, gmu, w.gmu1_2, 3 * N, F, 2 * F, FM_ACT_NONE))) return rc;
      be.rows("fm_painn_msg_gd", k_fm_painn_msg_gd<T>, E, w.gq1_2, w.gmu1_2, w.c2[l], mu, Phi, ld, w.c.u, b.ii, b.jj, E, N, F, w.c.gd, w.c.gu);
      if (l > 0) {
        be.slotted("fm_painn_msg_T", k_fm_painn_msg_T<T>, N * F, w.gq1_2, w.gmu1_2, w.c2[l], mu, Phi, ld, w.c.u, w.c.colptr, w.c.perm, w.c.csrc, w.c.e_act, N, F, gc, w.gmu_a);
        if ((rc = be.dense_bwd_input(gc, nullptr, P.ctx_w2, nullptr, w.gsa2, N, F, 3 * F, FM_ACT_NONE))) return rc;
        if ((rc = be.dense_bwd_input(w.gsa2, w.pa2[l], P.ctx_w1, w.gq1_2, w.gq_a, N, F, F, FM_ACT_SILU))) return rc;
        gmu = w.gmu_a;
        T* t = w.gmu_a; w.gmu_a = w.gmu_b; w.gmu_b = t;
      }
    }
    be.flat("fm_gr", k_fm_gr<T>, E, w.c.gd, (const T*)w.c.gu, w.c.u, w.c.d, E, w.c.gr);
    be.flat("fm_force", k_fm_force<T>, N * 3, w.c.gr, w.c.rowptr, w.c.colptr, w.c.perm, w.c.e_act, N, F_out);
    return be.chain_end();
  }

  // `grads` in the flat layout of fm_painn_grad_floats()
  int painn_backward(const FmPainnModel<T>& m, const FmHead<T>& hd, const FmRadial<T>& rb, const FmBatch<T>& b, void* ws, const T* gE, const T* gF, T* grads) {
    PainnWs w;
    const int F = m.F, L = m.L, K = rb.n_rbf, H = hd.n_hidden;
    const int64_t N = b.N, E = b.E;
    const int Lf = m.shared_filters ? 1 : L;
    const int ld = 3 * F * Lf;
    painn_carve(ws, m, K, H, N, E, b.M, b.n_types, w);
    be.set_gemm_ws(w.c.gemm_ws, w.c.tickets);
    int rc;
    be.chain_begin(N);
    be.flat("fm_tgeom", k_fm_tgeom<T>, E, gF, b.ii, b.jj, w.c.d, w.c.u, E, N, w.c.dt, w.c.ut);
    const int64_t NF = N * F;
    for (int l = 0; l < L; ++l) {                                            // ---- pass C
      const FmPainnLayer<T>& P = m.layers[l];
      const T* Phi = w.Phi2 + (m.shared_filters ? 0 : 3 * F * l);
      const int first = l == 0;
      if (!first) {
        if ((rc = be.dense_tangent(w.Q2[l] + NF, P.ctx_w1, w.pa2[l], w.sa2[l] + NF, w.pa2[l] + NF, N, F, F, FM_ACT_SILU, false))) return rc;
        if ((rc = be.dense(w.sa2[l] + NF, P.ctx_w2, nullptr, nullptr, w.c2[l] + 3 * NF, nullptr, N, F, 3 * F, FM_ACT_NONE))) return rc;
      }
      be.slotted("fm_painn_msg_t", k_fm_painn_msg_t<T>, NF, (const T*)(first ? nullptr : w.Q2[l] + NF), w.c2[l], (const T*)(first ? nullptr : w.MU2[l]), Phi, ld, E, w.c.dt,
              w.c.u, w.c.ut, w.c.rowptr, b.jj, w.c.e_act, N, F, first, w.q1_2[l] + NF, w.mu1_2[l] + 3 * NF);
      if ((rc = be.dense(w.mu1_2[l] + 3 * NF, P.mix_w, nullptr, nullptr, w.VW2[l] + 6 * NF, nullptr, 3 * N, F, 2 * F, FM_ACT_NONE))) return rc;
      ew(FM_EW_MIX_T, N, F, T(0), {w.q1_2[l] + NF, w.VW2[l], w.VW2[l] + 6 * NF, w.n2[l]}, {w.n2[l] + NF, w.svw2[l] + NF, w.ctx2[l] + 2 * NF});
      if ((rc = be.dense_tangent(w.ctx2[l] + 2 * NF, P.ictx_w1, w.pb2[l], w.sb2[l] + NF, w.pb2[l] + NF, N, 2 * F, F, FM_ACT_SILU, false))) return rc;
      if ((rc = be.dense(w.sb2[l] + NF, P.ictx_w2, nullptr, nullptr, w.a2[l] + 3 * NF, nullptr, N, F, 3 * F, FM_ACT_NONE))) return rc;
      ew(FM_EW_UPDATE_T, N, F, T(0), {w.q1_2[l] + NF, w.mu1_2[l] + 3 * NF, w.VW2[l], w.VW2[l] + 6 * NF, w.a2[l], w.a2[l] + 3 * NF, w.svw2[l], w.svw2[l] + NF},
         {w.Q2[l + 1] + NF, w.MU2[l + 1] + 3 * NF});
    }
    if ((rc = head_tangent(b, hd, F, w.Q2[L] + NF, w.c))) return rc;
    const int64_t lg = fm_painn_layer_grad_floats(F);
    T* g_fw = grads + L * lg;
    T* g_fb = g_fw + 3ll * F * Lf * K;
    T* g_head = g_fb + 3ll * F * Lf;
    T* g_emb = g_head + fm_head_grad_floats(F, H);
    if ((rc = head_dual_backward(b, hd, F, w.Q2[L], gE, w.c, w.gq_a, g_head))) return rc;
    const T* gmu2 = nullptr;
    for (int l = L - 1; l >= 0; --l) {                                       // ---- pass D
      const FmPainnLayer<T>& P = m.layers[l];
      const T* Phi = w.Phi2 + (m.shared_filters ? 0 : 3 * F * l);
      const int first = l == 0;
      T* g = grads + l * lg;
      T* g_cw1 = g; g += (int64_t)F * F;
      T* g_cb1 = g; g += F;
      T* g_cw2 = g; g += 3ll * F * F;
      T* g_cb2 = g; g += 3 * F;
      T* g_mix = g; g += 2ll * F * F;
      T* g_iw1 = g; g += 2ll * F * F;
      T* g_ib1 = g; g += F;
      T* g_iw2 = g; g += 3ll * F * F;
      T* g_ib2 = g;
      T *ga2 = w.ga2[l], *gpb2 = w.gpb2[l], *gVW2 = w.gVW2[l], *gP2 = w.gP2[l], *gc2 = w.gc2[l], *gpa2 = w.gpa2[l];
      // mixing (painn.py:99-116)
      ew(FM_EW_UPDATE_DUAL_BWD, N, F, T(0), {w.gq_a, gmu2, w.VW2[l], w.a2[l], w.svw2[l]}, {ga2, gVW2});
      if ((rc = be.gemm_tn(ga2, w.sb2[l], 2 * N, 3 * F, F, g_iw2, g_ib2, N))) return rc;
      if ((rc = be.dense_dual_bwd(ga2, P.ictx_w2, w.pb2[l], gpb2, w.gsb2, N, F, 3 * F, FM_ACT_SILU))) return rc;
      if ((rc = be.gemm_tn(gpb2, w.ctx2[l], 2 * N, F, 2 * F, g_iw1, g_ib1, N))) return rc;
      if ((rc = be.dense_bwd_input(gpb2, nullptr, P.ictx_w1, nullptr, w.gctx2, 2 * N, 2 * F, F, FM_ACT_NONE))) return rc;
      ew(FM_EW_MIX_DUAL_BWD, N, F, T(0), {w.gq_a, w.gctx2, w.VW2[l], w.n2[l]}, {w.gq1_2, gVW2});
      if ((rc = be.gemm_tn(gVW2, w.mu1_2[l], 6 * N, 2 * F, F, g_mix, nullptr, 6 * N))) return rc;
      if ((rc = be.dense_bwd_input(gVW2, nullptr, P.mix_w, gmu2, w.gmu1_2, 6 * N, F, 2 * F, FM_ACT_NONE))) return rc;
      // message (painn.py:50-66)
      const T* mu2 = first ? nullptr : w.MU2[l];
      be.flat("fm_painn_filter_cot", k_fm_painn_filter_cot<T>, E * F, w.gq1_2, w.gmu1_2, w.c2[l], mu2, w.c.dt, w.c.u, w.c.ut, w.c.fc, w.c.fc1, b.ii, b.jj, N, E, F, first,
              gP2);
      if (m.shared_filters && l != L - 1) {
        // one filter slice shared by all interactions: the slots of the interactions are added up after the batched launch
        T* slot = w.gtmp + (int64_t)l * (3ll * F * K + 3 * F);
        if ((rc = be.gemm_tn(gP2, w.c.phi2, 2 * E, 3 * F, K, slot, slot + 3ll * F * K, E))) return rc;
      } else {
        const int64_t row0 = m.shared_filters ? 0 : 3ll * F * l;
        if ((rc = be.gemm_tn(gP2, w.c.phi2, 2 * E, 3 * F, K, g_fw + row0 * K, g_fb + row0, E))) return rc;
      }
      be.slotted("fm_painn_msg_T_dual", k_fm_painn_msg_T_dual<T>, NF, w.gq1_2, w.gmu1_2, w.c2[l], mu2, Phi, ld, E, w.c.dt, w.c.u, w.c.ut, w.c.colptr, w.c.perm, w.c.csrc,
              w.c.e_act, N, F, first, gc2, w.gmu_a);
      const int64_t nr = first ? N : 2 * N;
      if ((rc = be.gemm_tn(gc2, w.sa2[l], nr, 3 * F, F, g_cw2, g_cb2, N))) return rc;
      // (the first interaction has no tangent at its input: only the value cotangent flows, g_a = act'(a) g_s)
      if (first) { if ((rc = be.dense_tangent(gc2, P.ctx_w2, w.pa2[l], gpa2, w.gsa2, N, 3 * F, F, FM_ACT_SILU, true))) return rc; }
      else if ((rc = be.dense_dual_bwd(gc2, P.ctx_w2, w.pa2[l], gpa2, w.gsa2, N, F, 3 * F, FM_ACT_SILU))) return rc;
      if ((rc = be.gemm_tn(gpa2, w.Q2[l], nr, F, F, g_cw1, g_cb1, N))) return rc;
      if ((rc = be.dense_bwd_input(gpa2, nullptr, P.ctx_w1, w.gq1_2, w.gq_a, nr, F, F, FM_ACT_NONE))) return rc;
      gmu2 = w.gmu_a;
      T* t = w.gmu_a; w.gmu_a = w.gmu_b; w.gmu_b = t;
    }
    if ((rc = be.gemm_tn(w.c.onehot, w.gq_a, N, b.n_types, F, g_emb, nullptr, N))) return rc;      // embedding table: onehot(Z)^T gq_0
    if ((rc = be.chain_end())) return rc;
    if ((rc = be.gemm_flush())) return rc;
    if (m.shared_filters)
      for (int l = 0; l < L - 1; ++l)
        be.flat("fm_axpy", k_fm_axpy<T>, 3ll * F * K + 3 * F, w.gtmp + (int64_t)l * (3ll * F * K + 3 * F), 3ll * F * K + 3 * F, g_fw);   // (g_fw and g_fb are adjacent when Lf == 1)
    return 0;
  }
};
