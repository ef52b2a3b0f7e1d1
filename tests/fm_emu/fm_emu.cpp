// TEST INFRASTRUCTURE ONLY -- serial CPU instantiation of the force-matching gradient engine (schnetpack_amd/csrc/spk_fm_engine.h,
// spk_fm_kernels.h) in float64 and float32: the SAME orchestration and the SAME kernel bodies as the HIP build, with naive loops for
// the Dense / weight-gradient GEMMs and std::stable_sort for the by-neighbour CSR.  Built and loaded by tests/test_fm_engine_emu.py
// only; nothing under schnetpack_amd/ links or loads it (the product refuses CPU tensors).
#define SPK_FM_EMU 1
#include <algorithm>
#include <cstring>
#include <numeric>
#include "../../schnetpack_amd/csrc/spk_fm_engine.h"

template <class T>
struct EmuBackend {
  bool can_fork(int) { return true; }       // (the emulation runs everything in issue order: the fork / back / wait calls of the engine are exercised as calls)
  void fork(int) {}
  void back(int) {}
  void wait(int) {}
  void set_gemm_ws(T*, uint32_t*) {}
  // row chains are a launch-count device of the HIP backend: here every stage runs at once, in issue order
  void chain_begin(int64_t) {}
  int chain_end() { return 0; }
  void ew(const FmEwArgs<T>& a) { for (int64_t t = 0; t < a.N * a.F; ++t) fm_ew_at<T>(a, t); }
  int64_t gemm_tn_ws_floats(int64_t, int, int) { return 0; }
  size_t transpose_tmp_bytes(int64_t, int64_t) { return 0; }
  int zero_u32(uint32_t* p, int64_t n) { std::memset(p, 0, (size_t)n * 4); return 0; }
  int rowptr(const int64_t* idx, int64_t n, int64_t rows, int32_t* out, int32_t* err) {
    int64_t e = 0;
    for (int64_t r = 0; r <= rows; ++r) {
      while (e < n && idx[e] < r) ++e;
      out[r] = (int32_t)e;
    }
    for (int64_t k = 1; k < n; ++k) if (idx[k - 1] > idx[k] && err) *err |= 1;
    return 0;
  }
  int rowptr2(const int64_t* ia, int64_t na, int64_t ra, int32_t* oa, const int64_t* ib, int64_t nb, int64_t rb, int32_t* ob, int32_t* err) {
    const int rc = rowptr(ia, na, ra, oa, err);
    return rc ? rc : rowptr(ib, nb, rb, ob, err);
  }
  int transpose_plan(const int64_t* jj, int64_t E, int64_t N, int32_t* colptr, int32_t* perm, void*) {
    std::vector<int32_t> ids((size_t)E);
    std::iota(ids.begin(), ids.end(), 0);
    auto key = [&](int32_t e) { return (uint64_t)jj[e] < (uint64_t)N ? jj[e] : N; };
    std::stable_sort(ids.begin(), ids.end(), [&](int32_t a, int32_t b) { return key(a) < key(b); });
    int64_t p = 0;
    for (int64_t r = 0; r <= N + 1; ++r) {
      while (p < E && key(ids[(size_t)p]) < r) ++p;
      colptr[r] = (int32_t)p;
    }
    std::copy(ids.begin(), ids.end(), perm);
    return 0;
  }
  // y = act(x W^T + b) + res;  w [n_out, k]
  int dense(const T* x, const T* w, const T* b, const T* res, T* y, T* pre, int64_t m, int k, int n_out, int act) {
    for (int64_t r = 0; r < m; ++r)
      for (int o = 0; o < n_out; ++o) {
        T acc = b ? b[o] : T(0);
        for (int c = 0; c < k; ++c) acc += x[r * k + c] * w[(int64_t)o * k + c];
        if (pre) pre[r * n_out + o] = acc;
        T v = fm_act(act, 0, acc);
        if (res) v += res[r * n_out + o];
        y[r * n_out + o] = v;
      }
    return 0;
  }
  // dx = (dy * act'(pre)) W + res;  dy [m, n_out], w [n_out, k], dx [m, k]
  int dense_bwd_input(const T* dy, const T* pre, const T* w, const T* res, T* dx, int64_t m, int k, int n_out, int act) {
    std::vector<T> row((size_t)n_out);
    for (int64_t r = 0; r < m; ++r) {
      for (int o = 0; o < n_out; ++o) row[(size_t)o] = dy[r * n_out + o] * (pre ? fm_act(act, 1, pre[r * n_out + o]) : T(1));
      for (int c = 0; c < k; ++c) {
        T acc = res ? res[r * k + c] : T(0);
        for (int o = 0; o < n_out; ++o) acc += row[(size_t)o] * w[(int64_t)o * k + c];
        dx[r * k + c] = acc;
      }
    }
    return 0;
  }
  // Dense on a (value, tangent) pair, rows stacked [2M] (the device runs these as one launch, spk_dense_dual_f32)
  int dense_dual(const T* x2, const T* w, const T* b, T* y2, T* pre2, int64_t M, int k, int n_out, int act, const T* fc, const T* fc1) {
    for (int64_t r = 0; r < M; ++r)
      for (int o = 0; o < n_out; ++o) {
        T pv = b ? b[o] : T(0), pt = 0;
        for (int c = 0; c < k; ++c) {
          pv += x2[r * k + c] * w[(int64_t)o * k + c];
          pt += x2[(M + r) * k + c] * w[(int64_t)o * k + c];
        }
        if (pre2) { pre2[r * n_out + o] = pv; pre2[(M + r) * n_out + o] = pt; }
        if (fc) {
          y2[r * n_out + o] = pv * fc[r];
          y2[(M + r) * n_out + o] = pt * fc[r] + pv * fc1[r];
        } else {
          y2[r * n_out + o] = fm_act(act, 0, pv);
          y2[(M + r) * n_out + o] = fm_act(act, 1, pv) * pt;
        }
      }
    return 0;
  }
  // yt = act'(pre_v) (xt W^T)  (trans: xt W with W [KC, NW]);  pre_t receives the product
  int dense_tangent(const T* xt, const T* w, const T* pre_v, T* yt, T* pre_t, int64_t M, int KC, int NW, int act, bool trans) {
    for (int64_t r = 0; r < M; ++r)
      for (int o = 0; o < NW; ++o) {
        T pt = 0;
        for (int c = 0; c < KC; ++c) pt += xt[r * KC + c] * (trans ? w[(int64_t)c * NW + o] : w[(int64_t)o * KC + c]);
        pre_t[r * NW + o] = pt;
        yt[r * NW + o] = fm_act(act, 1, pre_v[r * NW + o]) * pt;
      }
    return 0;
  }
  // [g_z ; h_z] = g2 W  (g2 [2M, n_out], W [n_out, k]), then the reverse of the activation pair at pre2 = [a ; a_t] ([2M, k])
  int dense_dual_bwd(const T* g2, const T* w, const T* pre2, T* gx2, T* tmp2, int64_t M, int k, int n_out, int act) {
    (void)tmp2;
    for (int64_t r = 0; r < M; ++r)
      for (int c = 0; c < k; ++c) {
        T gz = 0, hz = 0;
        for (int o = 0; o < n_out; ++o) {
          gz += g2[r * n_out + o] * w[(int64_t)o * k + c];
          hz += g2[(M + r) * n_out + o] * w[(int64_t)o * k + c];
        }
        const T a = pre2[r * k + c], at = pre2[(M + r) * k + c], a1 = fm_act(act, 1, a);
        gx2[r * k + c] = gz * a1 + hz * fm_act(act, 2, a) * at;
        gx2[(M + r) * k + c] = hz * a1;
      }
    return 0;
  }
  // the weight-gradient GEMMs are DEFERRED to gemm_flush() exactly like on the device (one batched launch at the end of a pass): an operand
  // buffer that the engine overwrites before the flush would give wrong gradients here too
  struct Tn { const T* U; const T* X; int64_t n; int O, K; T* G; T* gb; int64_t nb; };
  std::vector<Tn> pending;
  int gemm_tn(const T* U, const T* X, int64_t n, int O, int K, T* G, T* gb, int64_t n_bias) {
    pending.push_back(Tn{U, X, n, O, K, G, gb, n_bias});
    return 0;
  }
  int gemm_flush() {
    for (const Tn& t : pending) gemm_tn_now(t.U, t.X, t.n, t.O, t.K, t.G, t.gb, t.nb);
    pending.clear();
    return 0;
  }
  // G [O, K] = U^T X over n rows; gb [O] = column sums of the first n_bias rows of U
  int gemm_tn_now(const T* U, const T* X, int64_t n, int O, int K, T* G, T* gb, int64_t n_bias) {
    for (int o = 0; o < O; ++o) {
      for (int c = 0; c < K; ++c) {
        T acc = 0;
        for (int64_t r = 0; r < n; ++r) acc += U[r * O + o] * X[r * K + c];
        G[(int64_t)o * K + c] = acc;
      }
      if (gb) {
        T s = 0;
        for (int64_t r = 0; r < n_bias; ++r) s += U[r * O + o];
        gb[o] = s;
      }
    }
    return 0;
  }
  template <class... KA, class... A>
  void flat(const char*, void (*k)(KA...), int64_t, A... a) { k(static_cast<KA>(a)...); }
  template <class... KA, class... A>
  void rows(const char*, void (*k)(KA...), int64_t, A... a) { k(static_cast<KA>(a)...); }
  template <class... KA, class... A>
  void slotted(const char*, void (*k)(KA...), int64_t, A... a) { k(static_cast<KA>(a)...); }
};

extern "C" {
struct EmuDesc {
  int32_t kind, F, nf, L, K, H, head_act, rbf_kind, shared, n_types;
  double cutoff, eps;
  int64_t N, E, M;
};
}

template <class T>
struct Bound {
  std::vector<FmSchnetLayer<T>> sl;
  std::vector<FmPainnLayer<T>> pl;
  FmSchnetModel<T> sm;
  FmPainnModel<T> pm;
  FmHead<T> hd;
  FmRadial<T> rb;
  FmBatch<T> b;
};

template <class T>
static void bind(const EmuDesc* d, const void** wv, const int64_t* Z, const int64_t* ii, const int64_t* jj, const int64_t* idx_m, const void* R, const void* off,
                 Bound<T>& o) {
  const T** w = (const T**)wv;
  int p = 0;
  if (d->kind == 0) {
    o.sl.resize((size_t)d->L);
    for (int l = 0; l < d->L; ++l) {
      FmSchnetLayer<T>& s = o.sl[(size_t)l];
      s.in2f_w = w[p++]; s.fn_w1 = w[p++]; s.fn_b1 = w[p++]; s.fn_w2 = w[p++]; s.fn_b2 = w[p++];
      s.f2out_w1 = w[p++]; s.f2out_b1 = w[p++]; s.f2out_w2 = w[p++]; s.f2out_b2 = w[p++];
    }
    o.sm = FmSchnetModel<T>{d->F, d->nf, d->L, o.sl.data()};
  } else {
    o.pl.resize((size_t)d->L);
    for (int l = 0; l < d->L; ++l) {
      FmPainnLayer<T>& s = o.pl[(size_t)l];
      s.ctx_w1 = w[p++]; s.ctx_b1 = w[p++]; s.ctx_w2 = w[p++]; s.ctx_b2 = w[p++]; s.mix_w = w[p++];
      s.ictx_w1 = w[p++]; s.ictx_b1 = w[p++]; s.ictx_w2 = w[p++]; s.ictx_b2 = w[p++];
    }
    o.pm = FmPainnModel<T>{d->F, d->L, d->shared, (T)d->eps, o.pl.data(), nullptr, nullptr};
    o.pm.filt_w = w[p++]; o.pm.filt_b = w[p++];
  }
  o.hd.w1 = w[p++]; o.hd.b1 = w[p++]; o.hd.w2 = w[p++]; o.hd.b2 = w[p++];
  o.hd.n_hidden = d->H; o.hd.act = d->head_act;
  o.b.emb = w[p++];
  o.rb.kind = d->rbf_kind; o.rb.n_rbf = d->K; o.rb.p0 = w[p++]; o.rb.p1 = w[p++]; o.rb.cutoff = (T)d->cutoff;
  o.b.N = d->N; o.b.E = d->E; o.b.M = d->M; o.b.Z = Z; o.b.ii = ii; o.b.jj = jj; o.b.idx_m = idx_m; o.b.R = (const T*)R; o.b.off = (const T*)off;
  o.b.n_types = d->n_types;
}

template <class T>
static int64_t ws_bytes(const EmuDesc* d) {
  EmuBackend<T> be;
  FmEngine<T, EmuBackend<T>> eng(be);
  if (d->kind == 0) {
    typename FmEngine<T, EmuBackend<T>>::SchnetWs w;
    FmSchnetModel<T> m{d->F, d->nf, d->L, nullptr};
    eng.schnet_carve(nullptr, m, d->K, d->H, d->N, d->E, d->M, d->n_types, w);
    return (int64_t)w.bytes;
  }
  typename FmEngine<T, EmuBackend<T>>::PainnWs w;
  FmPainnModel<T> m{d->F, d->L, d->shared, (T)d->eps, nullptr, nullptr, nullptr};
  eng.painn_carve(nullptr, m, d->K, d->H, d->N, d->E, d->M, d->n_types, w);
  return (int64_t)w.bytes;
}

template <class T>
static int run(const EmuDesc* d, int backward, const void** wv, const int64_t* Z, const int64_t* ii, const int64_t* jj, const int64_t* idx_m, const void* R,
               const void* off, void* ws, void* a0, void* a1, void* a2) {
  Bound<T> o;
  bind<T>(d, wv, Z, ii, jj, idx_m, R, off, o);
  EmuBackend<T> be;
  FmEngine<T, EmuBackend<T>> eng(be);
  int32_t err = 0;
  int rc;
  if (d->kind == 0)
    rc = backward ? eng.schnet_backward(o.sm, o.hd, o.rb, o.b, ws, (const T*)a0, (const T*)a1, (T*)a2) : eng.schnet_forward(o.sm, o.hd, o.rb, o.b, ws, (T*)a0, (T*)a1, &err);
  else
    rc = backward ? eng.painn_backward(o.pm, o.hd, o.rb, o.b, ws, (const T*)a0, (const T*)a1, (T*)a2) : eng.painn_forward(o.pm, o.hd, o.rb, o.b, ws, (T*)a0, (T*)a1, &err);
  return rc ? rc : -err;
}

extern "C" {
int64_t fm_emu_ws_bytes(const EmuDesc* d, int f64) { return f64 ? ws_bytes<double>(d) : ws_bytes<float>(d); }
int64_t fm_emu_grad_floats(const EmuDesc* d) {
  return d->kind == 0 ? fm_schnet_grad_floats(d->F, d->nf, d->L, d->K, d->H, d->n_types) : fm_painn_grad_floats(d->F, d->L, d->K, d->H, d->n_types, d->shared);
}
// forward: a0 = E_out [M], a1 = F_out [N,3];  backward: a0 = gE [M], a1 = gF [N,3], a2 = grads (flat)
int fm_emu_forward(const EmuDesc* d, int f64, const void** w, const int64_t* Z, const int64_t* ii, const int64_t* jj, const int64_t* idx_m, const void* R, const void* off,
                   void* ws, void* E_out, void* F_out) {
  return f64 ? run<double>(d, 0, w, Z, ii, jj, idx_m, R, off, ws, E_out, F_out, nullptr) : run<float>(d, 0, w, Z, ii, jj, idx_m, R, off, ws, E_out, F_out, nullptr);
}
int fm_emu_backward(const EmuDesc* d, int f64, const void** w, const int64_t* Z, const int64_t* ii, const int64_t* jj, const int64_t* idx_m, const void* R, const void* off,
                    void* ws, const void* gE, const void* gF, void* grads) {
  return f64 ? run<double>(d, 1, w, Z, ii, jj, idx_m, R, off, ws, (void*)gE, (void*)gF, grads) : run<float>(d, 1, w, Z, ii, jj, idx_m, R, off, ws, (void*)gE, (void*)gF, grads);
}
}
