// Operators of the training regime (included by spk_torch.cpp inside its anonymous namespace).
//
// Force matching differentiates every operator twice (atomistic/response.py:59-68, create_graph = training).  The operators
// below wrap the kernel family of spk_train.hip, which is closed under differentiation: each backward is written in terms of
// the same differentiable operators, so the recorded backward pass and the backward of THAT pass are again these HIP launches.
// Argument order of every autograd Function: required tensors first, optional tensors next, plain values last -- the
// needs_input_grad() index of a required tensor is then its position (optional tensors that are absent take no slot).

// ------------------------------------------------------------------------------------------------ raw launchers
Tensor act_mul_raw(const OptT& a_in, const Tensor& z_in, int64_t act, int64_t order, const OptT& c_in) {
  Tensor z = f32(z_in, "act_mul"), a = opt_f32(a_in, "act_mul"), c = opt_f32(c_in, "act_mul");
  TORCH_CHECK(!a.defined() || a.sizes() == z.sizes(), "act_mul: a ", a.sizes(), " and z ", z.sizes(), " differ in shape");
  TORCH_CHECK(!c.defined() || c.sizes() == z.sizes(), "act_mul: c ", c.sizes(), " and z ", z.sizes(), " differ in shape");
  c10::DeviceGuard guard(z.device());
  Tensor out = at::empty_like(z);
  check(spk_act_mul_f32(fp(a), fp(z), fp(c), z.numel(), (int32_t)act, (int32_t)order, fpm(out), stream_of(z)));
  return out;
}

Tensor linear_raw(const Tensor& x, const Tensor& w, const OptT& b) { return std::get<0>(dense_raw(x, w, b, SPK_ACT_NONE)); }
Tensor matmul_nn_raw(const Tensor& u, const Tensor& w) {
  TORCH_CHECK(w.dim() == 2 && u.size(-1) == w.size(0), "matmul_nn: u ", u.sizes(), " and weight ", w.sizes(), " do not match");
  return dense_bwd_input_raw(u, Tensor(), w, SPK_ACT_NONE);
}

// ticket counters of the split contraction (zero between launches: the kernel resets them).  One buffer per (device, stream):
// launches on one stream are ordered, launches on different streams must not share counters.
std::vector<std::pair<std::pair<int64_t, void*>, Tensor>> g_tn_tickets;
uint32_t* tn_tickets(const Tensor& like) {
  const std::pair<int64_t, void*> key{(int64_t)like.device().index(), stream_of(like)};
  std::lock_guard<std::mutex> lock(g_mutex);
  for (auto& e : g_tn_tickets)
    if (e.first == key) return (uint32_t*)e.second.data_ptr<int32_t>();
  if (g_tn_tickets.size() >= 16) g_tn_tickets.erase(g_tn_tickets.begin());
  g_tn_tickets.emplace_back(key, at::zeros({4096}, like.options().dtype(at::kInt)));
  return (uint32_t*)g_tn_tickets.back().second.data_ptr<int32_t>();
}
std::tuple<Tensor, Tensor> matmul_tn_raw(const Tensor& u_in, const Tensor& x_in) {
  Tensor u = f32(u_in, "matmul_tn"), x = f32(x_in, "matmul_tn");
  const int64_t O = u.size(-1), K = x.size(-1);
  TORCH_CHECK(O > 0 && K > 0, "matmul_tn: empty feature dimension");
  const int64_t n = u.numel() / O;
  TORCH_CHECK(x.numel() / K == n, "matmul_tn: u ", u.sizes(), " and x ", x.sizes(), " differ in their leading dimensions");
  c10::DeviceGuard guard(u.device());
  Tensor G = at::empty({O, K}, u.options()), cs = at::empty({O}, u.options());
  int32_t S = 1, tiles = 0;
  int64_t wsf = 0;
  check(spk_gemm_tn_plan(n, (int32_t)O, (int32_t)K, &S, &wsf, &tiles));
  Tensor ws;
  if (S > 1) ws = at::empty({wsf}, u.options());
  check(spk_gemm_tn_f32(fp(u), fp(x), n, (int32_t)O, (int32_t)K, fpm(G), fpm(cs), fpm(ws), S > 1 ? tn_tickets(u) : nullptr, stream_of(u)));
  return {G, cs};
}

// out = a w (trans) or a w^T, and (G, cs) = (u^T x, column sums of u): the two independent products of a Dense backward in ONE
// launch where the shapes allow it (no autograd: used by backward passes that are not themselves recorded)
std::tuple<Tensor, Tensor, Tensor> gemm_pair_raw(const Tensor& a_in, const Tensor& w_in, bool trans, const Tensor& u_in, const Tensor& x_in) {
  Tensor a = f32(a_in, "gemm_pair"), w = f32(w_in, "gemm_pair"), u = f32(u_in, "gemm_pair"), x = f32(x_in, "gemm_pair");
  TORCH_CHECK(w.dim() == 2 && a.size(-1) == (trans ? w.size(0) : w.size(1)), "gemm_pair: a ", a.sizes(), " and weight ", w.sizes(), " do not match");
  const int64_t n_out = w.size(0), k = w.size(1), KC = trans ? n_out : k, NW = trans ? k : n_out;
  const int64_t O = u.size(-1), K = x.size(-1), n = O > 0 ? u.numel() / O : 0;
  const int64_t m = KC > 0 ? a.numel() / KC : 0;
  // (long contractions with few output tiles are better off in the split-K Dense kernel of their own)
  const bool long_k = KC >= 256 && ((m + 31) / 32) * ((NW + 31) / 32) <= 512;
  if (KC % 4 != 0 || NW % 4 != 0 || m == 0 || n == 0 || long_k) {
    Tensor out = trans ? matmul_nn_raw(a, w) : linear_raw(a, w, c10::nullopt);
    auto r = matmul_tn_raw(u, x);
    return {out, std::get<0>(r), std::get<1>(r)};
  }
  TORCH_CHECK(x.numel() / K == n, "gemm_pair: u ", u.sizes(), " and x ", x.sizes(), " differ in their leading dimensions");
  c10::DeviceGuard guard(a.device());
  auto shape = a.sizes().vec();
  shape.back() = NW;
  Tensor out = at::empty(shape, a.options()), G = at::empty({O, K}, a.options()), cs = at::empty({O}, a.options());
  int32_t S = 1, tiles = 0;
  int64_t wsf = 0;
  check(spk_gemm_tn_plan(n, (int32_t)O, (int32_t)K, &S, &wsf, &tiles));
  Tensor ws;
  if (S > 1) ws = at::empty({wsf}, a.options());
  check(spk_gemm_pair_f32(fp(a), fp(w), trans ? 1 : 0, m, (int32_t)k, (int32_t)n_out, fpm(out), fp(u), fp(x), n, (int32_t)O, (int32_t)K, fpm(G), fpm(cs),
                          fpm(ws), S > 1 ? tn_tickets(a) : nullptr, stream_of(a)));
  return {out, G, cs};
}

bool has(const OptT& t) { return t.has_value() && t->defined(); }

Tensor edge_mul_raw(const Tensor& a_in, const Tensor& b_in, const OptT& ia_in, const OptT& ib_in) {
  Tensor a = f32(a_in, "edge_mul"), b = f32(b_in, "edge_mul");
  TORCH_CHECK(has(ia_in) || has(ib_in), "edge_mul: at least one index is required (use a plain product otherwise)");
  Tensor ia = has(ia_in) ? i64(*ia_in, "edge_mul") : Tensor(), ib = has(ib_in) ? i64(*ib_in, "edge_mul") : Tensor();
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && a.size(1) == b.size(1), "edge_mul: a ", a.sizes(), " and b ", b.sizes(), " must share the feature dimension");
  const int64_t E = ia.defined() ? ia.size(0) : ib.size(0);
  TORCH_CHECK((!ia.defined() || (ia.dim() == 1 && ia.size(0) == E)) && (!ib.defined() || (ib.dim() == 1 && ib.size(0) == E)), "edge_mul: index tensors differ in length");
  TORCH_CHECK(ia.defined() || a.size(0) == E, "edge_mul: a without an index must have one row per pair");
  TORCH_CHECK(ib.defined() || b.size(0) == E, "edge_mul: b without an index must have one row per pair");
  c10::DeviceGuard guard(a.device());
  Tensor out = at::empty({E, a.size(1)}, a.options());
  check(spk_edge_mul_f32(fp(a), fp(b), ia.defined() ? ia.data_ptr<int64_t>() : nullptr, ib.defined() ? ib.data_ptr<int64_t>() : nullptr, E, a.size(0), b.size(0),
                         (int32_t)a.size(1), fpm(out), stream_of(a)));
  return out;
}

Tensor cfconv_raw(const Tensor& x_in, const Tensor& W_in, const OptT& idx_out_in, const OptT& idx_src_in, int64_t n_out) {
  if (!has(idx_out_in)) {      // one output row per pair: y[e] = x[src_e] . W_e
    TORCH_CHECK(n_out == W_in.size(0), "cfconv: without an output index n_out must be the number of pairs");
    return edge_mul_raw(W_in, x_in, c10::nullopt, idx_src_in);
  }
  Tensor x = f32(x_in, "cfconv"), W = f32(W_in, "cfconv");
  Tensor io = i64(*idx_out_in, "cfconv"), is = has(idx_src_in) ? i64(*idx_src_in, "cfconv") : Tensor();
  TORCH_CHECK(x.dim() == 2 && W.dim() == 2 && x.size(1) == W.size(1), "cfconv: x ", x.sizes(), " and W ", W.sizes(), " must be [n, F] and [E, F]");
  const int64_t E = W.size(0), F = W.size(1);
  TORCH_CHECK(io.dim() == 1 && io.size(0) == E && (!is.defined() || (is.dim() == 1 && is.size(0) == E)), "cfconv: index tensors must have ", E, " entries");
  TORCH_CHECK(is.defined() || x.size(0) == E, "cfconv: x without a source index must have one row per pair");
  c10::DeviceGuard guard(x.device());
  Tensor y = at::empty({n_out, F}, x.options());
  Tensor rp = E > 0 ? segment_rowptr(idx_out_in->scalar_type() == at::kLong && idx_out_in->is_contiguous() ? *idx_out_in : io, n_out) : Tensor();
  if (E == 0) y.zero_();
  check(spk_cfconv_edge_f32(fp(x), fp(W), E ? io.data_ptr<int64_t>() : nullptr, (E && is.defined()) ? is.data_ptr<int64_t>() : nullptr,
                            rp.defined() ? rp.data_ptr<int32_t>() : nullptr, E, n_out, x.size(0), (int32_t)F, fpm(y), stream_of(x)));
  return y;
}

// ---- 3-vector algebra: operands are read in place through a row stride where their layout allows it
struct Rows { Tensor keep; const float* p; int64_t rows, width, ld; };
Rows rows_of(const Tensor& x, const char* who) {
  require_device(x, who);
  TORCH_CHECK(x.scalar_type() == at::kFloat, who, ": dtype ", x.scalar_type(), " unsupported; the HIP path computes in float32");
  TORCH_CHECK(x.dim() >= 1 && x.size(-1) > 0, who, ": empty feature dimension");
  const int64_t width = x.size(-1), rows = x.numel() / width;
  bool ok = x.stride(-1) == 1 || width == 1;
  int64_t ld = -1, expected = -1;
  for (int64_t d = x.dim() - 2; ok && d >= 0; --d) {
    if (x.size(d) == 1) continue;
    if (ld < 0) { ld = x.stride(d); expected = ld * x.size(d); }
    else if (x.stride(d) != expected) ok = false;
    else expected *= x.size(d);
  }
  if (ld < 0) ld = width;
  if (!ok || ld < width || rows == 0) {
    Tensor c = x.contiguous();
    return {c, c.data_ptr<float>(), rows, width, width};
  }
  return {x, x.data_ptr<float>(), rows, width, ld};
}

Tensor vec3_raw(int64_t op, const Tensor& A_in, const Tensor& B_in) {
  static const char* names[] = {"vscale", "vdot", "vouter", "vcontract", "vrowdot"};
  TORCH_CHECK(op >= 0 && op <= 4, "vec3: unknown operation ", op);
  const char* who = names[op];
  const bool a_is_v = op != SPK_VEC3_OUTER;                       // A: V-type except for OUTER (s-type)
  const bool b_is_u = op == SPK_VEC3_OUTER || op == SPK_VEC3_CONTRACT;
  Rows A = rows_of(A_in, who);
  Tensor Bu = b_is_u ? f32(B_in, who) : Tensor();
  Rows B = b_is_u ? Rows{Bu, Bu.data_ptr<float>(), Bu.numel() / 3, 3, 3} : rows_of(B_in, who);
  const int64_t F = A.width;
  const int64_t M = a_is_v ? A.rows / 3 : A.rows;
  TORCH_CHECK(!a_is_v || (A_in.dim() >= 2 && A_in.size(-2) == 3 && A.rows == 3 * M), who, ": expected a [..., 3, F] operand, got ", A_in.sizes());
  if (b_is_u) {
    TORCH_CHECK(Bu.numel() == 3 * M && Bu.size(-1) == 3, who, ": expected a [..., 3] operand with ", M, " rows, got ", B_in.sizes());
  } else if (op == SPK_VEC3_DOT) {
    TORCH_CHECK(B.width == F && B.rows == 3 * M && B_in.dim() >= 2 && B_in.size(-2) == 3, who, ": operands ", A_in.sizes(), " and ", B_in.sizes(), " differ");
  } else {
    TORCH_CHECK(B.width == F && B.rows == M, who, ": expected one [F] row per vector, got ", B_in.sizes(), " for ", A_in.sizes());
  }
  c10::DeviceGuard guard(A_in.device());
  Tensor out;
  auto lead = [&](const Tensor& t, int64_t drop) { auto v = t.sizes().vec(); v.resize(v.size() - drop); return v; };
  if (op == SPK_VEC3_SCALE) out = at::empty(A_in.sizes(), A_in.options().memory_format(at::MemoryFormat::Contiguous));
  else if (op == SPK_VEC3_DOT || op == SPK_VEC3_CONTRACT) { auto v = lead(A_in, 2); v.push_back(1); v.push_back(F); out = at::empty(v, A_in.options()); }
  else if (op == SPK_VEC3_OUTER) { auto v = lead(B_in, 1); v.push_back(3); v.push_back(F); out = at::empty(v, A_in.options()); }
  else { auto v = lead(A_in, 1); out = at::empty(v, A_in.options()); }
  check(spk_vec3_f32((int32_t)op, A.p, A.ld, B.p, B.ld, M, (int32_t)F, fpm(out), stream_of(A_in)));
  return out;
}

spk_radial_t radial_k_of(int64_t kind, const Tensor& p0, const Tensor& p1, double cutoff) {
  spk_radial_t rb = radial_of(kind, p0, p1, cutoff);
  if (kind == 2) rb.n_rbf = 1;
  return rb;
}

Tensor radial_d_raw(const Tensor& d_in, const OptT& a_in, int64_t kind, const Tensor& p0_in, const OptT& p1_in, double cutoff, int64_t order) {
  Tensor d = f32(d_in, "radial_d"), a = opt_f32(a_in, "radial_d");
  Tensor p0 = f32(p0_in, "radial_d"), p1 = opt_f32(p1_in, "radial_d");
  TORCH_CHECK(!a.defined() || a.sizes() == d.sizes(), "radial_d: a ", a.sizes(), " and d ", d.sizes(), " differ in shape");
  c10::DeviceGuard guard(d.device());
  auto shape = d.sizes().vec();
  if (kind != 2) shape.push_back(p0.size(0));
  Tensor out = at::empty(shape, d.options());
  spk_radial_t rb = radial_k_of(kind, p0, p1, cutoff);
  check(spk_radial_d_f32(fp(d), fp(a), d.numel(), &rb, (int32_t)order, fpm(out), stream_of(d)));
  return out;
}

Tensor radial_c_raw(const Tensor& G_in, const Tensor& d_in, const OptT& a_in, int64_t kind, const Tensor& p0_in, const OptT& p1_in, double cutoff,
                    int64_t order) {
  Tensor G = f32(G_in, "radial_c"), d = f32(d_in, "radial_c"), a = opt_f32(a_in, "radial_c");
  Tensor p0 = f32(p0_in, "radial_c"), p1 = opt_f32(p1_in, "radial_c");
  TORCH_CHECK(kind == 0 || kind == 1, "radial_c: kind ", kind, " has no basis dimension to contract");
  TORCH_CHECK(G.dim() == d.dim() + 1 && G.size(-1) == p0.size(0) && G.numel() == d.numel() * p0.size(0), "radial_c: G ", G.sizes(), " does not match d ",
              d.sizes(), " x n_rbf ", p0.size(0));
  TORCH_CHECK(!a.defined() || a.sizes() == d.sizes(), "radial_c: a ", a.sizes(), " and d ", d.sizes(), " differ in shape");
  c10::DeviceGuard guard(d.device());
  Tensor out = at::empty_like(d);
  spk_radial_t rb = radial_k_of(kind, p0, p1, cutoff);
  check(spk_radial_c_f32(fp(G), fp(d), fp(a), d.numel(), &rb, (int32_t)order, fpm(out), stream_of(d)));
  return out;
}

Tensor rowscale_raw(const Tensor& W_in, const Tensor& s_in) {
  Tensor W = f32(W_in, "rowscale"), s = f32(s_in, "rowscale");
  TORCH_CHECK(W.dim() >= 1 && W.size(-1) > 0 && s.numel() * W.size(-1) == W.numel(), "rowscale: W ", W.sizes(), " needs one scale per row, got ", s.sizes());
  c10::DeviceGuard guard(W.device());
  Tensor out = at::empty_like(W);
  check(spk_rowscale_f32(fp(W), fp(s), s.numel(), (int32_t)W.size(-1), fpm(out), stream_of(W)));
  return out;
}

Tensor rowdot_raw(const Tensor& a_in, const Tensor& b_in) {
  Tensor a = f32(a_in, "rowdot"), b = f32(b_in, "rowdot");
  TORCH_CHECK(a.sizes() == b.sizes() && a.dim() >= 1 && a.size(-1) > 0, "rowdot: a ", a.sizes(), " and b ", b.sizes(), " differ in shape");
  c10::DeviceGuard guard(a.device());
  auto shape = a.sizes().vec();
  shape.pop_back();
  Tensor out = at::empty(shape, a.options());
  check(spk_rowdot_f32(fp(a), fp(b), out.numel(), (int32_t)a.size(-1), fpm(out), stream_of(a)));
  return out;
}

Tensor edge_norm_raw(const Tensor& r_in) {
  Tensor r = f32(r_in, "edge_norm");
  TORCH_CHECK(r.dim() == 2 && r.size(1) == 3, "edge_norm: r_ij must be [E, 3], got ", r.sizes());
  c10::DeviceGuard guard(r.device());
  Tensor d = at::empty({r.size(0)}, r.options());
  check(spk_edge_norm_f32(fp(r), r.size(0), fpm(d), nullptr, stream_of(r)));
  return d;
}

// ------------------------------------------------------------------------------------------------ dispatcher handles
Tensor call_act_mul(const OptT& a, const Tensor& z, int64_t act, int64_t order, const OptT& c) {
  static auto op = op_handle<Tensor(const OptT&, const Tensor&, int64_t, int64_t, const OptT&)>("spk_hip::act_mul");
  return op.call(a, z, act, order, c);
}
Tensor call_linear(const Tensor& x, const Tensor& w, const OptT& b) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&, const OptT&)>("spk_hip::linear");
  return op.call(x, w, b);
}
Tensor call_matmul_nn(const Tensor& u, const Tensor& w) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&)>("spk_hip::matmul_nn");
  return op.call(u, w);
}
std::tuple<Tensor, Tensor> call_matmul_tn(const Tensor& u, const Tensor& x) {
  static auto op = op_handle<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&)>("spk_hip::matmul_tn");
  return op.call(u, x);
}
std::tuple<Tensor, Tensor, Tensor> call_gemm_pair(const Tensor& a, const Tensor& w, bool trans, const Tensor& u, const Tensor& x) {
  static auto op = op_handle<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, bool, const Tensor&, const Tensor&)>("spk_hip::gemm_pair");
  return op.call(a, w, trans, u, x);
}
Tensor call_cfconv(const Tensor& x, const Tensor& W, const OptT& io, const OptT& is, int64_t n_out) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&, const OptT&, const OptT&, int64_t)>("spk_hip::cfconv");
  return op.call(x, W, io, is, n_out);
}
Tensor call_edge_mul(const Tensor& a, const Tensor& b, const OptT& ia, const OptT& ib) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&, const OptT&, const OptT&)>("spk_hip::edge_mul");
  return op.call(a, b, ia, ib);
}
Tensor call_vec3(int64_t opc, const Tensor& A, const Tensor& B) {
  static auto op = op_handle<Tensor(int64_t, const Tensor&, const Tensor&)>("spk_hip::vec3");
  return op.call(opc, A, B);
}
Tensor call_radial_d(const Tensor& d, const OptT& a, int64_t kind, const Tensor& p0, const OptT& p1, double cutoff, int64_t order) {
  static auto op = op_handle<Tensor(const Tensor&, const OptT&, int64_t, const Tensor&, const OptT&, double, int64_t)>("spk_hip::radial_d");
  return op.call(d, a, kind, p0, p1, cutoff, order);
}
Tensor call_radial_c(const Tensor& G, const Tensor& d, const OptT& a, int64_t kind, const Tensor& p0, const OptT& p1, double cutoff, int64_t order) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&, const OptT&, int64_t, const Tensor&, const OptT&, double, int64_t)>("spk_hip::radial_c");
  return op.call(G, d, a, kind, p0, p1, cutoff, order);
}
Tensor call_rowscale(const Tensor& W, const Tensor& s) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&)>("spk_hip::rowscale");
  return op.call(W, s);
}
// loss = wE mean((E - E_t)^2) + wF mean((F - F_t)^2) -> (loss [], gE, gF): value and gradients in one launch
std::tuple<Tensor, Tensor, Tensor> fm_loss_raw(const Tensor& E_in, const Tensor& Et_in, const Tensor& F_in, const Tensor& Ft_in, double wE, double wF) {
  Tensor E = f32(E_in, "fm_loss"), Et = f32(Et_in, "fm_loss"), F = f32(F_in, "fm_loss"), Ft = f32(Ft_in, "fm_loss");
  TORCH_CHECK(E.sizes() == Et.sizes() && F.sizes() == Ft.sizes(), "fm_loss: predictions and targets differ in shape");
  c10::DeviceGuard guard(E.device());
  Tensor loss = at::empty({}, E.options()), gE = at::empty_like(E), gF = at::empty_like(F);
  check(spk_fm_loss_f32(fp(E), fp(Et), E.numel(), fp(F), fp(Ft), F.numel(), (float)wE, (float)wF, fpm(loss), fpm(gE), fpm(gF), stream_of(E)));
  return {loss, gE, gF};
}
std::tuple<Tensor, Tensor> fm_loss_backward_raw(const Tensor& g_in, const Tensor& gE, const Tensor& gF) {
  Tensor g = f32(g_in, "fm_loss backward").reshape({1});
  c10::DeviceGuard guard(gE.device());
  Tensor oE = at::empty_like(gE), oF = at::empty_like(gF);
  check(spk_fm_loss_bwd_f32(fp(g), fp(gE), gE.numel(), fp(gF), gF.numel(), fpm(oE), fpm(oF), stream_of(gE)));
  return {oE, oF};
}
Tensor call_rowdot(const Tensor& a, const Tensor& b) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&)>("spk_hip::rowdot");
  return op.call(a, b);
}
Tensor call_edge_norm(const Tensor& r) {
  static auto op = op_handle<Tensor(const Tensor&)>("spk_hip::edge_norm");
  return op.call(r);
}

// ------------------------------------------------------------------------------------------------ autograd
// out = a . act^(k)(z) + c
struct ActMulFn : public torch::autograd::Function<ActMulFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& z, const OptT& a, const OptT& c, int64_t act, int64_t order) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const bool has_a = a.has_value() && a->defined(), has_c = c.has_value() && c->defined();
    ctx->save_for_backward({z, has_a ? *a : Tensor()});
    ctx->saved_data["cfg"] = std::vector<int64_t>{act, order, has_a, has_c};
    return call_act_mul(a, z, act, order, c);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    auto cfg = ctx->saved_data["cfg"].toIntVector();
    const Tensor &z = sv[0], &a = sv[1];
    const bool has_a = cfg[2] != 0, has_c = cfg[3] != 0;
    Tensor gz, ga, gc;
    if (ctx->needs_input_grad(0)) gz = call_act_mul(has_a ? OptT(at::mul(g[0], a)) : OptT(g[0]), z, cfg[0], cfg[1] + 1, c10::nullopt);
    if (has_a && ctx->needs_input_grad(1)) ga = call_act_mul(g[0], z, cfg[0], cfg[1], c10::nullopt);
    if (has_c && ctx->needs_input_grad(has_a ? 2 : 1)) gc = g[0];
    return {gz, ga, gc, Tensor(), Tensor()};
  }
};

struct LinearFn : public torch::autograd::Function<LinearFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w, const OptT& b) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({x, w});
    ctx->saved_data["has_bias"] = b.has_value() && b->defined();
    return call_linear(x, w, b);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    const bool need_b = ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(2);
    Tensor gx, gw, gb;
    if (!at::GradMode::is_enabled() && ctx->needs_input_grad(0) && (ctx->needs_input_grad(1) || need_b)) {
      auto r = call_gemm_pair(g[0], sv[1], true, g[0], sv[0]);      // g w and g^T x in one launch
      gx = std::get<0>(r);
      gw = std::get<1>(r);
      if (need_b) gb = std::get<2>(r);
      return {gx, gw, gb};
    }
    if (ctx->needs_input_grad(0)) gx = call_matmul_nn(g[0], sv[1]);
    if (ctx->needs_input_grad(1) || need_b) {
      auto r = call_matmul_tn(g[0], sv[0]);
      gw = std::get<0>(r);
      if (need_b) gb = std::get<1>(r);
    }
    return {gx, gw, gb};
  }
};

// out = u w   (u [..., O], w [O, K])
struct MatmulNNFn : public torch::autograd::Function<MatmulNNFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& u, const Tensor& w) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({u, w});
    return call_matmul_nn(u, w);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    Tensor gu, gw;
    if (!at::GradMode::is_enabled() && ctx->needs_input_grad(0) && ctx->needs_input_grad(1)) {
      auto r = call_gemm_pair(g[0], sv[1], false, sv[0], g[0]);     // g w^T and u^T g in one launch
      return {std::get<0>(r), std::get<1>(r)};
    }
    if (ctx->needs_input_grad(0)) gu = call_linear(g[0], sv[1], c10::nullopt);
    if (ctx->needs_input_grad(1)) gw = std::get<0>(call_matmul_tn(sv[0], g[0]));
    return {gu, gw};
  }
};

// (G, cs) = (u^T x, column sums of u)   over the flattened leading dimensions
struct MatmulTNFn : public torch::autograd::Function<MatmulTNFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& u, const Tensor& x) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({u, x});
    auto r = call_matmul_tn(u, x);
    return {std::get<0>(r), std::get<1>(r)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    const Tensor &u = sv[0], &x = sv[1];
    Tensor gu, gx;
    if (ctx->needs_input_grad(0)) {
      if (g[0].defined()) gu = call_linear(x, g[0], c10::nullopt).reshape(u.sizes());
      if (g[1].defined()) gu = gu.defined() ? gu + g[1] : g[1].expand_as(u);
    }
    if (ctx->needs_input_grad(1) && g[0].defined()) gx = call_matmul_nn(u, g[0]).reshape(x.sizes());
    return {gu, gx};
  }
};

// y[out_e] += x[src_e] . W_e   (an absent index is the identity)
struct CfconvFn : public torch::autograd::Function<CfconvFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& W, const OptT& io, const OptT& is, int64_t n_out) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({x, W, has(io) ? *io : Tensor(), has(is) ? *is : Tensor()});
    return call_cfconv(x, W, io, is, n_out);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    Tensor gx, gW;
    if (ctx->needs_input_grad(0)) gx = call_cfconv(g[0], sv[1], opt_of(sv[3]), opt_of(sv[2]), sv[0].size(0));
    if (ctx->needs_input_grad(1)) gW = call_edge_mul(g[0], sv[0], opt_of(sv[2]), opt_of(sv[3]));
    return {gx, gW, Tensor(), Tensor(), Tensor()};
  }
};

// out_e = a[ia_e] . b[ib_e]
struct EdgeMulFn : public torch::autograd::Function<EdgeMulFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& a, const Tensor& b, const OptT& ia, const OptT& ib) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({a, b, has(ia) ? *ia : Tensor(), has(ib) ? *ib : Tensor()});
    return call_edge_mul(a, b, ia, ib);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    Tensor ga, gb;
    if (ctx->needs_input_grad(0)) ga = call_cfconv(sv[1], g[0], opt_of(sv[2]), opt_of(sv[3]), sv[0].size(0));
    if (ctx->needs_input_grad(1)) gb = call_cfconv(sv[0], g[0], opt_of(sv[3]), opt_of(sv[2]), sv[1].size(0));
    return {ga, gb, Tensor(), Tensor()};
  }
};

// the five 3-vector products: each backward is two of the others
struct Vec3Fn : public torch::autograd::Function<Vec3Fn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& A, const Tensor& B, int64_t op) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({A, B});
    ctx->saved_data["op"] = op;
    return call_vec3(op, A, B);
  }
  static variable_list backward(AutogradContext* ctx, variable_list gl) {
    auto sv = ctx->get_saved_variables();
    const Tensor &A = sv[0], &B = sv[1], &g = gl[0];
    const int64_t op = ctx->saved_data["op"].toInt();
    const bool nA = ctx->needs_input_grad(0), nB = ctx->needs_input_grad(1);
    Tensor gA, gB;
    switch (op) {
      case SPK_VEC3_SCALE:      // V s
        if (nA) gA = call_vec3(SPK_VEC3_SCALE, g, B);
        if (nB) gB = call_vec3(SPK_VEC3_DOT, g, A).reshape(B.sizes());
        break;
      case SPK_VEC3_DOT:        // sum_k A B
        if (nA) gA = call_vec3(SPK_VEC3_SCALE, B, g);
        if (nB) gB = call_vec3(SPK_VEC3_SCALE, A, g);
        break;
      case SPK_VEC3_OUTER:      // s u_k
        if (nA) gA = call_vec3(SPK_VEC3_CONTRACT, g, B).reshape(A.sizes());
        if (nB) gB = call_vec3(SPK_VEC3_ROWDOT, g, A).reshape(B.sizes());
        break;
      case SPK_VEC3_CONTRACT:   // sum_k G u_k
        if (nA) gA = call_vec3(SPK_VEC3_OUTER, g, B).reshape(A.sizes());
        if (nB) gB = call_vec3(SPK_VEC3_ROWDOT, A, g).reshape(B.sizes());
        break;
      default:                  // sum_f G s
        if (nA) gA = call_vec3(SPK_VEC3_OUTER, B, g).reshape(A.sizes());
        if (nB) gB = call_vec3(SPK_VEC3_CONTRACT, A, g).reshape(B.sizes());
        break;
    }
    return {gA, gB, Tensor()};
  }
};

// Trainable radial parameters (nn/radial.py:36-45: GaussianRBF(trainable=True) makes `offsets` / `widths` nn.Parameters): the Gaussian
// family is closed under differentiation w.r.t. them as well.  With t = d - mu_k and phi_k^(n) the n-th derivative in d,
//   d phi_k^(n) / d mu_k = - phi_k^(n+1),      d phi_k^(n) / d w_k = - (n phi_k^(n) + t phi_k^(n+1)) / w_k      (phi(t; w) = g(t / w)),
// so the gradients are column sums over the pairs of the SAME products the d-gradients are made of -- expressed through the
// operators themselves (spk_hip::radial_d at order n, n + 1), hence differentiable again: the second order of force matching
// (atomistic/response.py:59-68) needs d/d(mu, w) of the recorded dE/dd nodes.  Bessel frequencies are buffers in the reference.
void check_fixed_basis(int64_t kind, const Tensor& p0, const OptT& p1, const char* who) {
  if (kind == 0) return;
  TORCH_CHECK(!p0.requires_grad() && !(p1.has_value() && p1->defined() && p1->requires_grad()), who,
              ": only the Gaussian basis has trainable parameters (offsets, widths); kind ", kind, " takes constants");
}
// column sums over every leading dimension: X [..., K] -> [K]
Tensor colsum_k(const Tensor& X) { return X.reshape({-1, X.size(-1)}).sum(0); }
// (g_mu, g_w) of  out = a phi^(n)(d)  contracted with `g` ([..., K]); Dn / Dn1 = a phi^(n) / a phi^(n+1) with the SAME a
std::pair<Tensor, Tensor> gaussian_param_grads(const Tensor& g, const Tensor& Dn, const Tensor& Dn1, const Tensor& d, const Tensor& mu, const Tensor& w,
                                               int64_t n, bool want_mu, bool want_w) {
  Tensor gmu, gw;
  const Tensor s1 = colsum_k(at::mul(g, Dn1));
  if (want_mu) gmu = at::neg(s1);
  if (want_w) {
    Tensor acc = at::sub(colsum_k(at::mul(at::mul(g, Dn1), d.unsqueeze(-1))), at::mul(mu, s1));
    if (n > 0) acc = at::add(acc, colsum_k(at::mul(g, Dn)), (double)n);
    gw = at::neg(at::div(acc, w));
  }
  return {gmu, gw};
}

// out = a phi^(k)(d)
struct RadialDFn : public torch::autograd::Function<RadialDFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& d, const Tensor& p0, const OptT& a, const OptT& p1, int64_t kind, double cutoff, int64_t order) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const bool has_a = a.has_value() && a->defined();
    ctx->save_for_backward({d, p0, has_a ? *a : Tensor(), (p1.has_value() && p1->defined()) ? *p1 : Tensor()});
    ctx->saved_data["cfg"] = std::vector<int64_t>{kind, order, has_a};
    ctx->saved_data["cutoff"] = cutoff;
    return call_radial_d(d, a, kind, p0, p1, cutoff, order);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    auto cfg = ctx->saved_data["cfg"].toIntVector();
    const double cutoff = ctx->saved_data["cutoff"].toDouble();
    const Tensor &d = sv[0], &p0 = sv[1], &a = sv[2];
    const OptT p1 = opt_of(sv[3]);
    const int64_t kind = cfg[0], k = cfg[1];
    const bool has_a = cfg[2] != 0;
    Tensor gd, ga;
    if (kind == 2) {
      if (ctx->needs_input_grad(0)) gd = call_radial_d(d, has_a ? OptT(at::mul(g[0], a)) : OptT(g[0]), kind, p0, p1, cutoff, k + 1);
      if (has_a && ctx->needs_input_grad(2)) ga = call_radial_d(d, g[0], kind, p0, p1, cutoff, k);
    } else {
      if (ctx->needs_input_grad(0)) gd = call_radial_c(g[0], d, opt_of(a), kind, p0, p1, cutoff, k + 1);
      if (has_a && ctx->needs_input_grad(2)) ga = call_radial_c(g[0], d, c10::nullopt, kind, p0, p1, cutoff, k);
    }
    Tensor gp0, gp1;
    // (needs_input_grad indexes the TENSOR inputs that were passed: an absent optional has no edge, so p1 follows a only if a is there)
    const size_t e_p1 = has_a ? 3 : 2;
    if (kind == 0 && sv[3].defined() && (ctx->needs_input_grad(1) || ctx->needs_input_grad(e_p1))) {
      TORCH_CHECK(k + 1 <= 3, "spk_hip::radial_d: parameter gradients of derivative order ", k, " are not provided");
      const Tensor Dn = call_radial_d(d, opt_of(a), kind, p0, p1, cutoff, k), Dn1 = call_radial_d(d, opt_of(a), kind, p0, p1, cutoff, k + 1);
      std::tie(gp0, gp1) = gaussian_param_grads(g[0], Dn, Dn1, d, p0, sv[3], k, ctx->needs_input_grad(1), ctx->needs_input_grad(e_p1));
    }
    return {gd, gp0, ga, gp1, Tensor(), Tensor(), Tensor()};
  }
};

// out = a sum_r G_r phi_r^(k)(d)
struct RadialCFn : public torch::autograd::Function<RadialCFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& G, const Tensor& d, const Tensor& p0, const OptT& a, const OptT& p1, int64_t kind, double cutoff,
                        int64_t order) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const bool has_a = a.has_value() && a->defined();
    ctx->save_for_backward({G, d, p0, has_a ? *a : Tensor(), (p1.has_value() && p1->defined()) ? *p1 : Tensor()});
    ctx->saved_data["cfg"] = std::vector<int64_t>{kind, order, has_a};
    ctx->saved_data["cutoff"] = cutoff;
    return call_radial_c(G, d, a, kind, p0, p1, cutoff, order);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    auto cfg = ctx->saved_data["cfg"].toIntVector();
    const double cutoff = ctx->saved_data["cutoff"].toDouble();
    const Tensor &G = sv[0], &d = sv[1], &p0 = sv[2], &a = sv[3];
    const OptT p1 = opt_of(sv[4]);
    const int64_t kind = cfg[0], k = cfg[1];
    const bool has_a = cfg[2] != 0;
    Tensor gG, gd, ga;
    Tensor ag = has_a ? at::mul(g[0], a) : g[0];
    if (ctx->needs_input_grad(0)) gG = call_radial_d(d, ag, kind, p0, p1, cutoff, k);
    if (ctx->needs_input_grad(1)) gd = call_radial_c(G, d, ag, kind, p0, p1, cutoff, k + 1);
    if (has_a && ctx->needs_input_grad(3)) ga = call_radial_c(G, d, g[0], kind, p0, p1, cutoff, k);
    Tensor gp0, gp1;
    const size_t e_p1 = has_a ? 4 : 3;
    if (kind == 0 && sv[4].defined() && (ctx->needs_input_grad(2) || ctx->needs_input_grad(e_p1))) {
      // out_e = (a g)_e sum_k G_ek phi_k^(n)(d_e): the same column sums with G in the place of the incoming gradient
      TORCH_CHECK(k + 1 <= 3, "spk_hip::radial_c: parameter gradients of derivative order ", k, " are not provided");
      const Tensor Dn = call_radial_d(d, ag, kind, p0, p1, cutoff, k), Dn1 = call_radial_d(d, ag, kind, p0, p1, cutoff, k + 1);
      std::tie(gp0, gp1) = gaussian_param_grads(G, Dn, Dn1, d, p0, sv[4], k, ctx->needs_input_grad(2), ctx->needs_input_grad(e_p1));
    }
    return {gG, gd, gp0, ga, gp1, Tensor(), Tensor(), Tensor()};
  }
};

struct RowscaleFn : public torch::autograd::Function<RowscaleFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& W, const Tensor& s) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({W, s});
    return call_rowscale(W, s);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    Tensor gW, gs;
    if (ctx->needs_input_grad(0)) gW = call_rowscale(g[0], sv[1]);
    if (ctx->needs_input_grad(1)) gs = call_rowdot(g[0], sv[0]).reshape(sv[1].sizes());
    return {gW, gs};
  }
};
struct RowdotFn : public torch::autograd::Function<RowdotFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& a, const Tensor& b) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({a, b});
    return call_rowdot(a, b);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    Tensor ga, gb;
    if (ctx->needs_input_grad(0)) ga = call_rowscale(sv[1], g[0]);
    if (ctx->needs_input_grad(1)) gb = call_rowscale(sv[0], g[0]);
    return {ga, gb};
  }
};

// d = |r_ij|; dd/dr = r / d with d the node's own (differentiable) output
struct EdgeNormFn : public torch::autograd::Function<EdgeNormFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& r) {
    at::AutoDispatchBelowADInplaceOrView guard;
    Tensor d = call_edge_norm(r);
    ctx->save_for_backward({r, d});
    return d;
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    return {call_rowscale(sv[0], at::div(g[0], sv[1]))};
  }
};

Tensor act_mul_ad(const OptT& a, const Tensor& z, int64_t act, int64_t order, const OptT& c) { return ActMulFn::apply(z, a, c, act, order); }
Tensor linear_ad(const Tensor& x, const Tensor& w, const OptT& b) { return LinearFn::apply(x, w, b); }
Tensor matmul_nn_ad(const Tensor& u, const Tensor& w) { return MatmulNNFn::apply(u, w); }
std::tuple<Tensor, Tensor> matmul_tn_ad(const Tensor& u, const Tensor& x) {
  auto r = MatmulTNFn::apply(u, x);
  return {r[0], r[1]};
}
Tensor cfconv_ad(const Tensor& x, const Tensor& W, const OptT& io, const OptT& is, int64_t n_out) { return CfconvFn::apply(x, W, io, is, n_out); }
Tensor edge_mul_ad(const Tensor& a, const Tensor& b, const OptT& ia, const OptT& ib) { return EdgeMulFn::apply(a, b, ia, ib); }
Tensor vec3_ad(int64_t op, const Tensor& A, const Tensor& B) { return Vec3Fn::apply(A, B, op); }
Tensor radial_d_ad(const Tensor& d, const OptT& a, int64_t kind, const Tensor& p0, const OptT& p1, double cutoff, int64_t order) {
  check_fixed_basis(kind, p0, p1, "spk_hip::radial_d");
  return RadialDFn::apply(d, p0, a, p1, kind, cutoff, order);
}
Tensor radial_c_ad(const Tensor& G, const Tensor& d, const OptT& a, int64_t kind, const Tensor& p0, const OptT& p1, double cutoff, int64_t order) {
  check_fixed_basis(kind, p0, p1, "spk_hip::radial_c");
  return RadialCFn::apply(G, d, p0, a, p1, kind, cutoff, order);
}
Tensor rowscale_ad(const Tensor& W, const Tensor& s) { return RowscaleFn::apply(W, s); }
// First order only: the gradient of the loss w.r.t. the forces flows on into the recorded force graph (that is where the second
// order of force matching lives); the loss node itself is not differentiated twice.
struct FmLossFn : public torch::autograd::Function<FmLossFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& E, const Tensor& Et, const Tensor& F, const Tensor& Ft, double wE, double wF) {
    at::AutoDispatchBelowADInplaceOrView guard;
    static auto op = op_handle<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, double, double)>("spk_hip::fm_loss_forward");
    auto r = op.call(E, Et, F, Ft, wE, wF);
    ctx->save_for_backward({std::get<1>(r), std::get<2>(r)});
    return std::get<0>(r);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    TORCH_CHECK(!at::GradMode::is_enabled(), "spk_hip::fm_loss: the backward of the loss node is not recorded (create_graph=True through the loss itself would silently "
                "drop d(dL/dE)/dE and d(dL/dF)/dF); write the loss with torch arithmetic if it has to be differentiated twice");
    auto sv = ctx->get_saved_variables();
    static auto op = op_handle<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&)>("spk_hip::fm_loss_backward");
    at::AutoDispatchBelowADInplaceOrView guard;
    auto r = op.call(g[0], sv[0], sv[1]);
    return {std::get<0>(r), Tensor(), std::get<1>(r), Tensor(), Tensor(), Tensor()};
  }
};
Tensor fm_loss_ad(const Tensor& E, const Tensor& Et, const Tensor& F, const Tensor& Ft, double wE, double wF) { return FmLossFn::apply(E, Et, F, Ft, wE, wF); }
Tensor fm_loss_dev(const Tensor& E, const Tensor& Et, const Tensor& F, const Tensor& Ft, double wE, double wF) { return std::get<0>(fm_loss_raw(E, Et, F, Ft, wE, wF)); }
Tensor rowdot_ad(const Tensor& a, const Tensor& b) { return RowdotFn::apply(a, b); }
Tensor edge_norm_ad(const Tensor& r) { return EdgeNormFn::apply(r); }

// ------------------------------------------------------------------------------------------------ Meta
Tensor act_mul_meta(const OptT&, const Tensor& z, int64_t, int64_t, const OptT&) { return at::empty_like(z); }
Tensor linear_meta(const Tensor& x, const Tensor& w, const OptT&) {
  auto shape = x.sizes().vec();
  shape.back() = w.size(0);
  return at::empty(shape, x.options());
}
Tensor matmul_nn_meta(const Tensor& u, const Tensor& w) {
  auto shape = u.sizes().vec();
  shape.back() = w.size(1);
  return at::empty(shape, u.options());
}
std::tuple<Tensor, Tensor, Tensor> gemm_pair_meta(const Tensor& a, const Tensor& w, bool trans, const Tensor& u, const Tensor& x) {
  auto shape = a.sizes().vec();
  shape.back() = trans ? w.size(1) : w.size(0);
  return {at::empty(shape, a.options()), at::empty({u.size(-1), x.size(-1)}, a.options()), at::empty({u.size(-1)}, a.options())};
}
std::tuple<Tensor, Tensor> matmul_tn_meta(const Tensor& u, const Tensor& x) {
  return {at::empty({u.size(-1), x.size(-1)}, u.options()), at::empty({u.size(-1)}, u.options())};
}
Tensor cfconv_meta(const Tensor& x, const Tensor& W, const OptT&, const OptT&, int64_t n_out) { return at::empty({n_out, W.size(1)}, x.options()); }
Tensor edge_mul_meta(const Tensor& a, const Tensor& b, const OptT& ia, const OptT& ib) {
  const int64_t E = has(ia) ? ia->size(0) : (has(ib) ? ib->size(0) : a.size(0));
  return at::empty({E, a.size(1)}, a.options());
}
Tensor vec3_meta(int64_t op, const Tensor& A, const Tensor& B) {
  auto lead = [](const Tensor& t, int64_t drop) { auto v = t.sizes().vec(); v.resize(v.size() - drop); return v; };
  const int64_t F = A.size(-1);
  if (op == SPK_VEC3_SCALE) return at::empty(A.sizes(), A.options());
  if (op == SPK_VEC3_DOT || op == SPK_VEC3_CONTRACT) { auto v = lead(A, 2); v.push_back(1); v.push_back(F); return at::empty(v, A.options()); }
  if (op == SPK_VEC3_OUTER) { auto v = lead(B, 1); v.push_back(3); v.push_back(F); return at::empty(v, A.options()); }
  return at::empty(lead(A, 1), A.options());
}
Tensor radial_d_meta(const Tensor& d, const OptT&, int64_t kind, const Tensor& p0, const OptT&, double, int64_t) {
  auto shape = d.sizes().vec();
  if (kind != 2) shape.push_back(p0.size(0));
  return at::empty(shape, d.options());
}
Tensor radial_c_meta(const Tensor&, const Tensor& d, const OptT&, int64_t, const Tensor&, const OptT&, double, int64_t) { return at::empty_like(d); }
Tensor rowscale_meta(const Tensor& W, const Tensor&) { return at::empty_like(W); }
Tensor rowdot_meta(const Tensor& a, const Tensor&) {
  auto shape = a.sizes().vec();
  shape.pop_back();
  return at::empty(shape, a.options());
}
Tensor edge_norm_meta(const Tensor& r) { return at::empty({r.size(0)}, r.options()); }
Tensor fm_loss_meta(const Tensor& E, const Tensor&, const Tensor&, const Tensor&, double, double) { return at::empty({}, E.options()); }
std::tuple<Tensor, Tensor, Tensor> fm_loss_forward_meta(const Tensor& E, const Tensor&, const Tensor& F, const Tensor&, double, double) {
  return {at::empty({}, E.options()), at::empty_like(E), at::empty_like(F)};
}
std::tuple<Tensor, Tensor> fm_loss_backward_meta(const Tensor&, const Tensor& gE, const Tensor& gF) { return {at::empty_like(gE), at::empty_like(gF)}; }

// ------------------------------------------------------------------------------------------------ registration
const char* const kTrainOps[] = {"act_mul", "linear", "matmul_nn", "matmul_tn", "cfconv", "edge_mul", "radial_d", "radial_c", "rowscale", "rowdot", "edge_norm", "vec3", "gemm_pair", "fm_loss", "fm_loss_forward", "fm_loss_backward"};

void train_defs(torch::Library& m) {
  m.def("act_mul(Tensor? a, Tensor z, int act, int order, Tensor? c=None) -> Tensor");                   // a . act^(order)(z) + c
  m.def("linear(Tensor x, Tensor weight, Tensor? bias) -> Tensor");                                       // x W^T + b
  m.def("matmul_nn(Tensor u, Tensor weight) -> Tensor");                                                  // u W
  m.def("matmul_tn(Tensor u, Tensor x) -> (Tensor, Tensor)");                                             // (u^T x, column sums of u)
  m.def("gemm_pair(Tensor a, Tensor weight, bool trans, Tensor u, Tensor x) -> (Tensor, Tensor, Tensor)");  // raw: (a w | a w^T, u^T x, column sums of u), one launch
  m.def("cfconv(Tensor x, Tensor W, Tensor? idx_out, Tensor? idx_src, int n_out) -> Tensor");             // schnet.py:64-66 (None: identity)
  m.def("edge_mul(Tensor a, Tensor b, Tensor? idx_a, Tensor? idx_b) -> Tensor");                          // Wij * x[idx_j], painn.py:57
  m.def("vec3(int op, Tensor A, Tensor B) -> Tensor");                                                    // 3-vector products of painn.py:60-63, 104-114
  m.def("radial_d(Tensor d, Tensor? a, int kind, Tensor p0, Tensor? p1, float cutoff, int order) -> Tensor");   // nn/radial.py, nn/cutoff.py, order-th derivative
  m.def("radial_c(Tensor G, Tensor d, Tensor? a, int kind, Tensor p0, Tensor? p1, float cutoff, int order) -> Tensor");
  m.def("rowscale(Tensor W, Tensor s) -> Tensor");                                                        // Wij * rcut_ij[:, None], schnet.py:61
  m.def("rowdot(Tensor a, Tensor b) -> Tensor");
  m.def("edge_norm(Tensor r_ij) -> Tensor");                                                              // torch.norm(r_ij, dim=1), schnet.py:156
  m.def("fm_loss(Tensor E, Tensor E_t, Tensor F, Tensor F_t, float w_e, float w_f) -> Tensor");           // w_e MSE(E) + w_f MSE(F): the loss of a force-matching step (task.py:59-66, 142-146)
  m.def("fm_loss_forward(Tensor E, Tensor E_t, Tensor F, Tensor F_t, float w_e, float w_f) -> (Tensor, Tensor, Tensor)");
  m.def("fm_loss_backward(Tensor g, Tensor gE, Tensor gF) -> (Tensor, Tensor)");
}
void train_impl_device(torch::Library& m) {
  m.impl("act_mul", act_mul_raw);
  m.impl("linear", linear_raw);
  m.impl("matmul_nn", matmul_nn_raw);
  m.impl("matmul_tn", matmul_tn_raw);
  m.impl("gemm_pair", gemm_pair_raw);
  m.impl("cfconv", cfconv_raw);
  m.impl("edge_mul", edge_mul_raw);
  m.impl("vec3", vec3_raw);
  m.impl("radial_d", radial_d_raw);
  m.impl("radial_c", radial_c_raw);
  m.impl("rowscale", rowscale_raw);
  m.impl("rowdot", rowdot_raw);
  m.impl("edge_norm", edge_norm_raw);
  m.impl("fm_loss", fm_loss_dev);
  m.impl("fm_loss_forward", fm_loss_raw);
  m.impl("fm_loss_backward", fm_loss_backward_raw);
}
void train_impl_autograd(torch::Library& m) {
  m.impl("act_mul", act_mul_ad);
  m.impl("linear", linear_ad);
  m.impl("matmul_nn", matmul_nn_ad);
  m.impl("matmul_tn", matmul_tn_ad);
  m.impl("cfconv", cfconv_ad);
  m.impl("edge_mul", edge_mul_ad);
  m.impl("vec3", vec3_ad);
  m.impl("radial_d", radial_d_ad);
  m.impl("radial_c", radial_c_ad);
  m.impl("rowscale", rowscale_ad);
  m.impl("rowdot", rowdot_ad);
  m.impl("edge_norm", edge_norm_ad);
  m.impl("fm_loss", fm_loss_ad);
}
void train_impl_meta(torch::Library& m) {
  m.impl("act_mul", act_mul_meta);
  m.impl("linear", linear_meta);
  m.impl("matmul_nn", matmul_nn_meta);
  m.impl("matmul_tn", matmul_tn_meta);
  m.impl("gemm_pair", gemm_pair_meta);
  m.impl("cfconv", cfconv_meta);
  m.impl("edge_mul", edge_mul_meta);
  m.impl("vec3", vec3_meta);
  m.impl("radial_d", radial_d_meta);
  m.impl("radial_c", radial_c_meta);
  m.impl("rowscale", rowscale_meta);
  m.impl("rowdot", rowdot_meta);
  m.impl("edge_norm", edge_norm_meta);
  m.impl("fm_loss", fm_loss_meta);
  m.impl("fm_loss_forward", fm_loss_forward_meta);
  m.impl("fm_loss_backward", fm_loss_backward_meta);
}
