// torch.ops.spk_hip.schnet_fm / painn_fm: the standard potential in TRAINING mode -- PairwiseDistances -> SchNet / PaiNN -> Atomwise ->
// Forces (model/base.py:174-190 with atomistic/response.py:59-68, create_graph = self.training) -- as ONE autograd node whose backward is
// the forward-over-reverse engine of spk_fm.hip: given dL/dE and dL/dF it returns the gradient of the loss w.r.t. every weight
// (include/spk_hip.h, "force-matching gradients").  The node is differentiable ONCE w.r.t. the weights, which is what a training step asks
// for (task.py:166-185); gradients w.r.t. the positions or a recorded (create_graph) backward are refused, not silently dropped.
// Included by spk_torch.cpp (inside its anonymous namespace, after spk_torch_train.h).

struct FmCall {
  std::vector<Tensor> keep;
  std::vector<spk_schnet_layer_t> sl;
  std::vector<spk_painn_layer_t> pl;
  spk_schnet_t sm;
  spk_painn_t pm;
  spk_head_t head;
  spk_radial_t rb;
  spk_fm_batch_t b;
  Tensor p0, p1, emb, Z, R, off, ii, jj, im;
  bool painn;
};

// kind 0: SchNet (ws = 9 tensors per interaction), 1: PaiNN (9 per interaction + filter_net.{weight, bias})
void fm_setup(FmCall& c, bool painn, const Tensor& emb_in, const Tensor& Z_in, const Tensor& R_in, const OptT& off_in, const Tensor& ii_in, const Tensor& jj_in,
              const Tensor& im_in, int64_t n_mol, at::TensorList ws, at::TensorList head, int64_t n_filters, bool shared_filters, double eps, int64_t rbf_kind,
              const Tensor& p0_in, const OptT& p1_in, double cutoff, int64_t head_act, const char* who) {
  TORCH_CHECK(head.size() == 4, who, ": head = [outnet.0.weight, outnet.0.bias, outnet.1.weight, outnet.1.bias]");
  c.painn = painn;
  c.emb = f32(emb_in.detach(), who);
  TORCH_CHECK(c.emb.dim() == 2, who, ": embedding table must be [n_types, F]");
  const int64_t F = c.emb.size(1);
  c.Z = i64(Z_in, who); c.R = f32(R_in.detach(), who); c.off = opt_f32(off_in, who);
  c.ii = i64(ii_in, who); c.jj = i64(jj_in, who); c.im = i64(im_in, who);
  c.p0 = f32(p0_in, who); c.p1 = opt_f32(p1_in, who);
  const int64_t N = c.Z.size(0), E = c.ii.size(0);
  TORCH_CHECK(c.R.dim() == 2 && c.R.size(0) == N && c.R.size(1) == 3 && c.jj.size(0) == E && c.im.size(0) == N && (!c.off.defined() || c.off.size(0) == E),
              who, ": inconsistent batch shapes");
  auto w = [&](const Tensor& t) { Tensor r = f32(t.detach(), who); c.keep.push_back(r); return fp(r); };
  if (!painn) {
    TORCH_CHECK(ws.size() % kSchnetPerLayer == 0, who, ": expected 9 weight tensors per interaction, got ", ws.size());
    const int64_t L = ws.size() / kSchnetPerLayer;
    c.sl.resize(std::max<int64_t>(L, 1));
    for (int64_t l = 0; l < L; ++l) {
      spk_schnet_layer_t& P = c.sl[l];
      std::memset(&P, 0, sizeof(P));
      const int64_t o = l * kSchnetPerLayer;
      TORCH_CHECK(ws[o].size(0) == n_filters && ws[o].size(1) == F && ws[o + 7].size(0) == F, who, ": weight shapes do not match n_filters / n_atom_basis");
      P.in2f_w = w(ws[o]); P.fn_w1 = w(ws[o + 1]); P.fn_b1 = w(ws[o + 2]); P.fn_w2 = w(ws[o + 3]); P.fn_b2 = w(ws[o + 4]);
      P.f2out_w1 = w(ws[o + 5]); P.f2out_b1 = w(ws[o + 6]); P.f2out_w2 = w(ws[o + 7]); P.f2out_b2 = w(ws[o + 8]);
    }
    std::memset(&c.sm, 0, sizeof(c.sm));
    c.sm.n_atom_basis = (int32_t)F; c.sm.n_filters = (int32_t)n_filters; c.sm.n_interactions = (int32_t)L; c.sm.layers = c.sl.data();
  } else {
    TORCH_CHECK(ws.size() >= 2 + kPainnPerLayer && (ws.size() - 2) % kPainnPerLayer == 0, who, ": expected 9 weight tensors per interaction + filter_net.{weight,bias}, got ", ws.size());
    const int64_t L = (ws.size() - 2) / kPainnPerLayer;
    const float* fw = w(ws[ws.size() - 2]);
    const float* fb = w(ws[ws.size() - 1]);
    const int64_t K = ws[ws.size() - 2].size(1);
    TORCH_CHECK(ws[ws.size() - 2].size(0) == 3 * F * (shared_filters ? 1 : L) && K == c.p0.size(0), who, ": filter_net.weight shape does not match the model");
    c.pl.resize(L);
    for (int64_t l = 0; l < L; ++l) {
      spk_painn_layer_t& P = c.pl[l];
      std::memset(&P, 0, sizeof(P));
      const int64_t o = l * kPainnPerLayer;
      P.ctx_w1 = w(ws[o]); P.ctx_b1 = w(ws[o + 1]); P.ctx_w2 = w(ws[o + 2]); P.ctx_b2 = w(ws[o + 3]); P.mix_w = w(ws[o + 4]);
      P.ictx_w1 = w(ws[o + 5]); P.ictx_b1 = w(ws[o + 6]); P.ictx_w2 = w(ws[o + 7]); P.ictx_b2 = w(ws[o + 8]);
      const int64_t row0 = shared_filters ? 0 : 3 * F * l;
      P.filt_w = fw + row0 * K;
      P.filt_b = fb + row0;
    }
    std::memset(&c.pm, 0, sizeof(c.pm));
    c.pm.n_atom_basis = (int32_t)F; c.pm.n_interactions = (int32_t)L; c.pm.epsilon = (float)eps; c.pm.layers = c.pl.data();
    TORCH_CHECK(L == 1 || shared_filters == (c.pl[1].filt_w == c.pl[0].filt_w), who, ": shared_filters flag does not match the filter rows");
  }
  std::memset(&c.head, 0, sizeof(c.head));
  c.head.w1 = w(head[0]); c.head.b1 = w(head[1]); c.head.w2 = w(head[2]); c.head.b2 = w(head[3]);
  c.head.n_hidden = (int32_t)head[0].size(0); c.head.act = (int32_t)head_act;
  TORCH_CHECK(head[0].dim() == 2 && head[0].size(1) == F && head[2].numel() == head[0].size(0) && head[3].numel() == 1, who, ": head shapes (default build_mlp(F, 1, n_layers=2) expected)");
  c.rb = radial_of(rbf_kind, c.p0, c.p1, cutoff);
  std::memset(&c.b, 0, sizeof(c.b));
  c.b.n_atoms = N; c.b.n_edges = E; c.b.n_mol = n_mol;
  c.b.Z = c.Z.data_ptr<int64_t>(); c.b.idx_i = c.ii.data_ptr<int64_t>(); c.b.idx_j = c.jj.data_ptr<int64_t>(); c.b.idx_m = c.im.data_ptr<int64_t>();
  c.b.R = fp(c.R); c.b.offsets = fp(c.off); c.b.embedding = fp(c.emb); c.b.n_types = (int32_t)c.emb.size(0);
}

// raw forward: (E [n_mol], F [N,3], workspace, err [1] int32).  err stays on the device (bit 0 = idx_i / idx_m not ascending: the energies
// of such a batch are NaN -- loud without a host round trip)
std::tuple<Tensor, Tensor, Tensor, Tensor> fm_forward_raw(bool painn, const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off, const Tensor& ii, const Tensor& jj,
                                                         const Tensor& im, int64_t n_mol, at::TensorList ws, at::TensorList head, int64_t n_filters,
                                                         bool shared_filters, double eps, int64_t rbf_kind, const Tensor& p0, const OptT& p1, double cutoff,
                                                         int64_t head_act) {
  const char* who = painn ? "painn_fm" : "schnet_fm";
  FmCall c;
  fm_setup(c, painn, emb, Z, R, off, ii, jj, im, n_mol, ws, head, n_filters, shared_filters, eps, rbf_kind, p0, p1, cutoff, head_act, who);
  c10::DeviceGuard guard(c.R.device());
  const int64_t bytes = painn ? spk_painn_fm_workspace_bytes(&c.pm, &c.head, &c.rb, c.b.n_atoms, c.b.n_edges, n_mol, c.b.n_types)
                              : spk_schnet_fm_workspace_bytes(&c.sm, &c.head, &c.rb, c.b.n_atoms, c.b.n_edges, n_mol, c.b.n_types);
  TORCH_CHECK(bytes > 0, who, ": bad sizes");
  Tensor wsb = at::empty({bytes}, c.R.options().dtype(at::kByte));
  Tensor Eo = at::empty({n_mol}, c.R.options()), Fo = at::empty({c.b.n_atoms, 3}, c.R.options());
  // static-shape mode (graph-replayed steps): the declared index buffers are validated by StaticLists.refresh() inside the same graph
  Tensor err = g_static_on ? at::empty({0}, c.R.options().dtype(at::kInt)) : at::zeros({1}, c.R.options().dtype(at::kInt));
  int32_t* errp = err.numel() ? err.data_ptr<int32_t>() : nullptr;
  void* st = stream_of(c.R);
  if (painn) check(spk_painn_fm_forward_f32(&c.pm, &c.head, &c.rb, &c.b, wsb.data_ptr(), fpm(Eo), fpm(Fo), errp, st));
  else check(spk_schnet_fm_forward_f32(&c.sm, &c.head, &c.rb, &c.b, wsb.data_ptr(), fpm(Eo), fpm(Fo), errp, st));
  return {Eo, Fo, wsb, err};
}
// raw backward: the flat gradient buffer of include/spk_hip.h
Tensor fm_backward_raw(bool painn, const Tensor& wsb, const Tensor& gE_in, const Tensor& gF_in, const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off,
                       const Tensor& ii, const Tensor& jj, const Tensor& im, int64_t n_mol, at::TensorList ws, at::TensorList head, int64_t n_filters, bool shared_filters,
                       double eps, int64_t rbf_kind, const Tensor& p0, const OptT& p1, double cutoff, int64_t head_act) {
  const char* who = painn ? "painn_fm" : "schnet_fm";
  FmCall c;
  fm_setup(c, painn, emb, Z, R, off, ii, jj, im, n_mol, ws, head, n_filters, shared_filters, eps, rbf_kind, p0, p1, cutoff, head_act, who);
  c10::DeviceGuard guard(c.R.device());
  Tensor gE = f32(gE_in, who), gF = f32(gF_in, who);
  TORCH_CHECK(gE.numel() == n_mol && gF.numel() == 3 * c.b.n_atoms, who, ": gradient shapes");
  const int64_t n = painn ? spk_painn_fm_grad_floats(&c.pm, &c.head, &c.rb, c.b.n_types) : spk_schnet_fm_grad_floats(&c.sm, &c.head, &c.rb, c.b.n_types);
  Tensor flat = at::empty({n}, c.R.options());
  Tensor wsm = wsb;
  void* st = stream_of(c.R);
  if (painn) check(spk_painn_fm_backward_f32(&c.pm, &c.head, &c.rb, &c.b, wsm.data_ptr(), fp(gE), fp(gF), fpm(flat), st));
  else check(spk_schnet_fm_backward_f32(&c.sm, &c.head, &c.rb, &c.b, wsm.data_ptr(), fp(gE), fp(gF), fpm(flat), st));
  return flat;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> schnet_fm_forward_op(const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off, const Tensor& ii, const Tensor& jj,
                                                               const Tensor& im, int64_t n_mol, at::TensorList ws, at::TensorList head, int64_t n_filters, int64_t rbf_kind,
                                                               const Tensor& p0, const OptT& p1, double cutoff, int64_t head_act) {
  return fm_forward_raw(false, emb, Z, R, off, ii, jj, im, n_mol, ws, head, n_filters, false, 0.0, rbf_kind, p0, p1, cutoff, head_act);
}
Tensor schnet_fm_backward_op(const Tensor& wsb, const Tensor& gE, const Tensor& gF, const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off, const Tensor& ii,
                             const Tensor& jj, const Tensor& im, int64_t n_mol, at::TensorList ws, at::TensorList head, int64_t n_filters, int64_t rbf_kind, const Tensor& p0,
                             const OptT& p1, double cutoff, int64_t head_act) {
  return fm_backward_raw(false, wsb, gE, gF, emb, Z, R, off, ii, jj, im, n_mol, ws, head, n_filters, false, 0.0, rbf_kind, p0, p1, cutoff, head_act);
}
std::tuple<Tensor, Tensor, Tensor, Tensor> painn_fm_forward_op(const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off, const Tensor& ii, const Tensor& jj,
                                                              const Tensor& im, int64_t n_mol, at::TensorList ws, at::TensorList head, bool shared_filters, double eps,
                                                              int64_t rbf_kind, const Tensor& p0, const OptT& p1, double cutoff, int64_t head_act) {
  return fm_forward_raw(true, emb, Z, R, off, ii, jj, im, n_mol, ws, head, emb.size(1), shared_filters, eps, rbf_kind, p0, p1, cutoff, head_act);
}
Tensor painn_fm_backward_op(const Tensor& wsb, const Tensor& gE, const Tensor& gF, const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off, const Tensor& ii,
                            const Tensor& jj, const Tensor& im, int64_t n_mol, at::TensorList ws, at::TensorList head, bool shared_filters, double eps, int64_t rbf_kind,
                            const Tensor& p0, const OptT& p1, double cutoff, int64_t head_act) {
  return fm_backward_raw(true, wsb, gE, gF, emb, Z, R, off, ii, jj, im, n_mol, ws, head, emb.size(1), shared_filters, eps, rbf_kind, p0, p1, cutoff, head_act);
}

// ------------------------------------------------------------------------------------------------ autograd node
constexpr const char* kFmFirstOrder =
    ": the force-matching operator provides first-order gradients of a loss(E, F) w.r.t. the weights (what a training step asks for, task.py:166-185); "
    "a recorded backward (create_graph=True) or a gradient w.r.t. the positions is not provided -- set `model.fm_engine = False` to run the operator-by-operator "
    "training path, which is differentiable to any order";

// (capture status of the current stream)
struct PotentialFmFn : public torch::autograd::Function<PotentialFmFn> {
  // argument slots: 0 emb | 1 Z | 2 R | 3 idx_i | 4 idx_j | 5 idx_m | 6 p0 | 7 offsets | 8 p1 | ws... | head... | 8 scalars
  static variable_list forward(AutogradContext* ctx, const Tensor& emb, const Tensor& Z, const Tensor& R, const Tensor& idx_i, const Tensor& idx_j, const Tensor& idx_m,
                               const Tensor& p0, const OptT& offsets, const OptT& p1, at::TensorList ws, at::TensorList head, bool painn, int64_t n_mol, int64_t n_filters,
                               bool shared_filters, double eps, int64_t rbf_kind, double cutoff, int64_t head_act) {
    at::AutoDispatchBelowADInplaceOrView guard;
    std::tuple<Tensor, Tensor, Tensor, Tensor> out;
    if (painn) {
      static auto op = op_handle<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const OptT&, const Tensor&, const Tensor&, const Tensor&,
                                                                            int64_t, at::TensorList, at::TensorList, bool, double, int64_t, const Tensor&, const OptT&, double,
                                                                            int64_t)>("spk_hip::painn_fm_forward");
      out = op.call(emb, Z, R, offsets, idx_i, idx_j, idx_m, n_mol, ws, head, shared_filters, eps, rbf_kind, p0, p1, cutoff, head_act);
    } else {
      static auto op = op_handle<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const OptT&, const Tensor&, const Tensor&, const Tensor&,
                                                                            int64_t, at::TensorList, at::TensorList, int64_t, int64_t, const Tensor&, const OptT&, double,
                                                                            int64_t)>("spk_hip::schnet_fm_forward");
      out = op.call(emb, Z, R, offsets, idx_i, idx_j, idx_m, n_mol, ws, head, n_filters, rbf_kind, p0, p1, cutoff, head_act);
    }
    const bool has_off = offsets.has_value() && offsets->defined(), has_p1 = p1.has_value() && p1->defined();
    std::vector<Tensor> sv{emb, Z, R, idx_i, idx_j, idx_m, p0, has_off ? *offsets : Tensor(), has_p1 ? *p1 : Tensor(), std::get<2>(out)};
    for (const auto& w : ws) sv.push_back(w);
    for (const auto& w : head) sv.push_back(w);
    ctx->save_for_backward(sv);
    ctx->saved_data["cfg"] = std::vector<int64_t>{painn, n_mol, n_filters, shared_filters, rbf_kind, head_act, (int64_t)ws.size(), (int64_t)head.size(), has_off, has_p1};
    ctx->saved_data["eps"] = eps;
    ctx->saved_data["cutoff"] = cutoff;
    ctx->saved_data["err"] = std::get<3>(out);
    return {std::get<0>(out), std::get<1>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto cfg = ctx->saved_data["cfg"].toIntVector();
    const bool painn = cfg[0] != 0;
    const char* who = painn ? "spk_hip::painn_fm" : "spk_hip::schnet_fm";
    TORCH_CHECK(!at::GradMode::is_enabled(), who, kFmFirstOrder);
    TORCH_CHECK(!ctx->needs_input_grad(2), who, kFmFirstOrder);
    {
      // the forward's validity flag (bit 0: idx_i / idx_m not ascending -- the engine's row kernels need the list sorted by centre atom, the
      // energies are NaN then; bit 1: an atomic number outside the embedding table): surfaced here, at the first point the caller waits for
      // the device anyway.  Not inside a stream capture (no host read there: static-shape steps poll StaticLists.check() instead).
      const Tensor err = ctx->saved_data["err"].toTensor();
      if (err.defined() && err.numel() == 1 && c10::hip::currentStreamCaptureStatusMayInitCtx() == c10::hip::CaptureStatus::None) {
        const int flag = err.item<int>();
        TORCH_CHECK((flag & 1) == 0, who, ": idx_i / idx_m are not sorted ascending. The force-matching engine walks the pair list by centre atom "
                    "(every neighbour list of the reference is sorted this way; CountNeighbors(sorted=False)-style lists are not) -- sort the list, or set "
                    "`model.fm_engine = False` to train through the operator-by-operator path, which takes any pair order");
        // (bit 1 is shared by the range checks of the step: spk_index_jobs sets it for a pair / molecule index out of range, k_fm_embed for an
        //  atomic number outside the embedding table)
        TORCH_CHECK((flag & 2) == 0, who, ": an index is out of range -- a pair index (idx_i / idx_j >= n_atoms), a molecule index (idx_m >= n_molecules) "
                    "or an atomic number outside the embedding table (>= max_z)");
      }
    }
    auto sv = ctx->get_saved_variables();
    const size_t n_ws = (size_t)cfg[6], n_head = (size_t)cfg[7];
    std::vector<Tensor> ws(sv.begin() + 10, sv.begin() + 10 + n_ws), head(sv.begin() + 10 + n_ws, sv.begin() + 10 + n_ws + n_head);
    const Tensor &emb = sv[0], &R = sv[2];
    at::AutoDispatchBelowADInplaceOrView guard;
    Tensor gE = grads[0].defined() ? grads[0] : at::zeros({cfg[1]}, R.options());
    Tensor gF = grads[1].defined() ? grads[1] : at::zeros_like(R);
    Tensor flat;
    if (painn) {
      static auto op = op_handle<Tensor(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const OptT&, const Tensor&, const Tensor&,
                                        const Tensor&, int64_t, at::TensorList, at::TensorList, bool, double, int64_t, const Tensor&, const OptT&, double, int64_t)>(
          "spk_hip::painn_fm_backward");
      flat = op.call(sv[9], gE, gF, emb, sv[1], R, opt_of(sv[7]), sv[3], sv[4], sv[5], cfg[1], ws, head, cfg[3] != 0, ctx->saved_data["eps"].toDouble(), cfg[4], sv[6],
                     opt_of(sv[8]), ctx->saved_data["cutoff"].toDouble(), cfg[5]);
    } else {
      static auto op = op_handle<Tensor(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const OptT&, const Tensor&, const Tensor&,
                                        const Tensor&, int64_t, at::TensorList, at::TensorList, int64_t, int64_t, const Tensor&, const OptT&, double, int64_t)>(
          "spk_hip::schnet_fm_backward");
      flat = op.call(sv[9], gE, gF, emb, sv[1], R, opt_of(sv[7]), sv[3], sv[4], sv[5], cfg[1], ws, head, cfg[2], cfg[4], sv[6], opt_of(sv[8]),
                     ctx->saved_data["cutoff"].toDouble(), cfg[5]);
    }
    // views of the flat buffer in the layout of include/spk_hip.h: interaction weights (list order) | head | embedding
    // `out` has one slot per forward ARGUMENT; needs_input_grad() is indexed by the defined tensor inputs only (absent optionals have no edge)
    variable_list out(9 + n_ws + n_head + 8);
    const size_t edge0 = 7 + (size_t)cfg[8] + (size_t)cfg[9];
    int64_t o = 0;
    for (size_t k = 0; k < n_ws + n_head; ++k) {
      const Tensor& w = k < n_ws ? ws[k] : head[k - n_ws];
      const int64_t n = w.numel();
      if (ctx->needs_input_grad(edge0 + k)) out[9 + k] = flat.narrow(0, o, n).view(w.sizes());
      o += n;
    }
    if (ctx->needs_input_grad(0)) out[0] = flat.narrow(0, o, emb.numel()).view(emb.sizes());
    TORCH_CHECK(o + emb.numel() == flat.numel(), who, ": gradient layout mismatch");
    return out;
  }
};

std::tuple<Tensor, Tensor> schnet_fm_ad(const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off, const Tensor& ii, const Tensor& jj, const Tensor& im, int64_t n_mol,
                                        at::TensorList ws, at::TensorList head, int64_t n_filters, int64_t rbf_kind, const Tensor& p0, const OptT& p1, double cutoff,
                                        int64_t head_act) {
  auto r = PotentialFmFn::apply(emb, Z, R, ii, jj, im, p0, off, p1, ws, head, false, n_mol, n_filters, false, 0.0, rbf_kind, cutoff, head_act);
  return {r[0], r[1]};
}
std::tuple<Tensor, Tensor> painn_fm_ad(const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off, const Tensor& ii, const Tensor& jj, const Tensor& im, int64_t n_mol,
                                       at::TensorList ws, at::TensorList head, bool shared_filters, double eps, int64_t rbf_kind, const Tensor& p0, const OptT& p1, double cutoff,
                                       int64_t head_act) {
  auto r = PotentialFmFn::apply(emb, Z, R, ii, jj, im, p0, off, p1, ws, head, true, n_mol, emb.size(1), shared_filters, eps, rbf_kind, cutoff, head_act);
  return {r[0], r[1]};
}
std::tuple<Tensor, Tensor> schnet_fm_dev(const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off, const Tensor& ii, const Tensor& jj, const Tensor& im, int64_t n_mol,
                                         at::TensorList ws, at::TensorList head, int64_t n_filters, int64_t rbf_kind, const Tensor& p0, const OptT& p1, double cutoff,
                                         int64_t head_act) {
  auto r = fm_forward_raw(false, emb, Z, R, off, ii, jj, im, n_mol, ws, head, n_filters, false, 0.0, rbf_kind, p0, p1, cutoff, head_act);
  return {std::get<0>(r), std::get<1>(r)};
}
std::tuple<Tensor, Tensor> painn_fm_dev(const Tensor& emb, const Tensor& Z, const Tensor& R, const OptT& off, const Tensor& ii, const Tensor& jj, const Tensor& im, int64_t n_mol,
                                        at::TensorList ws, at::TensorList head, bool shared_filters, double eps, int64_t rbf_kind, const Tensor& p0, const OptT& p1, double cutoff,
                                        int64_t head_act) {
  auto r = fm_forward_raw(true, emb, Z, R, off, ii, jj, im, n_mol, ws, head, emb.size(1), shared_filters, eps, rbf_kind, p0, p1, cutoff, head_act);
  return {std::get<0>(r), std::get<1>(r)};
}

// ------------------------------------------------------------------------------------------------ Meta
std::tuple<Tensor, Tensor> schnet_fm_meta(const Tensor&, const Tensor&, const Tensor& R, const OptT&, const Tensor&, const Tensor&, const Tensor&, int64_t n_mol, at::TensorList,
                                          at::TensorList, int64_t, int64_t, const Tensor&, const OptT&, double, int64_t) {
  return {at::empty({n_mol}, R.options()), at::empty_like(R)};
}
std::tuple<Tensor, Tensor> painn_fm_meta(const Tensor&, const Tensor&, const Tensor& R, const OptT&, const Tensor&, const Tensor&, const Tensor&, int64_t n_mol, at::TensorList,
                                         at::TensorList, bool, double, int64_t, const Tensor&, const OptT&, double, int64_t) {
  return {at::empty({n_mol}, R.options()), at::empty_like(R)};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> schnet_fm_forward_meta(const Tensor&, const Tensor&, const Tensor& R, const OptT&, const Tensor&, const Tensor&, const Tensor&,
                                                                 int64_t n_mol, at::TensorList, at::TensorList, int64_t, int64_t, const Tensor&, const OptT&, double, int64_t) {
  return {at::empty({n_mol}, R.options()), at::empty_like(R), at::empty({0}, R.options().dtype(at::kByte)), at::empty({0}, R.options().dtype(at::kInt))};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> painn_fm_forward_meta(const Tensor&, const Tensor&, const Tensor& R, const OptT&, const Tensor&, const Tensor&, const Tensor&,
                                                                int64_t n_mol, at::TensorList, at::TensorList, bool, double, int64_t, const Tensor&, const OptT&, double, int64_t) {
  return {at::empty({n_mol}, R.options()), at::empty_like(R), at::empty({0}, R.options().dtype(at::kByte)), at::empty({0}, R.options().dtype(at::kInt))};
}
int64_t fm_flat_numel(const Tensor& emb, at::TensorList ws, at::TensorList head) {
  int64_t n = emb.numel();
  for (const auto& w : ws) n += w.numel();
  for (const auto& w : head) n += w.numel();
  return n;
}
Tensor schnet_fm_backward_meta(const Tensor&, const Tensor&, const Tensor&, const Tensor& emb, const Tensor&, const Tensor& R, const OptT&, const Tensor&, const Tensor&,
                               const Tensor&, int64_t, at::TensorList ws, at::TensorList head, int64_t, int64_t, const Tensor&, const OptT&, double, int64_t) {
  return at::empty({fm_flat_numel(emb, ws, head)}, R.options());
}
Tensor painn_fm_backward_meta(const Tensor&, const Tensor&, const Tensor&, const Tensor& emb, const Tensor&, const Tensor& R, const OptT&, const Tensor&, const Tensor&,
                              const Tensor&, int64_t, at::TensorList ws, at::TensorList head, bool, double, int64_t, const Tensor&, const OptT&, double, int64_t) {
  return at::empty({fm_flat_numel(emb, ws, head)}, R.options());
}

// ------------------------------------------------------------------------------------------------ registration
const char* const kFmOps[] = {"schnet_fm", "painn_fm", "schnet_fm_forward", "schnet_fm_backward", "painn_fm_forward", "painn_fm_backward"};
void fm_defs(torch::Library& m) {
  // (E [n_mol], forces [N,3]) of the standard potential with weight gradients of any loss(E, F) in its backward (training mode)
  m.def("schnet_fm(Tensor embedding, Tensor Z, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, Tensor[] weights, Tensor[] head, "
        "int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> (Tensor, Tensor)");
  m.def("painn_fm(Tensor embedding, Tensor Z, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, Tensor[] weights, Tensor[] head, "
        "bool shared_filters, float epsilon, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> (Tensor, Tensor)");
  m.def("schnet_fm_forward(Tensor embedding, Tensor Z, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, Tensor[] weights, Tensor[] head, "
        "int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("schnet_fm_backward(Tensor workspace, Tensor gE, Tensor gF, Tensor embedding, Tensor Z, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, "
        "Tensor[] weights, Tensor[] head, int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> Tensor");
  m.def("painn_fm_forward(Tensor embedding, Tensor Z, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, Tensor[] weights, Tensor[] head, "
        "bool shared_filters, float epsilon, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("painn_fm_backward(Tensor workspace, Tensor gE, Tensor gF, Tensor embedding, Tensor Z, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, "
        "Tensor[] weights, Tensor[] head, bool shared_filters, float epsilon, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> Tensor");
}
void fm_impl_device(torch::Library& m) {
  m.impl("schnet_fm", schnet_fm_dev);
  m.impl("painn_fm", painn_fm_dev);
  m.impl("schnet_fm_forward", schnet_fm_forward_op);
  m.impl("schnet_fm_backward", schnet_fm_backward_op);
  m.impl("painn_fm_forward", painn_fm_forward_op);
  m.impl("painn_fm_backward", painn_fm_backward_op);
}
void fm_impl_autograd(torch::Library& m) {
  m.impl("schnet_fm", schnet_fm_ad);
  m.impl("painn_fm", painn_fm_ad);
}
void fm_impl_meta(torch::Library& m) {
  m.impl("schnet_fm", schnet_fm_meta);
  m.impl("painn_fm", painn_fm_meta);
  m.impl("schnet_fm_forward", schnet_fm_forward_meta);
  m.impl("painn_fm_forward", painn_fm_forward_meta);
  m.impl("schnet_fm_backward", schnet_fm_backward_meta);
  m.impl("painn_fm_backward", painn_fm_backward_meta);
}
