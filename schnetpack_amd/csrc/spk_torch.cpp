// libspk_torch.so -- the PyTorch-ROCm face of libspk_hip.so:
// `TORCH_LIBRARY(spk_hip, ...)` custom operators with C++ `torch::autograd::Function`s and Meta kernels over the
// raw C-ABI launchers of include/spk_hip.h.  The Python mirrors of schnetpack.representation.{SchNet, PaiNN},
// schnetpack.nn.{Dense, scatter_add, GaussianRBF, BesselRBF, CosineCutoff} and
// schnetpack.atomistic.{PairwiseDistances, Atomwise} call `torch.ops.spk_hip.*`, so they are TorchScript-able
// (tests/nn/test_schnet.py:83-96 of the reference, src/scripts/spkdeploy:16-40,
// md/calculators/schnetpack_calculator.py:105-107) and a scripted model loads from C++
// (interfaces/lammps/pair_schnetpack.cpp:128) once this library has been dlopen'ed.
//
// No compute lives here: every operator forwards device pointers to libspk_hip.so on torch's current HIP stream.
// There is no CPU kernel: CPU tensors raise.  Derived per-list data (CSR row pointers, reverse-edge map, canonical
// pairs: `spk_edge_plan`) and per-model data (transposed / packed weight images) are cached here, keyed by the
// identity and version of the tensors they were derived from.
//
// Autograd contract of the fused eval-path operators (schnet, painn, atomwise): first-order gradients w.r.t. the
// geometry (r_ij) and the input features (embedding rows) -- what `Forces` (atomistic/response.py:59-76) asks for in
// eval mode.  The weights are real inputs of the autograd node; if a backward pass asks for THEIR gradient, or asks to
// record the backward itself (create_graph=True), the operator raises instead of returning zeros: use the training
// path (`module.train()`), which is built from the primitives below that are differentiable to any order.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGraphsC10Utils.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <cstring>
#include <algorithm>
#include <atomic>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/spk_hip.h"

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace {

// ------------------------------------------------------------------------------------------------ helpers
void check(int rc) { TORCH_CHECK(rc == 0, "spk_hip error ", rc, ": ", spk_last_error()); }

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void require_device(const Tensor& t, const char* who) {
  TORCH_CHECK(t.is_cuda(), who, ": tensor on ", t.device(),
              " -- schnetpack_amd runs on ROCm devices only (there is no CPU fallback)");
}

// contiguous fp32 device tensor (the C ABI's only floating-point layout)
Tensor f32(const Tensor& t, const char* who) {
  require_device(t, who);
  TORCH_CHECK(t.scalar_type() == at::kFloat, who, ": dtype ", t.scalar_type(), " unsupported; the HIP path computes in float32");
  return t.contiguous();
}
Tensor i64(const Tensor& t, const char* who) {
  require_device(t, who);
  return t.scalar_type() == at::kLong ? t.contiguous() : t.to(at::kLong).contiguous();
}
const float* fp(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* fpm(Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
const float* fpo(const c10::optional<Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }
Tensor opt_f32(const c10::optional<Tensor>& t, const char* who) { return (t.has_value() && t->defined()) ? f32(*t, who) : Tensor(); }

uint64_t version_of(const Tensor& t) { return t.defined() ? (uint64_t)t._version() : 0; }

template <class T>
struct Lru {
  std::list<std::pair<std::vector<uint64_t>, std::shared_ptr<T>>> items;
  size_t cap;
  explicit Lru(size_t c) : cap(c) {}
  std::shared_ptr<T> get(const std::vector<uint64_t>& key) {
    for (auto it = items.begin(); it != items.end(); ++it)
      if (it->first == key) {
        items.splice(items.begin(), items, it);
        return items.front().second;
      }
    return nullptr;
  }
  void put(const std::vector<uint64_t>& key, std::shared_ptr<T> v) {
    items.emplace_front(key, std::move(v));
    while (items.size() > cap) items.pop_back();
  }
  void clear() { items.clear(); }
};

std::mutex g_mutex;

// ------------------------------------------------------------------------------------------------ neighbour-list plan
struct Plan {
  Tensor idx_i, idx_j, rowptr, rev, half, edge_pair, grp_atom0, grp_pair0, grp_tile0;   // keep-alive + device data
  Tensor src_i, src_j;       // the CALLER's index tensors: the cache key is their data_ptr, so they must stay allocated while the entry lives
                             // (an int32 list converted to int64 above would otherwise be freed and its address re-used by another list)
  bool sorted = false, symmetric = false;
  int64_t n_atoms = 0, n_edges = 0, n_half = 0;
  int32_t n_groups = 0, max_group_atoms = 0, max_group_pairs = 0;
  int64_t n_tiles_grouped = 0;
  int filter_pairs = -1;     // -1: undecided
  bool has_r = false;
  std::vector<std::pair<std::vector<uint64_t>, bool>> inside;   // per idx_m tensor: every molecule inside one group, none empty
  // block plan of a large sorted list (PaiNN message kernels of the box regime, spk_painn_blk.hip): index arrays + per-call workspace
  Tensor b_sub_n, b_sub_u, b_uniq, b_jl, b_tile0, b_tinfo, b_desc, b_apack, b_adpack, b_rec, b_part;
  spk_blocks_t blocks;
  int blocks_state = 0;      // 0: not built, 1: usable, -1: the list does not fit / too small
  // the list sorted by neighbour (sorted, asymmetric lists: transposed sums of the backward as row passes, spk_transposed_t)
  Tensor t_idx_i, t_idx_j, t_rowptr, t_perm, t_rperm;
  spk_transposed_t transposed;
  bool has_transposed = false;
  int blocks_rbf = 0, blocks_F = 0;

  spk_graph_t graph() const {
    spk_graph_t g;
    std::memset(&g, 0, sizeof(g));
    g.n_atoms = n_atoms;
    g.n_edges = n_edges;
    g.idx_i = n_edges ? idx_i.data_ptr<int64_t>() : nullptr;
    g.idx_j = n_edges ? idx_j.data_ptr<int64_t>() : nullptr;
    g.rowptr = sorted ? rowptr.data_ptr<int32_t>() : nullptr;
    g.sorted = sorted;
    g.symmetric = symmetric;
    g.rev = symmetric && rev.defined() ? rev.data_ptr<int32_t>() : nullptr;
    g.half = half.defined() ? half.data_ptr<int32_t>() : nullptr;
    g.n_half = n_half;
    if (n_groups > 0) {
      g.grp_atom0 = grp_atom0.data_ptr<int32_t>();
      g.grp_pair0 = grp_pair0.data_ptr<int32_t>();
      g.grp_tile0 = grp_tile0.data_ptr<int32_t>();
      g.n_groups = n_groups;
      g.max_group_atoms = max_group_atoms;
      g.n_tiles_grouped = n_tiles_grouped;
      g.max_group_pairs = max_group_pairs;
    }
    g.filter_pairs = filter_pairs > 0 ? 1 : 0;
    g.edge_pair = edge_pair.defined() ? edge_pair.data_ptr<int32_t>() : nullptr;
    g.blocks = blocks_state == 1 ? &blocks : nullptr;
    g.transposed = has_transposed ? &transposed : nullptr;
    return g;
  }
};

Lru<Plan> g_plans(16);
constexpr int64_t kMaxGroupAtoms = 32;   // one 32-row MFMA tile of atoms per group (spk_schnet_mol.hip)

// Block-diagonal structure of a symmetric list (molecule batches): connected atom ranges that no edge leaves, merged
// greedily into groups of at most 32 atoms.  Plan time only (one small D2H copy).
void build_groups(Plan& p) {
  const int64_t N = p.n_atoms;
  auto dev = p.idx_i.device();
  Tensor ar = at::arange(N, at::TensorOptions().dtype(at::kLong).device(dev));
  Tensor mj = ar.clone();
  mj.scatter_reduce_(0, p.idx_i, p.idx_j, "amax", true);
  Tensor cm = std::get<0>(at::cummax(mj, 0));
  Tensor ends_h = (at::nonzero(cm == ar).flatten() + 1).cpu();
  if (ends_h.numel() == 0) return;
  Tensor sizes = at::diff(ends_h, 1, 0, at::zeros({1}, ends_h.options()));
  if (sizes.max().item<int64_t>() > kMaxGroupAtoms) return;
  const int64_t cap = kMaxGroupAtoms;
  std::vector<int64_t> atom0{0};
  int64_t cur = 0;
  auto sz = sizes.accessor<int64_t, 1>();
  for (int64_t k = 0; k < sizes.numel(); ++k) {
    if (cur + sz[k] > cap && cur > 0) {
      atom0.push_back(atom0.back() + cur);
      cur = 0;
    }
    cur += sz[k];
  }
  atom0.push_back(atom0.back() + cur);
  Tensor atom0_t = at::tensor(atom0, at::TensorOptions().dtype(at::kLong)).to(dev);
  Tensor hi = p.idx_i.index_select(0, p.half.to(at::kLong));
  Tensor pair0 = at::searchsorted(hi, atom0_t).to(at::kInt);
  Tensor tiles = at::floor_divide(at::diff(pair0.to(at::kLong)) + 31, 32);
  Tensor tile0 = at::cat({at::zeros({1}, tiles.options()), at::cumsum(tiles, 0)}).to(at::kInt);
  p.grp_atom0 = atom0_t.to(at::kInt).contiguous();
  p.grp_pair0 = pair0.contiguous();
  p.grp_tile0 = tile0.contiguous();
  p.n_groups = (int32_t)atom0.size() - 1;
  p.max_group_atoms = (int32_t)at::diff(atom0_t).max().item<int64_t>();
  p.n_tiles_grouped = tile0[-1].item<int64_t>();
  p.max_group_pairs = (int32_t)at::diff(pair0.to(at::kLong)).max().item<int64_t>();
}

// Cached plan of a neighbour list (spk_edge_plan: one 16-byte D2H per NEW list; never inside a graph capture --
// callers run one eager call per list first, as GraphedForceCall / the MD loops do).
std::shared_ptr<Plan> get_plan(const Tensor& idx_i_in, const Tensor& idx_j_in, int64_t n_atoms, const Tensor& r_ij) {
  const bool want_groups = true;
  std::vector<uint64_t> key{(uint64_t)idx_i_in.data_ptr(), (uint64_t)idx_j_in.data_ptr(), version_of(idx_i_in), version_of(idx_j_in),
                            (uint64_t)idx_i_in.size(0), (uint64_t)n_atoms, (uint64_t)idx_i_in.device().index(), (uint64_t)r_ij.defined(),
                            (uint64_t)want_groups, (uint64_t)idx_i_in.scalar_type()};
  std::lock_guard<std::mutex> lock(g_mutex);
  if (auto hit = g_plans.get(key)) return hit;
  auto p = std::make_shared<Plan>();
  p->idx_i = i64(idx_i_in, "edge_plan");
  p->idx_j = i64(idx_j_in, "edge_plan");
  p->src_i = idx_i_in;
  p->src_j = idx_j_in;
  p->n_atoms = n_atoms;
  p->n_edges = p->idx_i.size(0);
  p->has_r = r_ij.defined();
  TORCH_CHECK(p->idx_j.size(0) == p->n_edges, "edge_plan: idx_i and idx_j differ in length");
  auto dev = p->idx_i.device();
  c10::DeviceGuard guard(dev);
  auto iopt = at::TensorOptions().dtype(at::kInt).device(dev);
  p->rowptr = at::empty({n_atoms + 1}, iopt);
  p->rev = at::full({std::max<int64_t>(p->n_edges, 1)}, -1, iopt);
  Tensor scratch = at::zeros({4}, iopt);
  int32_t flags[4] = {0, 0, 0, 0};
  Tensor r;
  if (r_ij.defined() && p->n_edges > 0) r = f32(r_ij.detach(), "edge_plan");
  check(spk_edge_plan(p->n_edges ? p->idx_i.data_ptr<int64_t>() : nullptr, p->n_edges ? p->idx_j.data_ptr<int64_t>() : nullptr, fp(r),
                      p->n_edges, n_atoms, p->rowptr.data_ptr<int32_t>(), p->rev.data_ptr<int32_t>(), scratch.data_ptr<int32_t>(), flags,
                      stream_of(p->idx_i)));
  p->sorted = flags[0] != 0;
  p->symmetric = flags[2] != 0;
  if (p->symmetric && p->n_edges > 0) {
    // canonical edge of every undirected pair (e < rev[e]); one-off compaction per list
    Tensor ar = at::arange(p->n_edges, iopt);
    p->half = at::nonzero(p->rev.slice(0, 0, p->n_edges) > ar).flatten().to(at::kInt).contiguous();
    p->n_half = p->half.size(0);
    if (2 * p->n_half != p->n_edges) {
      p->symmetric = false;
      p->half = Tensor();
      p->n_half = 0;
    }
  }
  if (p->sorted && !p->symmetric && r_ij.defined() && p->n_edges >= 4096) {
    // asymmetric list: its by-neighbour copy, so that the backward's scatter over idx_j runs as a row pass (no host sync)
    auto lopt = at::TensorOptions().dtype(at::kLong).device(dev);
    p->t_idx_i = at::empty({p->n_edges}, lopt); p->t_idx_j = at::empty({p->n_edges}, lopt);
    p->t_rowptr = at::zeros({n_atoms + 2}, iopt); p->t_perm = at::empty({p->n_edges}, iopt);
    p->t_rperm = at::empty({p->n_edges, 3}, at::TensorOptions().dtype(at::kFloat).device(dev));
    Tensor tmp = at::empty({std::max<int64_t>(spk_transpose_plan_bytes(p->n_edges, n_atoms), 16)}, at::TensorOptions().dtype(at::kByte).device(dev));
    check(spk_transposed_build(p->idx_i.data_ptr<int64_t>(), p->idx_j.data_ptr<int64_t>(), p->n_edges, n_atoms, p->t_idx_i.data_ptr<int64_t>(),
                               p->t_idx_j.data_ptr<int64_t>(), p->t_rowptr.data_ptr<int32_t>(), p->t_perm.data_ptr<int32_t>(), tmp.data_ptr(),
                               stream_of(p->idx_i)));
    p->transposed.idx_i = p->t_idx_i.data_ptr<int64_t>(); p->transposed.idx_j = p->t_idx_j.data_ptr<int64_t>();
    p->transposed.rowptr = p->t_rowptr.data_ptr<int32_t>(); p->transposed.perm = p->t_perm.data_ptr<int32_t>();
    p->transposed.r_perm = p->t_rperm.data_ptr<float>();
    p->has_transposed = true;
  }
  if (p->symmetric && p->n_half > 0) {
    // position in `half` of the pair of every directed edge (molecule-resident SchNet kernels)
    Tensor k = at::arange(p->n_half, iopt);
    Tensor hl = p->half.to(at::kLong);
    p->edge_pair = at::empty({p->n_edges}, iopt);
    p->edge_pair.index_put_({hl}, k);
    p->edge_pair.index_put_({p->rev.slice(0, 0, p->n_edges).index_select(0, hl).to(at::kLong)}, k);
    if (want_groups) build_groups(*p);
  }
  g_plans.put(key, p);
  return p;
}

// Block plan of a sorted list for the block kernels of the PaiNN message (spk_painn_blk.hip): an opt-in EXPERIMENT (SPK_BLOCKS=1 and
// spk_painn_set_block(1)) -- measured slower than the row / tile kernels on the water box.  Built once per list next to the plan
// (one small D2H copy), keyed by the radial-basis size and the feature width the workspace is sized for.
void ensure_blocks(Plan& p, int64_t n_rbf, int64_t F) {
  const char* env = getenv("SPK_BLOCKS");
  const int force = env ? (env[0] == '0' ? -1 : 1) : 0;
  if (p.blocks_state != 0 && p.blocks_rbf == n_rbf && p.blocks_F == F) return;
  p.blocks_state = -1; p.blocks_rbf = (int)n_rbf; p.blocks_F = (int)F;
  if (force <= 0 || !p.sorted || p.n_edges == 0 || n_rbf > 32 || F % 16 != 0 || F < 16) return;
  int64_t sz[12];
  check(spk_blocks_sizes(p.n_atoms, p.n_edges, (int32_t)n_rbf, (int32_t)F, sz));
  auto dev = p.idx_i.device();
  auto iopt = at::TensorOptions().dtype(at::kInt).device(dev);
  auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
  p.b_sub_n = at::zeros({sz[0]}, iopt); p.b_sub_u = at::zeros({sz[1]}, iopt); p.b_uniq = at::zeros({sz[2]}, iopt);
  p.b_jl = at::zeros({sz[3]}, at::TensorOptions().dtype(at::kShort).device(dev));
  p.b_tile0 = at::zeros({sz[4]}, iopt); p.b_tinfo = at::zeros({sz[5]}, iopt); p.b_desc = at::zeros({sz[10]}, iopt);
  Tensor stats = at::zeros({4}, iopt);
  std::memset(&p.blocks, 0, sizeof(p.blocks));
  p.blocks.sub_n = p.b_sub_n.data_ptr<int32_t>(); p.blocks.sub_u = p.b_sub_u.data_ptr<int32_t>(); p.blocks.uniq = p.b_uniq.data_ptr<int32_t>();
  p.blocks.jl = (const uint16_t*)p.b_jl.data_ptr<int16_t>(); p.blocks.atom_tile0 = p.b_tile0.data_ptr<int32_t>(); p.blocks.tile_info = p.b_tinfo.data_ptr<int32_t>(); p.blocks.blk_desc = p.b_desc.data_ptr<int32_t>();
  int32_t host[4] = {0, 0, 0, 0};
  spk_graph_t g = p.graph();
  check(spk_blocks_build(&g, (int32_t)n_rbf, 0, &p.blocks, stats.data_ptr<int32_t>(), host, stream_of(p.idx_i)));
  if (!p.blocks.ok) { p.b_uniq = Tensor(); p.b_jl = Tensor(); return; }
  const int64_t nt = std::max<int64_t>(p.blocks.n_tiles, 1);
  p.b_apack = at::empty({nt * p.blocks.ks * 64}, fopt); p.b_adpack = at::empty({nt * p.blocks.ks * 64}, fopt);
  p.b_rec = at::empty({nt * 6 * 16}, fopt); p.b_part = at::empty({sz[8]}, fopt);
  p.blocks.apack = p.b_apack.data_ptr<float>(); p.blocks.adpack = p.b_adpack.data_ptr<float>();
  p.blocks.rec = p.b_rec.data_ptr<float>(); p.blocks.part = p.b_part.data_ptr<float>();
  p.blocks_state = 1;
}

// Lists with pairs at or beyond the cutoff (MD skin lists): switch the per-call pair compaction of the fused SchNet path
// on when more than 5 % of the pairs are outside right now (one D2H per list).
void decide_filter(Plan& p, const Tensor& r_ij, double cutoff) {
  if (p.filter_pairs >= 0) return;
  if (p.n_edges == 0 || !p.symmetric) {
    p.filter_pairs = 0;
    return;
  }
  Tensor d = at::linalg_vector_norm(r_ij.detach(), 2, at::IntArrayRef{1});
  p.filter_pairs = (d >= cutoff).to(at::kFloat).mean().item<float>() > 0.05f ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ static-shape mode
// HIP-graph replays of the training step refill the index BUFFERS between replays: no plan cache, no host round trip.
// Declared ascending indices get their CSR row pointers from a device-only kernel (spk_segment_rowptr_i32) launched by
// static_refresh() inside the captured step; every other index takes the atomic scatter.
// Declarations belong to an OWNER (one per StaticLists object = one captured step): a second stepper (another shape bucket, a
// validation step) adds its own entries and never touches the buffers a live captured graph of the first one points at; an
// owner's entries (its row pointers and its error word) are freed only by static_release(owner).
struct StaticEntry { Tensor idx, rowptr; int64_t n_rows; int64_t owner; };
struct StaticRange { Tensor idx; int64_t hi; int64_t owner; };
std::vector<StaticEntry> g_static;
std::vector<StaticRange> g_static_ranges;   // unsorted indices that must lie in [0, hi)
std::map<int64_t, Tensor> g_static_err;     // owner -> device error word
int64_t g_static_next_owner = 1;
bool g_static_on = false;

Tensor static_rowptr(const Tensor& idx, int64_t dim_size) {
  for (auto& e : g_static)
    if (e.idx.data_ptr() == idx.data_ptr() && e.n_rows == dim_size && e.idx.size(0) == idx.size(0)) return e.rowptr;
  return Tensor();
}

// rowptr of an index if it is ascending (plan cache; device-only in static-shape mode), else undefined
Tensor segment_rowptr(const Tensor& idx, int64_t dim_size) {
  if (g_static_on) return static_rowptr(idx, dim_size);
  if (idx.size(0) == 0) return Tensor();
  auto p = get_plan(idx, idx, dim_size, Tensor());
  return p->sorted ? p->rowptr : Tensor();
}

// ------------------------------------------------------------------------------------------------ raw launchers
struct Dim3 { int64_t dim, outer, len, inner; };
Dim3 as_3d(const Tensor& x, int64_t dim) {
  dim = at::maybe_wrap_dim(dim, x.dim());
  int64_t outer = 1, inner = 1;
  for (int64_t d = 0; d < dim; ++d) outer *= x.size(d);
  for (int64_t d = dim + 1; d < x.dim(); ++d) inner *= x.size(d);
  return {dim, outer, x.size(dim), inner};
}

Tensor scatter_add_raw(const Tensor& x_in, const Tensor& idx_in, int64_t dim_size, int64_t dim) {
  Tensor x = f32(x_in, "scatter_add");
  Tensor idx = i64(idx_in, "scatter_add");
  Dim3 s = as_3d(x, dim);
  TORCH_CHECK(idx.dim() == 1 && idx.size(0) == s.len, "scatter_add: idx_i must have ", s.len, " entries, got ", idx.sizes());
  auto shape = x.sizes().vec();
  shape[s.dim] = dim_size;
  c10::DeviceGuard guard(x.device());
  Tensor y = at::empty(shape, x.options());
  Tensor rp = segment_rowptr(idx_in.scalar_type() == at::kLong && idx_in.is_contiguous() ? idx_in : idx, dim_size);
  check(spk_scatter_add_f32(fp(x), idx.data_ptr<int64_t>(), rp.defined() ? rp.data_ptr<int32_t>() : nullptr, s.outer, s.len, s.inner,
                            dim_size, fpm(y), stream_of(x)));
  return y;
}

Tensor gather_raw(const Tensor& x_in, const Tensor& idx_in, int64_t dim) {
  Tensor x = f32(x_in, "gather");
  Tensor idx = i64(idx_in, "gather");
  Dim3 s = as_3d(x, dim);
  auto shape = x.sizes().vec();
  shape[s.dim] = idx.size(0);
  c10::DeviceGuard guard(x.device());
  Tensor y = at::empty(shape, x.options());
  check(spk_gather_f32(fp(x), idx.data_ptr<int64_t>(), s.outer, s.len, idx.size(0), s.inner, fpm(y), stream_of(x)));
  return y;
}

Tensor pairwise_raw(const Tensor& R_in, const Tensor& idx_i_in, const Tensor& idx_j_in, const c10::optional<Tensor>& offsets) {
  Tensor R = f32(R_in, "pairwise");
  Tensor ii = i64(idx_i_in, "pairwise"), jj = i64(idx_j_in, "pairwise");
  Tensor off = opt_f32(offsets, "pairwise");
  const int64_t E = ii.size(0);
  c10::DeviceGuard guard(R.device());
  Tensor r = at::empty({E, 3}, R.options());
  if (E > 0 && R.size(0) > 0)
    check(spk_pairwise_n_f32(fp(R), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), fp(off), E, R.size(0), fpm(r), stream_of(R)));
  return r;
}

// dE/dR from dE/dr_ij: segmented row sum on sorted + symmetric lists (plan), float atomics otherwise
Tensor pairwise_bwd_raw(const Tensor& gr_in, const Tensor& idx_i_in, const Tensor& idx_j_in, int64_t n_atoms) {
  Tensor gr = f32(gr_in, "pairwise_backward");
  Tensor ii = i64(idx_i_in, "pairwise_backward"), jj = i64(idx_j_in, "pairwise_backward");
  c10::DeviceGuard guard(gr.device());
  Tensor gR = at::empty({n_atoms, 3}, gr.options());
  std::shared_ptr<Plan> plan;
  if (!g_static_on && ii.size(0) > 0) {
    // only a plan that already knows the geometry (built by the representation of this list) has the reverse map;
    // the vector handed to this function is a GRADIENT, never geometry: a plan is looked up, not built, from it
    std::vector<uint64_t> key{(uint64_t)idx_i_in.data_ptr(), (uint64_t)idx_j_in.data_ptr(), version_of(idx_i_in), version_of(idx_j_in),
                              (uint64_t)idx_i_in.size(0), (uint64_t)n_atoms, (uint64_t)idx_i_in.device().index(), 1, 1,
                              (uint64_t)idx_i_in.scalar_type()};
    std::lock_guard<std::mutex> lock(g_mutex);
    plan = g_plans.get(key);
  }
  if (plan && plan->sorted && plan->symmetric) {
    spk_graph_t g = plan->graph();
    check(spk_pairwise_bwd_graph_f32(fp(gr), &g, fpm(gR), stream_of(gr)));
  } else {
    check(spk_pairwise_bwd_f32(fp(gr), ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), ii.size(0), n_atoms, fpm(gR), stream_of(gr)));
  }
  return gR;
}

spk_radial_t radial_of(int64_t kind, const Tensor& p0, const Tensor& p1, double cutoff) {
  spk_radial_t rb;
  rb.kind = (int32_t)kind;
  rb.n_rbf = (int32_t)p0.size(0);
  rb.p0 = fp(p0);
  rb.p1 = fp(p1);
  rb.cutoff = (float)cutoff;
  return rb;
}

std::tuple<Tensor, Tensor> radial_cutoff_raw(const Tensor& d_in, int64_t kind, const Tensor& p0_in, const c10::optional<Tensor>& p1_in,
                                             double cutoff, bool want_phi, bool want_cut) {
  Tensor d = f32(d_in, "radial_cutoff");
  Tensor p0 = f32(p0_in, "radial_cutoff"), p1 = opt_f32(p1_in, "radial_cutoff");
  c10::DeviceGuard guard(d.device());
  auto shape = d.sizes().vec();
  Tensor fc = want_cut ? at::empty(shape, d.options()) : Tensor();
  shape.push_back(p0.size(0));
  Tensor phi = want_phi ? at::empty(shape, d.options()) : Tensor();
  spk_radial_t rb = radial_of(kind, p0, p1, cutoff);
  check(spk_radial_cutoff_f32(fp(d), d.numel(), &rb, fpm(phi), fpm(fc), stream_of(d)));
  return {phi.defined() ? phi : at::empty({0}, d.options()), fc.defined() ? fc : at::empty({0}, d.options())};
}

Tensor radial_cutoff_bwd_raw(const Tensor& d_in, int64_t kind, const Tensor& p0_in, const c10::optional<Tensor>& p1_in, double cutoff,
                             const c10::optional<Tensor>& gphi_in, const c10::optional<Tensor>& gfc_in) {
  Tensor d = f32(d_in, "radial_cutoff_backward");
  Tensor p0 = f32(p0_in, "radial_cutoff_backward"), p1 = opt_f32(p1_in, "radial_cutoff_backward");
  Tensor gphi = opt_f32(gphi_in, "radial_cutoff_backward"), gfc = opt_f32(gfc_in, "radial_cutoff_backward");
  c10::DeviceGuard guard(d.device());
  Tensor gd = at::empty_like(d);
  spk_radial_t rb = radial_of(kind, p0, p1, cutoff);
  check(spk_radial_cutoff_bwd_f32(fp(d), d.numel(), &rb, fp(gphi), fp(gfc), fpm(gd), stream_of(d)));
  return gd;
}

// y = act(x w^T + b); returns (y, pre) -- pre is the pre-activation (empty when act == NONE)
std::tuple<Tensor, Tensor> dense_raw(const Tensor& x_in, const Tensor& w_in, const c10::optional<Tensor>& b_in, int64_t act) {
  Tensor x = f32(x_in, "dense"), w = f32(w_in, "dense"), b = opt_f32(b_in, "dense");
  const int64_t k = x.size(-1), n_out = w.size(0);
  TORCH_CHECK(w.dim() == 2 && w.size(1) == k, "dense: weight ", w.sizes(), " does not match input width ", k);
  const int64_t m = k > 0 ? x.numel() / k : 0;
  c10::DeviceGuard guard(x.device());
  auto shape = x.sizes().vec();
  shape.back() = n_out;
  Tensor y = at::empty(shape, x.options());
  Tensor pre = act != SPK_ACT_NONE ? at::empty(shape, x.options()) : Tensor();
  check(spk_dense_f32(fp(x), fp(w), fp(b), nullptr, fpm(y), fpm(pre), m, (int32_t)k, (int32_t)n_out, (int32_t)act, stream_of(x)));
  return {y, pre.defined() ? pre : at::empty({0}, x.options())};
}

Tensor dense_bwd_input_raw(const Tensor& gy_in, const Tensor& pre_in, const Tensor& w_in, int64_t act) {
  Tensor gy = f32(gy_in, "dense_backward_input"), w = f32(w_in, "dense_backward_input");
  const int64_t n_out = w.size(0), k = w.size(1);
  const int64_t m = n_out > 0 ? gy.numel() / n_out : 0;
  Tensor pre = act != SPK_ACT_NONE ? f32(pre_in, "dense_backward_input") : Tensor();
  c10::DeviceGuard guard(gy.device());
  auto shape = gy.sizes().vec();
  shape.back() = k;
  Tensor dx = at::empty(shape, gy.options());
  check(spk_dense_bwd_input_f32(fp(gy), fp(pre), fp(w), nullptr, fpm(dx), m, (int32_t)k, (int32_t)n_out, (int32_t)act, stream_of(gy)));
  return dx;
}

// ------------------------------------------------------------------------------------------------ model parameter blocks
// Device-pointer blocks of the C ABI (spk_schnet_t / spk_painn_t) built from the state_dict tensors, with the cached
// transposed copies and packed images of the atom-wise weights; rebuilt when a parameter changes (data_ptr / version).
struct SchnetModel {
  std::vector<Tensor> keep;
  std::vector<spk_schnet_layer_t> layers;
  spk_schnet_t m;
};
struct PainnModel {
  std::vector<Tensor> keep;
  std::vector<spk_painn_layer_t> layers;
  spk_painn_t m;
};
Lru<SchnetModel> g_schnet(8);
Lru<PainnModel> g_painn(8);

// Parameters updated IN PLACE by a captured optimizer step keep their data_ptr AND their version counter (graph replays do not
// go through the tensor's bookkeeping), so identity + version alone would keep serving transposed / packed copies of the weights
// as they were before the replays.  Whoever changes parameters behind torch's back (train.GraphedTrainStep after every replay)
// calls weights_changed(), which moves every weight-derived cache entry out of reach.
std::atomic<uint64_t> g_weight_generation{0};

std::vector<uint64_t> weights_key(at::TensorList ws, std::initializer_list<int64_t> extra) {
  std::vector<uint64_t> key;
  key.reserve(2 * ws.size() + extra.size() + 1);
  key.push_back(g_weight_generation.load(std::memory_order_acquire));
  for (const auto& w : ws) {
    key.push_back((uint64_t)w.data_ptr());
    key.push_back(version_of(w));
  }
  for (auto e : extra) key.push_back((uint64_t)e);
  return key;
}

constexpr int kSchnetPerLayer = 9;   // in2f.w | filter_network.0.{w,b} | filter_network.1.{w,b} | f2out.0.{w,b} | f2out.1.{w,b}
std::shared_ptr<SchnetModel> get_schnet(at::TensorList ws, int64_t F, int64_t nf) {
  TORCH_CHECK(ws.size() % kSchnetPerLayer == 0, "schnet: expected 9 weight tensors per interaction, got ", ws.size());
  const int64_t L = ws.size() / kSchnetPerLayer;
  auto key = weights_key(ws, {F, nf});
  std::lock_guard<std::mutex> lock(g_mutex);
  if (auto hit = g_schnet.get(key)) return hit;
  auto M = std::make_shared<SchnetModel>();
  M->layers.resize(std::max<int64_t>(L, 1));
  for (int64_t l = 0; l < L; ++l) {
    Tensor t[kSchnetPerLayer];
    for (int k = 0; k < kSchnetPerLayer; ++k) {
      t[k] = f32(ws[l * kSchnetPerLayer + k].detach(), "schnet weights");
      M->keep.push_back(t[k]);
    }
    Tensor in2fT = t[0].t().contiguous(), w1T = t[5].t().contiguous(), w2T = t[7].t().contiguous();
    M->keep.insert(M->keep.end(), {in2fT, w1T, w2T});
    spk_schnet_layer_t& P = M->layers[l];
    P.in2f_w = fp(t[0]); P.fn_w1 = fp(t[1]); P.fn_b1 = fp(t[2]); P.fn_w2 = fp(t[3]); P.fn_b2 = fp(t[4]);
    P.f2out_w1 = fp(t[5]); P.f2out_b1 = fp(t[6]); P.f2out_w2 = fp(t[7]); P.f2out_b2 = fp(t[8]);
    P.in2f_wT = fp(in2fT); P.f2out_w1T = fp(w1T); P.f2out_w2T = fp(w2T);
    if (spk_filter_table_drop_if_stale(P.fn_w2, 1 + version_of(t[1]) + version_of(t[2]) + version_of(t[3]) + version_of(t[4])))      // (versions only grow: their sum names the state of the four filter-network tensors)
      TORCH_WARN("spk_hip: the filter table of a SchNet interaction was built from other weights (version changed); dropped -- the exact filter network runs");
  }
  M->m.n_atom_basis = (int32_t)F; M->m.n_filters = (int32_t)nf; M->m.n_interactions = (int32_t)L; M->m.reserved = 0;
  M->m.layers = M->layers.data(); M->m.wpack = nullptr;
  const int64_t n_pack = L > 0 ? spk_schnet_packed_floats(&M->m) : 0;
  if (n_pack > 0 && !getenv("SPK_NO_PACK")) {
    Tensor wpack = at::empty({n_pack}, ws[0].options().dtype(at::kFloat));
    c10::DeviceGuard guard(wpack.device());
    check(spk_schnet_pack_weights_f32(&M->m, fpm(wpack), stream_of(wpack)));
    M->m.wpack = fp(wpack);
    M->keep.push_back(wpack);
  }
  g_schnet.put(key, M);
  return M;
}

// per interaction: interatomic_context_net.0.{w,b} | .1.{w,b} | mixing mu_channel_mix.w | intraatomic_context_net.0.{w,b} | .1.{w,b};
// the last two tensors of the list are filter_net.{weight, bias}
constexpr int kPainnPerLayer = 9;
std::shared_ptr<PainnModel> get_painn(at::TensorList ws, int64_t F, bool shared_filters, double eps) {
  TORCH_CHECK(ws.size() >= 2 && (ws.size() - 2) % kPainnPerLayer == 0, "painn: expected 9 weight tensors per interaction + filter_net.{weight,bias}, got ", ws.size());
  const int64_t L = (ws.size() - 2) / kPainnPerLayer;
  float epsf = (float)eps;
  uint32_t eps_bits;
  std::memcpy(&eps_bits, &epsf, 4);
  auto key = weights_key(ws, {F, (int64_t)shared_filters, (int64_t)eps_bits});
  std::lock_guard<std::mutex> lock(g_mutex);
  if (auto hit = g_painn.get(key)) return hit;
  auto M = std::make_shared<PainnModel>();
  M->layers.resize(std::max<int64_t>(L, 1));
  Tensor fw = f32(ws[ws.size() - 2].detach(), "painn weights"), fb = f32(ws[ws.size() - 1].detach(), "painn weights");
  M->keep.insert(M->keep.end(), {fw, fb});
  const int64_t n_rbf = fw.size(1);
  for (int64_t l = 0; l < L; ++l) {
    Tensor t[kPainnPerLayer];
    for (int k = 0; k < kPainnPerLayer; ++k) {
      t[k] = f32(ws[l * kPainnPerLayer + k].detach(), "painn weights");
      M->keep.push_back(t[k]);
    }
    Tensor c1T = t[0].t().contiguous(), c2T = t[2].t().contiguous(), mxT = t[4].t().contiguous(), i1T = t[5].t().contiguous(), i2T = t[7].t().contiguous();
    M->keep.insert(M->keep.end(), {c1T, c2T, mxT, i1T, i2T});
    spk_painn_layer_t& P = M->layers[l];
    P.ctx_w1 = fp(t[0]); P.ctx_b1 = fp(t[1]); P.ctx_w2 = fp(t[2]); P.ctx_b2 = fp(t[3]); P.mix_w = fp(t[4]);
    P.ictx_w1 = fp(t[5]); P.ictx_b1 = fp(t[6]); P.ictx_w2 = fp(t[7]); P.ictx_b2 = fp(t[8]);
    P.ctx_w1T = fp(c1T); P.ctx_w2T = fp(c2T); P.mix_wT = fp(mxT); P.ictx_w1T = fp(i1T); P.ictx_w2T = fp(i2T);
    const int64_t row0 = shared_filters ? 0 : 3 * F * l;
    P.filt_w = fp(fw) + row0 * n_rbf;
    P.filt_b = fp(fb) + row0;
    if (spk_filter_table_drop_if_stale(P.filt_w, 1 + version_of(fw) + version_of(fb)))
      TORCH_WARN("spk_hip: the filter table of a PaiNN interaction was built from other weights (version changed); dropped -- the exact filter runs");
  }
  M->m.n_atom_basis = (int32_t)F; M->m.n_interactions = (int32_t)L; M->m.epsilon = epsf; M->m.reserved = 0;
  M->m.layers = M->layers.data(); M->m.wpack = nullptr;
  const int64_t n_pack = L > 0 ? spk_painn_packed_floats(&M->m) : 0;
  if (n_pack > 0 && !getenv("SPK_NO_PACK")) {
    Tensor wpack = at::empty({n_pack}, fw.options());
    c10::DeviceGuard guard(wpack.device());
    check(spk_painn_pack_weights_f32(&M->m, fpm(wpack), stream_of(wpack)));
    M->m.wpack = fp(wpack);
    M->keep.push_back(wpack);
  }
  g_painn.put(key, M);
  return M;
}

// ------------------------------------------------------------------------------------------------ fused SchNet (raw)
// (x_out, saved, scratch) = schnet_forward(...): saved / scratch travel to schnet_backward
std::tuple<Tensor, Tensor, Tensor> schnet_forward_raw(const Tensor& x0_in, const Tensor& r_in, const Tensor& idx_i, const Tensor& idx_j,
                                                      at::TensorList ws, int64_t n_filters, int64_t rbf_kind, const Tensor& p0_in,
                                                      const c10::optional<Tensor>& p1_in, double cutoff, bool save_filters) {
  Tensor x0 = f32(x0_in, "SchNet"), r = f32(r_in, "SchNet");
  Tensor p0 = f32(p0_in, "SchNet"), p1 = opt_f32(p1_in, "SchNet");
  const int64_t N = x0.size(0), F = x0.size(1);
  c10::DeviceGuard guard(x0.device());
  auto plan = get_plan(idx_i, idx_j, N, r);
  decide_filter(*plan, r, cutoff);
  auto M = get_schnet(ws, F, n_filters);
  spk_schnet_t m = M->m;            // per-call copy: the cached block is never mutated
  m.reserved = save_filters ? 1 : 0;
  spk_graph_t g = plan->graph();
  spk_radial_t rb = radial_of(rbf_kind, p0, p1, cutoff);
  const int64_t n_saved = save_filters ? spk_schnet_saved_floats_graph(&m, &g, &rb) : spk_schnet_saved_floats(&m, N);
  Tensor out = at::empty({N, F}, x0.options());
  Tensor saved = at::empty({std::max<int64_t>(1, n_saved)}, x0.options());
  Tensor scratch = at::empty({std::max<int64_t>(1, spk_schnet_scratch_floats(&m, N))}, x0.options());
  check(spk_schnet_forward_f32(&m, &g, &rb, fp(x0), fp(r), fpm(out), fpm(saved), fpm(scratch), stream_of(x0)));
  return {out, saved, scratch};
}

std::tuple<Tensor, Tensor> schnet_backward_raw(const Tensor& gx_in, const Tensor& r_in, const Tensor& saved, const Tensor& scratch_in,
                                               const Plan& plan, at::TensorList ws, int64_t F, int64_t n_filters, int64_t rbf_kind,
                                               const Tensor& p0_in, const c10::optional<Tensor>& p1_in, double cutoff, bool saved_filters,
                                               bool want_gx0) {
  Tensor gx = f32(gx_in, "SchNet backward"), r = f32(r_in, "SchNet backward");
  Tensor p0 = f32(p0_in, "SchNet"), p1 = opt_f32(p1_in, "SchNet");
  const int64_t N = gx.size(0);
  c10::DeviceGuard guard(gx.device());
  auto M = get_schnet(ws, F, n_filters);
  spk_schnet_t m = M->m;
  m.reserved = saved_filters ? 1 : 0;
  spk_graph_t g = plan.graph();
  spk_radial_t rb = radial_of(rbf_kind, p0, p1, cutoff);
  Tensor scratch = scratch_in;
  Tensor gr = at::empty_like(r);
  Tensor gx0 = want_gx0 ? at::empty({N, F}, gx.options()) : Tensor();
  check(spk_schnet_backward_f32(&m, &g, &rb, fp(gx), fp(r), fp(saved), fpm(scratch), fpm(gr), fpm(gx0), stream_of(gx)));
  return {gr, gx0};
}

// ------------------------------------------------------------------------------------------------ fused PaiNN (raw)
std::tuple<Tensor, Tensor, Tensor, Tensor> painn_forward_raw(const Tensor& q0_in, const Tensor& r_in, const Tensor& idx_i, const Tensor& idx_j,
                                                             at::TensorList ws, bool shared_filters, double eps, int64_t rbf_kind,
                                                             const Tensor& p0_in, const c10::optional<Tensor>& p1_in, double cutoff,
                                                             std::shared_ptr<Plan>* plan_out) {
  Tensor q0 = f32(q0_in, "PaiNN"), r = f32(r_in, "PaiNN");
  Tensor p0 = f32(p0_in, "PaiNN"), p1 = opt_f32(p1_in, "PaiNN");
  const int64_t N = q0.size(0), F = q0.size(1);
  c10::DeviceGuard guard(q0.device());
  auto plan = get_plan(idx_i, idx_j, N, r);
  if (plan->filter_pairs < 0 && plan->n_edges >= (1 << 19))
    decide_filter(*plan, r, cutoff);   // large lists: tells the message dispatch whether the list carries a skin
  ensure_blocks(*plan, p0.size(0), F);
  auto M = get_painn(ws, F, shared_filters, eps);
  spk_graph_t g = plan->graph();
  spk_radial_t rb = radial_of(rbf_kind, p0, p1, cutoff);
  Tensor q = at::empty({N, F}, q0.options()), mu = at::empty({N, 3, F}, q0.options());
  Tensor saved = at::empty({std::max<int64_t>(1, spk_painn_saved_floats(&M->m, N))}, q0.options());
  Tensor scratch = at::empty({std::max<int64_t>(1, spk_painn_scratch_floats(&M->m, N))}, q0.options());
  check(spk_painn_forward_f32(&M->m, &g, &rb, fp(q0), fp(r), fpm(q), fpm(mu), fpm(saved), fpm(scratch), stream_of(q0)));
  if (plan_out) *plan_out = plan;
  return {q, mu, saved, scratch};
}

std::tuple<Tensor, Tensor> painn_backward_raw(const Tensor& gq_in, const Tensor& gmu_in, const Tensor& r_in, const Tensor& saved,
                                              const Tensor& scratch_in, const Plan& plan, at::TensorList ws, int64_t F, bool shared_filters,
                                              double eps, int64_t rbf_kind, const Tensor& p0_in, const c10::optional<Tensor>& p1_in,
                                              double cutoff, bool want_gq0) {
  Tensor r = f32(r_in, "PaiNN backward");
  Tensor p0 = f32(p0_in, "PaiNN"), p1 = opt_f32(p1_in, "PaiNN");
  const int64_t N = plan.n_atoms;
  c10::DeviceGuard guard(r.device());
  Tensor gq = gq_in.defined() ? f32(gq_in, "PaiNN backward") : Tensor();
  Tensor gmu = gmu_in.defined() ? f32(gmu_in, "PaiNN backward") : Tensor();
  if (!gq.defined() && !gmu.defined()) gq = at::zeros({N, F}, r.options());
  auto M = get_painn(ws, F, shared_filters, eps);
  spk_graph_t g = plan.graph();
  spk_radial_t rb = radial_of(rbf_kind, p0, p1, cutoff);
  Tensor scratch = scratch_in;
  Tensor gr = at::empty_like(r);
  Tensor gq0 = want_gq0 ? at::empty({N, F}, r.options()) : Tensor();
  check(spk_painn_backward_f32(&M->m, &g, &rb, fp(gq), fp(gmu), fp(r), fp(saved), fpm(scratch), fpm(gr), fpm(gq0), stream_of(r)));
  return {gr, gq0};
}

// ------------------------------------------------------------------------------------------------ fused Atomwise head (raw)
std::tuple<Tensor, Tensor, Tensor> atomwise_forward_raw(const Tensor& x_in, const Tensor& w1_in, const c10::optional<Tensor>& b1_in,
                                                        const Tensor& w2_in, const c10::optional<Tensor>& b2_in, const Tensor& idx_m_in,
                                                        int64_t n_mol, int64_t act) {
  Tensor x = f32(x_in, "Atomwise"), w1 = f32(w1_in, "Atomwise"), w2 = f32(w2_in, "Atomwise").reshape({-1});
  Tensor b1 = opt_f32(b1_in, "Atomwise"), b2 = opt_f32(b2_in, "Atomwise");
  Tensor idx_m = i64(idx_m_in, "Atomwise");
  const int64_t N = x.size(0), n_in = x.size(1), H = w1.size(0);
  TORCH_CHECK(spk_atomwise_supported((int32_t)n_in, (int32_t)H, (int32_t)act), "atomwise: the fused head covers n_in % 32 == 0, n_hidden % 32 == 0, ssp / silu");
  c10::DeviceGuard guard(x.device());
  Tensor pre = at::empty({N, H}, x.options()), y_atom = at::empty({N, 1}, x.options()), E = at::empty({n_mol}, x.options());
  check(spk_atomwise_fwd_f32(fp(x), fp(w1), fp(b1), fp(w2), fp(b2), idx_m.data_ptr<int64_t>(), N, (int32_t)n_in, (int32_t)H, (int32_t)act,
                             n_mol, fpm(pre), fpm(y_atom), fpm(E), stream_of(x)));
  return {E, y_atom, pre};
}

Tensor atomwise_backward_raw(const Tensor& gE_in, const Tensor& gy_in, const Tensor& pre, const Tensor& w1_in, const Tensor& w2_in,
                             const Tensor& idx_m_in, int64_t n_mol, int64_t act) {
  Tensor w1 = f32(w1_in, "Atomwise backward"), w2 = f32(w2_in, "Atomwise backward").reshape({-1});
  Tensor idx_m = i64(idx_m_in, "Atomwise backward");
  const int64_t N = pre.size(0), H = w1.size(0), n_in = w1.size(1);
  c10::DeviceGuard guard(pre.device());
  Tensor gE = gE_in.defined() ? f32(gE_in, "Atomwise backward") : Tensor();
  Tensor gy = gy_in.defined() ? f32(gy_in, "Atomwise backward").reshape({-1}) : Tensor();
  Tensor gx = at::empty({N, n_in}, pre.options());
  if (!gE.defined() && !gy.defined()) return gx.zero_();
  check(spk_atomwise_bwd_f32(fp(gE), fp(gy), fp(pre), fp(w1), fp(w2), idx_m.data_ptr<int64_t>(), N, (int32_t)n_in, (int32_t)H, (int32_t)act,
                             n_mol, fpm(gx), stream_of(pre)));
  return gx;
}

// ------------------------------------------------------------------------------------------------ the standard potential (raw)
// PairwiseDistances -> SchNet -> Atomwise(sum) as ONE forward and ONE backward launch where the molecule-resident kernels cover
// the list (spk_schnet_potential_*); everywhere else the same three stages through their own launchers -- the operator always
// works, the fusion is a property of (model, list).
struct HeadCache { Tensor w1, w1t, b1, w2, b2; };
Lru<HeadCache> g_heads(8);
std::shared_ptr<HeadCache> get_head(const Tensor& w1_in, const c10::optional<Tensor>& b1_in, const Tensor& w2_in, const c10::optional<Tensor>& b2_in) {
  std::vector<uint64_t> key{(uint64_t)w1_in.data_ptr(), version_of(w1_in), (uint64_t)w2_in.data_ptr(), version_of(w2_in),
                            (uint64_t)((b1_in.has_value() && b1_in->defined()) ? b1_in->data_ptr() : nullptr),
                            (b1_in.has_value() && b1_in->defined()) ? version_of(*b1_in) : 0,
                            (uint64_t)((b2_in.has_value() && b2_in->defined()) ? b2_in->data_ptr() : nullptr),
                            (b2_in.has_value() && b2_in->defined()) ? version_of(*b2_in) : 0,
                            g_weight_generation.load(std::memory_order_acquire)};
  std::lock_guard<std::mutex> lock(g_mutex);
  if (auto hit = g_heads.get(key)) return hit;
  auto h = std::make_shared<HeadCache>();
  h->w1 = f32(w1_in.detach(), "potential head");
  h->w1t = h->w1.t().contiguous();
  h->w2 = f32(w2_in.detach(), "potential head").reshape({-1});
  h->b1 = opt_f32(b1_in, "potential head");
  h->b2 = opt_f32(b2_in, "potential head");
  if (h->b1.defined()) h->b1 = h->b1.detach();
  if (h->b2.defined()) h->b2 = h->b2.detach();
  g_heads.put(key, h);
  return h;
}

std::shared_ptr<Plan> find_plan(const Tensor& idx_i, const Tensor& idx_j, int64_t n_atoms) {
  std::vector<uint64_t> key{(uint64_t)idx_i.data_ptr(), (uint64_t)idx_j.data_ptr(), version_of(idx_i), version_of(idx_j), (uint64_t)idx_i.size(0),
                            (uint64_t)n_atoms, (uint64_t)idx_i.device().index(), 1, 1, (uint64_t)idx_i.scalar_type()};
  std::lock_guard<std::mutex> lock(g_mutex);
  return g_plans.get(key);
}

struct PotentialCall {
  std::shared_ptr<Plan> plan;
  std::shared_ptr<SchnetModel> M;
  std::shared_ptr<HeadCache> H;
  spk_schnet_t m;
  spk_head_t head;
  spk_graph_t g;
  spk_radial_t rb;
  Tensor p0, p1;
  bool fused;
};
PotentialCall potential_setup(const Tensor& R, const c10::optional<Tensor>& offsets, const Tensor& idx_i, const Tensor& idx_j, int64_t N, int64_t F,
                              at::TensorList ws, at::TensorList head, int64_t n_filters, int64_t rbf_kind, const Tensor& p0_in,
                              const c10::optional<Tensor>& p1_in, double cutoff, int64_t head_act) {
  TORCH_CHECK(head.size() == 4, "schnet_potential: head = [outnet.0.weight, outnet.0.bias, outnet.1.weight, outnet.1.bias]");
  PotentialCall c;
  c.plan = find_plan(idx_i, idx_j, N);
  if (!c.plan || c.plan->filter_pairs < 0) {       // first call on this list: the plan needs the geometry once
    Tensor r = pairwise_raw(R.detach(), idx_i, idx_j, offsets);
    c.plan = get_plan(idx_i, idx_j, N, r);
    decide_filter(*c.plan, r, cutoff);
  }
  c.M = get_schnet(ws, F, n_filters);
  c.H = get_head(head[0], head[1], head[2], head[3]);
  c.p0 = f32(p0_in, "schnet_potential");
  c.p1 = opt_f32(p1_in, "schnet_potential");
  c.m = c.M->m;
  c.m.reserved = 1;
  c.g = c.plan->graph();
  c.rb = radial_of(rbf_kind, c.p0, c.p1, cutoff);
  c.head.w1 = fp(c.H->w1); c.head.w1t = fp(c.H->w1t); c.head.b1 = fp(c.H->b1); c.head.w2 = fp(c.H->w2); c.head.b2 = fp(c.H->b2);
  c.head.n_hidden = (int32_t)c.H->w1.size(0); c.head.act = (int32_t)head_act;
  c.fused = c.H->w1.dim() == 2 && c.H->w1.size(1) == F && c.H->w2.numel() == c.H->w1.size(0) && c.H->b1.defined() &&
            spk_schnet_potential_supported(&c.m, &c.head, &c.g, &c.rb) != 0;
  return c;
}

// -> (E [n_mol], scalar_representation [N, F], saved, pre_h); `saved` / `pre_h` are what the backward consumes
std::tuple<Tensor, Tensor, Tensor, Tensor> schnet_potential_forward_raw(const Tensor& x0_in, const Tensor& R_in, const c10::optional<Tensor>& offsets_in,
                                                                        const Tensor& idx_i, const Tensor& idx_j, const Tensor& idx_m_in, int64_t n_mol,
                                                                        at::TensorList ws, at::TensorList head, int64_t n_filters, int64_t rbf_kind,
                                                                        const Tensor& p0, const c10::optional<Tensor>& p1, double cutoff,
                                                                        int64_t head_act) {
  Tensor x0 = f32(x0_in, "schnet_potential"), R = f32(R_in, "schnet_potential");
  Tensor off = opt_f32(offsets_in, "schnet_potential");
  Tensor idx_m = i64(idx_m_in, "schnet_potential");
  const int64_t N = x0.size(0), F = x0.size(1);
  TORCH_CHECK(R.dim() == 2 && R.size(0) == N && R.size(1) == 3, "schnet_potential: positions ", R.sizes(), " do not match ", N, " atoms");
  c10::DeviceGuard guard(x0.device());
  PotentialCall c = potential_setup(R, offsets_in, idx_i, idx_j, N, F, ws, head, n_filters, rbf_kind, p0, p1, cutoff, head_act);
  if (c.fused) {
    Tensor x = at::empty({N, F}, x0.options()), E = at::empty({n_mol}, x0.options());
    Tensor pre_h = at::empty({N, (int64_t)c.head.n_hidden}, x0.options());
    Tensor saved = at::empty({std::max<int64_t>(1, spk_schnet_saved_floats_graph(&c.m, &c.g, &c.rb))}, x0.options());
    check(spk_schnet_potential_forward_f32(&c.m, &c.head, &c.g, &c.rb, fp(x0), fp(R), fp(off), idx_m.data_ptr<int64_t>(), n_mol, fpm(x), fpm(E),
                                           fpm(pre_h), fpm(saved), stream_of(x0)));
    return {E, x, saved, pre_h};
  }
  Tensor r = pairwise_raw(R, idx_i, idx_j, offsets_in);
  auto rep = schnet_forward_raw(x0, r, idx_i, idx_j, ws, n_filters, rbf_kind, p0, p1, cutoff, true);
  auto hd = atomwise_forward_raw(std::get<0>(rep), head[0], head[1], head[2], head[3], idx_m, n_mol, head_act);
  return {std::get<0>(hd), std::get<0>(rep), std::get<1>(rep), std::get<2>(hd)};
}

// Does every molecule of idx_m lie inside ONE group of the plan, and has every molecule an atom?  (One D2H per (list, idx_m);
// then the energy head stores its sums instead of accumulating them and the energy buffer needs no clearing launch.)
bool molecules_inside_groups(Plan& p, const Tensor& idx_m, int64_t n_mol) {
  std::vector<uint64_t> key{(uint64_t)idx_m.data_ptr(), version_of(idx_m), (uint64_t)idx_m.size(0), (uint64_t)n_mol};
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto& e : p.inside)
      if (e.first == key) return e.second;
  }
  bool ok = p.n_groups > 0 && idx_m.size(0) == p.n_atoms && n_mol > 0;
  if (ok) {
    Tensor starts = p.grp_atom0.slice(0, 1, p.n_groups).to(at::kLong);                 // first atom of groups 1 .. G-1
    Tensor split = starts.numel() > 0 ? (idx_m.index_select(0, starts) == idx_m.index_select(0, starts - 1)).any() : at::zeros({}, idx_m.options().dtype(at::kBool));
    Tensor counts = at::bincount(idx_m.clamp(0, n_mol), {}, n_mol + 1);
    Tensor good = (~split) & (counts.slice(0, 0, n_mol) > 0).all() & (counts.slice(0, n_mol, n_mol + 1) == 0).all() & (idx_m >= 0).all();
    ok = good.item<bool>();
  }
  std::lock_guard<std::mutex> lock(g_mutex);
  if (p.inside.size() >= 4) p.inside.erase(p.inside.begin());
  p.inside.emplace_back(key, ok);
  return ok;
}

// -> (dL/dR [N, 3], dL/dx0 [N, F] or empty)
std::tuple<Tensor, Tensor> schnet_potential_backward_raw(const c10::optional<Tensor>& gE_in, const c10::optional<Tensor>& gx_in, const Tensor& x0_in,
                                                         const Tensor& R_in, const c10::optional<Tensor>& offsets_in, const Tensor& idx_i,
                                                         const Tensor& idx_j, const Tensor& idx_m_in, int64_t n_mol, const Tensor& saved,
                                                         const Tensor& pre_h, at::TensorList ws, at::TensorList head, int64_t n_filters,
                                                         int64_t rbf_kind, const Tensor& p0, const c10::optional<Tensor>& p1, double cutoff,
                                                         int64_t head_act, bool want_gx0) {
  Tensor R = f32(R_in, "schnet_potential backward");
  Tensor off = opt_f32(offsets_in, "schnet_potential backward");
  Tensor idx_m = i64(idx_m_in, "schnet_potential backward");
  const int64_t N = x0_in.size(0), F = x0_in.size(1);
  c10::DeviceGuard guard(R.device());
  Tensor gE = (gE_in.has_value() && gE_in->defined()) ? f32(*gE_in, "schnet_potential backward") : at::zeros({n_mol}, R.options());
  Tensor gx = (gx_in.has_value() && gx_in->defined()) ? f32(*gx_in, "schnet_potential backward") : Tensor();
  PotentialCall c = potential_setup(R, offsets_in, idx_i, idx_j, N, F, ws, head, n_filters, rbf_kind, p0, p1, cutoff, head_act);
  if (c.fused) {
    Tensor gR = at::empty({N, 3}, R.options());
    Tensor gx0 = want_gx0 ? at::empty({N, F}, R.options()) : Tensor();
    check(spk_schnet_potential_backward_f32(&c.m, &c.head, &c.g, &c.rb, fp(gE), fp(gx), fp(R), fp(off), idx_m.data_ptr<int64_t>(), fp(pre_h), fp(saved),
                                            fpm(gR), fpm(gx0), stream_of(R)));
    return {gR, gx0.defined() ? gx0 : at::empty({0}, R.options())};
  }
  Tensor r = pairwise_raw(R, idx_i, idx_j, offsets_in);
  Tensor gxh = atomwise_backward_raw(gE, Tensor(), pre_h, head[0], head[2], idx_m, n_mol, head_act);
  if (gx.defined()) gxh = gxh + gx;
  Tensor scratch = at::empty({std::max<int64_t>(1, spk_schnet_scratch_floats(&c.m, N))}, R.options());
  auto res = schnet_backward_raw(gxh, r, saved, scratch, *c.plan, ws, F, n_filters, rbf_kind, p0, p1, cutoff, true, want_gx0);
  Tensor gR = pairwise_bwd_raw(std::get<0>(res), idx_i, idx_j, N);
  return {gR, std::get<1>(res).defined() ? std::get<1>(res) : at::empty({0}, R.options())};
}

// Warm-up for callers that capture the force call into a HIP graph right after a list change (md.NVESimulation): evaluates the
// molecule / group relation of an EXISTING plan now, outside the capture (it costs one D2H).  Returns 1 if energies can be stored.
int64_t potential_plan_op(const Tensor& idx_i, const Tensor& idx_j, int64_t n_atoms, const Tensor& idx_m, int64_t n_mol) {
  auto plan = find_plan(idx_i, idx_j, n_atoms);
  if (!plan || plan->n_groups <= 0) return 0;
  c10::DeviceGuard guard(idx_i.device());
  return molecules_inside_groups(*plan, i64(idx_m, "potential_plan"), n_mol) ? 1 : 0;
}

// Energies and forces of the standard potential for eval: no autograd node, (E, F = -dE/dR, scalar_representation).  x0 or
// (embedding table, Z) -- with the table the lookup happens inside the forward launch.
std::tuple<Tensor, Tensor, Tensor> schnet_potential_forces_raw(const c10::optional<Tensor>& x0_in, const c10::optional<Tensor>& emb_in, const Tensor& Z_in,
                                                               const Tensor& R_in, const c10::optional<Tensor>& offsets_in, const Tensor& idx_i,
                                                               const Tensor& idx_j, const Tensor& idx_m_in, int64_t n_mol, at::TensorList ws,
                                                               at::TensorList head, int64_t n_filters, int64_t rbf_kind, const Tensor& p0,
                                                               const c10::optional<Tensor>& p1, double cutoff, int64_t head_act) {
  const bool has_x0 = x0_in.has_value() && x0_in->defined();
  TORCH_CHECK(has_x0 || (emb_in.has_value() && emb_in->defined()), "schnet_potential_forces: neither features nor an embedding table");
  Tensor R = f32(R_in.detach(), "schnet_potential_forces");
  Tensor off = opt_f32(offsets_in, "schnet_potential_forces");
  Tensor idx_m = i64(idx_m_in, "schnet_potential_forces"), Z = i64(Z_in, "schnet_potential_forces");
  Tensor x0 = has_x0 ? f32(x0_in->detach(), "schnet_potential_forces") : Tensor();
  Tensor emb = has_x0 ? Tensor() : f32(emb_in->detach(), "schnet_potential_forces");
  const int64_t N = R.size(0), F = has_x0 ? x0.size(1) : emb.size(1);
  c10::DeviceGuard guard(R.device());
  c10::optional<Tensor> off_d = off.defined() ? c10::optional<Tensor>(off) : c10::optional<Tensor>();
  PotentialCall c = potential_setup(R, off_d, idx_i, idx_j, N, F, ws, head, n_filters, rbf_kind, p0, p1, cutoff, head_act);
  if (c.fused) {
    const bool inside = molecules_inside_groups(*c.plan, idx_m, n_mol);
    Tensor x = at::empty({N, F}, R.options()), E = at::empty({n_mol}, R.options()), Fo = at::empty({N, 3}, R.options());
    Tensor pre_h = at::empty({N, (int64_t)c.head.n_hidden}, R.options());
    Tensor saved = at::empty({std::max<int64_t>(1, spk_schnet_saved_floats_graph(&c.m, &c.g, &c.rb))}, R.options());
    check(spk_schnet_potential_forces_f32(&c.m, &c.head, &c.g, &c.rb, fp(x0), fp(emb), Z.data_ptr<int64_t>(), emb.defined() ? (int32_t)emb.size(0) : 0, fp(R),
                                          fp(off), idx_m.data_ptr<int64_t>(), n_mol, inside ? 1 : 0, fpm(x), fpm(E), fpm(Fo), fpm(pre_h), fpm(saved),
                                          stream_of(R)));
    return {E, Fo, x};
  }
  if (!has_x0) x0 = emb.index_select(0, Z);
  auto fw = schnet_potential_forward_raw(x0, R, off_d, idx_i, idx_j, idx_m, n_mol, ws, head, n_filters, rbf_kind, p0, p1, cutoff, head_act);
  Tensor ones = at::ones({n_mol}, R.options());
  auto bw = schnet_potential_backward_raw(ones, c10::nullopt, x0, R, off_d, idx_i, idx_j, idx_m, n_mol, std::get<2>(fw), std::get<3>(fw), ws, head, n_filters,
                                          rbf_kind, p0, p1, cutoff, head_act, false);
  return {std::get<0>(fw), at::neg(std::get<0>(bw)), std::get<1>(fw)};
}

// The same for PaiNN: (E, F = -dE/dR, scalar_representation, vector_representation), eval only, no autograd node.  Batches of small
// molecules: TWO launches (spk_painn_potential_forces_f32); every other list runs the same stages through their own kernels.
std::tuple<Tensor, Tensor, Tensor, Tensor> painn_potential_forces_raw(const c10::optional<Tensor>& q0_in, const c10::optional<Tensor>& emb_in, const Tensor& Z_in,
                                                                      const Tensor& R_in, const c10::optional<Tensor>& offsets_in, const Tensor& idx_i,
                                                                      const Tensor& idx_j, const Tensor& idx_m_in, int64_t n_mol, at::TensorList ws,
                                                                      at::TensorList head, bool shared_filters, double eps, int64_t rbf_kind,
                                                                      const Tensor& p0_in, const c10::optional<Tensor>& p1_in, double cutoff, int64_t head_act) {
  const char* who = "painn_potential_forces";
  const bool has_q0 = q0_in.has_value() && q0_in->defined();
  TORCH_CHECK(has_q0 || (emb_in.has_value() && emb_in->defined()), who, ": neither features nor an embedding table");
  TORCH_CHECK(head.size() == 4, who, ": head = [outnet.0.weight, outnet.0.bias, outnet.1.weight, outnet.1.bias]");
  Tensor R = f32(R_in.detach(), who);
  Tensor off = opt_f32(offsets_in, who);
  Tensor idx_m = i64(idx_m_in, who), Z = i64(Z_in, who);
  Tensor q0 = has_q0 ? f32(q0_in->detach(), who) : Tensor();
  Tensor emb = has_q0 ? Tensor() : f32(emb_in->detach(), who);
  const int64_t N = R.size(0), F = has_q0 ? q0.size(1) : emb.size(1);
  c10::DeviceGuard guard(R.device());
  c10::optional<Tensor> off_d = off.defined() ? c10::optional<Tensor>(off) : c10::optional<Tensor>();
  auto plan = find_plan(idx_i, idx_j, N);
  if (!plan || plan->filter_pairs < 0) {       // first call on this list: the plan needs the geometry once
    Tensor r = pairwise_raw(R, idx_i, idx_j, off_d);
    plan = get_plan(idx_i, idx_j, N, r);
    decide_filter(*plan, r, cutoff);
  }
  auto M = get_painn(ws, F, shared_filters, eps);
  auto H = get_head(head[0], head[1], head[2], head[3]);
  Tensor p0 = f32(p0_in, who), p1 = opt_f32(p1_in, who);
  spk_graph_t g = plan->graph();
  spk_radial_t rb = radial_of(rbf_kind, p0, p1, cutoff);
  spk_head_t hd;
  hd.w1 = fp(H->w1); hd.w1t = fp(H->w1t); hd.b1 = fp(H->b1); hd.w2 = fp(H->w2); hd.b2 = fp(H->b2);
  hd.n_hidden = (int32_t)H->w1.size(0); hd.act = (int32_t)head_act;
  const bool fused = H->w1.dim() == 2 && H->w1.size(1) == F && H->w2.numel() == H->w1.size(0) && H->b1.defined() &&
                     spk_painn_potential_supported(&M->m, &hd, &g, &rb) != 0;
  if (fused) {
    const bool inside = molecules_inside_groups(*plan, idx_m, n_mol);
    Tensor q = at::empty({N, F}, R.options()), mu = at::empty({N, 3, F}, R.options());
    Tensor E = at::empty({n_mol}, R.options()), Fo = at::empty({N, 3}, R.options());
    Tensor pre_h = at::empty({N, (int64_t)hd.n_hidden}, R.options());
    Tensor saved = at::empty({std::max<int64_t>(1, spk_painn_saved_floats(&M->m, N))}, R.options());
    Tensor scratch = at::empty({std::max<int64_t>(1, spk_painn_scratch_floats(&M->m, N))}, R.options());
    check(spk_painn_potential_forces_f32(&M->m, &hd, &g, &rb, fp(q0), fp(emb), Z.data_ptr<int64_t>(), emb.defined() ? (int32_t)emb.size(0) : 0, fp(R), fp(off),
                                         idx_m.data_ptr<int64_t>(), n_mol, inside ? 1 : 0, fpm(q), fpm(mu), fpm(E), fpm(Fo), fpm(pre_h), fpm(saved), fpm(scratch),
                                         stream_of(R)));
    return {E, Fo, q, mu};
  }
  if (!has_q0) q0 = emb.index_select(0, Z);
  Tensor r = pairwise_raw(R, idx_i, idx_j, off_d);
  std::shared_ptr<Plan> pl;
  auto fw = painn_forward_raw(q0, r, idx_i, idx_j, ws, shared_filters, eps, rbf_kind, p0, p1_in, cutoff, &pl);
  auto hdf = atomwise_forward_raw(std::get<0>(fw), head[0], head[1], head[2], head[3], idx_m, n_mol, head_act);
  Tensor gq = atomwise_backward_raw(at::ones({n_mol}, R.options()), Tensor(), std::get<2>(hdf), head[0], head[2], idx_m, n_mol, head_act);
  auto bw = painn_backward_raw(gq, Tensor(), r, std::get<2>(fw), std::get<3>(fw), *pl, ws, F, shared_filters, eps, rbf_kind, p0, p1_in, cutoff, false);
  Tensor gR = pairwise_bwd_raw(std::get<0>(bw), idx_i, idx_j, N);
  return {std::get<0>(hdf), at::neg(gR), std::get<0>(fw), std::get<1>(fw)};
}

// ------------------------------------------------------------------------------------------------ dispatcher handles
template <class Sig>
c10::TypedOperatorHandle<Sig> op_handle(const char* name) {
  return c10::Dispatcher::singleton().findSchemaOrThrow(name, "").typed<Sig>();
}
Tensor call_gather(const Tensor& x, const Tensor& idx, int64_t dim) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&, int64_t)>("spk_hip::gather");
  return op.call(x, idx, dim);
}
Tensor call_scatter_add(const Tensor& x, const Tensor& idx, int64_t dim_size, int64_t dim) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&, int64_t, int64_t)>("spk_hip::scatter_add");
  return op.call(x, idx, dim_size, dim);
}
Tensor call_pairwise(const Tensor& R, const Tensor& ii, const Tensor& jj, const c10::optional<Tensor>& off) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&, const Tensor&, const c10::optional<Tensor>&)>("spk_hip::pairwise");
  return op.call(R, ii, jj, off);
}
Tensor call_pairwise_backward(const Tensor& gr, const Tensor& ii, const Tensor& jj, int64_t n) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&, const Tensor&, int64_t)>("spk_hip::pairwise_backward");
  return op.call(gr, ii, jj, n);
}

using OptT = c10::optional<Tensor>;
std::tuple<Tensor, Tensor> call_dense_forward(const Tensor& x, const Tensor& w, const OptT& b, int64_t act) {
  static auto op = op_handle<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const OptT&, int64_t)>("spk_hip::dense_forward");
  return op.call(x, w, b, act);
}
std::tuple<Tensor, Tensor> call_radial_cutoff(const Tensor& d, int64_t kind, const Tensor& p0, const OptT& p1, double cutoff, bool want_phi, bool want_cut) {
  static auto op = op_handle<std::tuple<Tensor, Tensor>(const Tensor&, int64_t, const Tensor&, const OptT&, double, bool, bool)>("spk_hip::radial_cutoff");
  return op.call(d, kind, p0, p1, cutoff, want_phi, want_cut);
}
std::tuple<Tensor, Tensor, Tensor> call_schnet_forward(const Tensor& x0, const Tensor& r, const Tensor& ii, const Tensor& jj, at::TensorList ws, int64_t nf,
                                                       int64_t kind, const Tensor& p0, const OptT& p1, double cutoff, bool save) {
  static auto op = op_handle<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, at::TensorList, int64_t, int64_t,
                                                                const Tensor&, const OptT&, double, bool)>("spk_hip::schnet_forward");
  return op.call(x0, r, ii, jj, ws, nf, kind, p0, p1, cutoff, save);
}
std::tuple<Tensor, Tensor, Tensor, Tensor> call_painn_forward(const Tensor& q0, const Tensor& r, const Tensor& ii, const Tensor& jj, at::TensorList ws,
                                                              bool shared, double eps, int64_t kind, const Tensor& p0, const OptT& p1, double cutoff) {
  static auto op = op_handle<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, at::TensorList, bool, double,
                                                                        int64_t, const Tensor&, const OptT&, double)>("spk_hip::painn_forward");
  return op.call(q0, r, ii, jj, ws, shared, eps, kind, p0, p1, cutoff);
}
std::tuple<Tensor, Tensor, Tensor> call_atomwise_forward(const Tensor& x, const Tensor& w1, const OptT& b1, const Tensor& w2, const OptT& b2, const Tensor& idx_m,
                                                         int64_t n_mol, int64_t act) {
  static auto op = op_handle<std::tuple<Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const OptT&, const Tensor&, const OptT&, const Tensor&, int64_t,
                                                                int64_t)>("spk_hip::atomwise_forward");
  return op.call(x, w1, b1, w2, b2, idx_m, n_mol, act);
}

std::tuple<Tensor, Tensor> call_schnet_backward(const Tensor& gx, const Tensor& r, const Tensor& saved, const Tensor& scratch, const Tensor& ii, const Tensor& jj,
                                                at::TensorList ws, int64_t nf, int64_t kind, const Tensor& p0, const OptT& p1, double cutoff, bool saved_filters,
                                                bool want_gx0) {
  static auto op = op_handle<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, at::TensorList,
                                                        int64_t, int64_t, const Tensor&, const OptT&, double, bool, bool)>("spk_hip::schnet_backward");
  return op.call(gx, r, saved, scratch, ii, jj, ws, nf, kind, p0, p1, cutoff, saved_filters, want_gx0);
}
std::tuple<Tensor, Tensor> call_painn_backward(const OptT& gq, const OptT& gmu, const Tensor& r, const Tensor& saved, const Tensor& scratch, const Tensor& ii,
                                               const Tensor& jj, int64_t n_atoms, at::TensorList ws, bool shared, double eps, int64_t kind, const Tensor& p0,
                                               const OptT& p1, double cutoff, bool want_gq0) {
  static auto op = op_handle<std::tuple<Tensor, Tensor>(const OptT&, const OptT&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t,
                                                        at::TensorList, bool, double, int64_t, const Tensor&, const OptT&, double, bool)>("spk_hip::painn_backward");
  return op.call(gq, gmu, r, saved, scratch, ii, jj, n_atoms, ws, shared, eps, kind, p0, p1, cutoff, want_gq0);
}
Tensor call_atomwise_backward(const OptT& gE, const OptT& gy, const Tensor& pre, const Tensor& w1, const Tensor& w2, const Tensor& idx_m, int64_t n_mol, int64_t act) {
  static auto op = op_handle<Tensor(const OptT&, const OptT&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t, int64_t)>("spk_hip::atomwise_backward");
  return op.call(gE, gy, pre, w1, w2, idx_m, n_mol, act);
}
Tensor call_dense_backward_input(const Tensor& gy, const Tensor& pre, const Tensor& w, int64_t act) {
  static auto op = op_handle<Tensor(const Tensor&, const Tensor&, const Tensor&, int64_t)>("spk_hip::dense_backward_input");
  return op.call(gy, pre, w, act);
}
Tensor call_radial_cutoff_backward(const Tensor& d, int64_t kind, const Tensor& p0, const OptT& p1, double cutoff, const OptT& gphi, const OptT& gfc) {
  static auto op = op_handle<Tensor(const Tensor&, int64_t, const Tensor&, const OptT&, double, const OptT&, const OptT&)>("spk_hip::radial_cutoff_backward");
  return op.call(d, kind, p0, p1, cutoff, gphi, gfc);
}
OptT opt_of(const Tensor& t) { return t.defined() ? OptT(t) : OptT(); }
std::tuple<Tensor, Tensor, Tensor, Tensor> call_potential_forward(const Tensor& x0, const Tensor& R, const OptT& off, const Tensor& ii, const Tensor& jj,
                                                                  const Tensor& idx_m, int64_t n_mol, at::TensorList ws, at::TensorList head, int64_t nf,
                                                                  int64_t kind, const Tensor& p0, const OptT& p1, double cutoff, int64_t head_act) {
  static auto op = op_handle<std::tuple<Tensor, Tensor, Tensor, Tensor>(const Tensor&, const Tensor&, const OptT&, const Tensor&, const Tensor&, const Tensor&,
                                                                        int64_t, at::TensorList, at::TensorList, int64_t, int64_t, const Tensor&, const OptT&,
                                                                        double, int64_t)>("spk_hip::schnet_potential_forward");
  return op.call(x0, R, off, ii, jj, idx_m, n_mol, ws, head, nf, kind, p0, p1, cutoff, head_act);
}
std::tuple<Tensor, Tensor> call_potential_backward(const OptT& gE, const OptT& gx, const Tensor& x0, const Tensor& R, const OptT& off, const Tensor& ii,
                                                   const Tensor& jj, const Tensor& idx_m, int64_t n_mol, const Tensor& saved, const Tensor& pre_h,
                                                   at::TensorList ws, at::TensorList head, int64_t nf, int64_t kind, const Tensor& p0, const OptT& p1,
                                                   double cutoff, int64_t head_act, bool want_gx0) {
  static auto op = op_handle<std::tuple<Tensor, Tensor>(const OptT&, const OptT&, const Tensor&, const Tensor&, const OptT&, const Tensor&, const Tensor&,
                                                        const Tensor&, int64_t, const Tensor&, const Tensor&, at::TensorList, at::TensorList, int64_t, int64_t,
                                                        const Tensor&, const OptT&, double, int64_t, bool)>("spk_hip::schnet_potential_backward");
  return op.call(gE, gx, x0, R, off, ii, jj, idx_m, n_mol, saved, pre_h, ws, head, nf, kind, p0, p1, cutoff, head_act, want_gx0);
}

const char* kEvalOnly =
    ": the fused eval-mode path computes first-order gradients w.r.t. the geometry and the input features only -- a gradient w.r.t. "
    "its weights (or a recorded backward, create_graph=True) was requested.  Put the module in training mode (module.train()): the "
    "training path is differentiable to any order in all parameters.";

#include "spk_torch_train.h"
#include "spk_torch_fm.h"

// ------------------------------------------------------------------------------------------------ autograd: primitives
// scatter_add <-> gather are each other's transposes; both backward passes call the differentiable operator again, so the
// pair is closed under differentiation (force training needs the second order, atomistic/response.py:67).
struct ScatterAddFn : public torch::autograd::Function<ScatterAddFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& idx, int64_t dim_size, int64_t dim) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->saved_data["idx"] = idx;
    ctx->saved_data["dim"] = dim;
    return call_scatter_add(x, idx, dim_size, dim);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    return {call_gather(g[0], ctx->saved_data["idx"].toTensor(), ctx->saved_data["dim"].toInt()), Tensor(), Tensor(), Tensor()};
  }
};
struct GatherFn : public torch::autograd::Function<GatherFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& idx, int64_t dim) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->saved_data["idx"] = idx;
    ctx->saved_data["dim"] = dim;
    ctx->saved_data["rows"] = x.size(at::maybe_wrap_dim(dim, x.dim()));
    return call_gather(x, idx, dim);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    return {call_scatter_add(g[0], ctx->saved_data["idx"].toTensor(), ctx->saved_data["rows"].toInt(), ctx->saved_data["dim"].toInt()), Tensor(), Tensor()};
  }
};

// r_ij = R[j] - R[i] + offsets is linear: backward = its transpose (pairwise_backward), whose backward is pairwise again
struct PairwiseFn : public torch::autograd::Function<PairwiseFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& R, const Tensor& ii, const Tensor& jj, const c10::optional<Tensor>& off) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->saved_data["ii"] = ii;
    ctx->saved_data["jj"] = jj;
    ctx->saved_data["n"] = R.size(0);
    ctx->saved_data["has_off"] = off.has_value() && off->defined();
    return call_pairwise(R, ii, jj, off);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    Tensor gR, goff;
    if (ctx->needs_input_grad(0)) gR = call_pairwise_backward(g[0], ctx->saved_data["ii"].toTensor(), ctx->saved_data["jj"].toTensor(), ctx->saved_data["n"].toInt());
    if (ctx->saved_data["has_off"].toBool() && ctx->needs_input_grad(3)) goff = g[0];    // d r_ij / d offsets = 1 (stress via Strain, atomistic/response.py:434-464)
    return {gR, Tensor(), Tensor(), goff};
  }
};
struct PairwiseBwdFn : public torch::autograd::Function<PairwiseBwdFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& gr, const Tensor& ii, const Tensor& jj, int64_t n) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->saved_data["ii"] = ii;
    ctx->saved_data["jj"] = jj;
    return call_pairwise_backward(gr, ii, jj, n);
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    return {call_pairwise(g[0], ctx->saved_data["ii"].toTensor(), ctx->saved_data["jj"].toTensor(), c10::nullopt), Tensor(), Tensor(), Tensor()};
  }
};

// Dense (nn/base.py:52-55).  The node has two outputs, (y, pre-activation): the pre-activation is saved as an OUTPUT of the
// node, so a recorded backward that multiplies by act'(pre) stays connected to x, w, b through this same node (its second
// incoming gradient) and the act'' terms of the second order come out exact.  Backward, decided per backward pass:
//  * plain first-order pass asking for the input gradient only (eval-mode Forces: needs_input_grad is per graph task, and
//    autograd.grad w.r.t. the positions does not ask for the weights): one fused kernel, (gy . act'(pre)) W;
//  * otherwise: u = gy . act'(pre) + gpre, gx = u W, (gw, gb) = (u^T x, column sums of u) -- three launches of the
//    differentiable operators of spk_torch_train.h, so a recorded pass (create_graph=True) can be differentiated again.
struct DenseFn : public torch::autograd::Function<DenseFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, int64_t act) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto yp = call_dense_forward(x, w, b, act);
    ctx->saved_data["act"] = act;
    ctx->saved_data["has_bias"] = b.has_value() && b->defined();
    ctx->save_for_backward({x, w, std::get<1>(yp)});
    return {std::get<0>(yp), std::get<1>(yp)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &w = saved[1], &pre = saved[2];
    const int64_t act = ctx->saved_data["act"].toInt();
    const bool has_bias = ctx->saved_data["has_bias"].toBool();
    const Tensor &gy = grads[0], &gpre = grads[1];
    const bool recorded = at::GradMode::is_enabled();
    const bool need_w = ctx->needs_input_grad(1), need_b = has_bias && ctx->needs_input_grad(2);
    const bool has_gpre = act != SPK_ACT_NONE && gpre.defined();
    Tensor gx, gw, gb;
    if (!gy.defined() && !has_gpre) return {gx, gw, gb, Tensor()};
    if (!recorded && !need_w && !need_b && !has_gpre) {
      if (ctx->needs_input_grad(0)) gx = call_dense_backward_input(gy, pre, w, act);
      return {gx, gw, gb, Tensor()};
    }
    Tensor u = gy;
    if (act != SPK_ACT_NONE) u = gy.defined() ? call_act_mul(gy, pre, act, 1, opt_of(has_gpre ? gpre : Tensor())) : gpre;
    if (!recorded && ctx->needs_input_grad(0) && (need_w || need_b)) {
      auto r = call_gemm_pair(u, w, true, u, x);                    // u W and u^T x (+ column sums) in one launch
      gx = std::get<0>(r);
      if (need_w) gw = std::get<1>(r);
      if (need_b) gb = std::get<2>(r);
      return {gx, gw, gb, Tensor()};
    }
    if (ctx->needs_input_grad(0)) gx = call_matmul_nn(u, w);
    if (need_w || need_b) {
      auto r = call_matmul_tn(u, x);
      if (need_w) gw = std::get<0>(r);
      if (need_b) gb = std::get<1>(r);
    }
    return {gx, gw, gb, Tensor()};
  }
};

// (phi, fcut)(d): first-order backward on the HIP kernel (eval path; the training path uses the torch formulas)
struct RadialCutoffFn : public torch::autograd::Function<RadialCutoffFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& d, int64_t kind, const Tensor& p0, const c10::optional<Tensor>& p1,
                               double cutoff, bool want_phi, bool want_cut) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto out = call_radial_cutoff(d, kind, p0, p1, cutoff, want_phi, want_cut);
    ctx->save_for_backward({d, p0, (p1.has_value() && p1->defined()) ? *p1 : Tensor()});
    ctx->saved_data["kind"] = kind;
    ctx->saved_data["cutoff"] = cutoff;
    ctx->saved_data["want"] = std::vector<int64_t>{want_phi, want_cut};
    return {std::get<0>(out), std::get<1>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    TORCH_CHECK(!at::GradMode::is_enabled(), "spk_hip::radial_cutoff", kEvalOnly);
    auto saved = ctx->get_saved_variables();
    auto want = ctx->saved_data["want"].toIntVector();
    c10::optional<Tensor> gphi, gfc, p1;
    if (want[0] && g[0].defined()) gphi = g[0];
    if (want[1] && g[1].defined()) gfc = g[1];
    if (saved[2].defined()) p1 = saved[2];
    Tensor gd = call_radial_cutoff_backward(saved[0], ctx->saved_data["kind"].toInt(), saved[1], p1, ctx->saved_data["cutoff"].toDouble(), gphi, gfc);
    return {gd, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// ------------------------------------------------------------------------------------------------ autograd: fused eval path
// needs_input_grad() is indexed by the TENSOR inputs of the node in argument order (its outgoing edges), not by argument
// position; `first` .. n_vars - 1 are the parameters (and integer index tensors, which never need a gradient)
void check_eval_backward(AutogradContext* ctx, const char* who, size_t first) {
  TORCH_CHECK(!at::GradMode::is_enabled(), who, kEvalOnly);
  const size_t n_vars = (size_t)ctx->saved_data["n_vars"].toInt();
  for (size_t k = first; k < n_vars; ++k) TORCH_CHECK(!ctx->needs_input_grad(k), who, kEvalOnly);
}

struct SchNetFn : public torch::autograd::Function<SchNetFn> {
  // tensor inputs in positions 0 (x0), 1 (r_ij), 6 / 7 (radial parameters) and 9.. (weights)
  // save_filters: keep the raw filter outputs for the backward (decided by the caller: grad mode is off inside forward())
  static Tensor forward(AutogradContext* ctx, const Tensor& x0, const Tensor& r_ij, const Tensor& idx_i, const Tensor& idx_j,
                        int64_t n_filters, int64_t rbf_kind, const Tensor& p0, const c10::optional<Tensor>& p1, double cutoff,
                        bool save_filters, at::TensorList ws) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto out = call_schnet_forward(x0, r_ij, idx_i, idx_j, ws, n_filters, rbf_kind, p0, p1, cutoff, save_filters);
    std::vector<Tensor> sv{r_ij, std::get<1>(out), std::get<2>(out), p0, (p1.has_value() && p1->defined()) ? *p1 : Tensor(), idx_i, idx_j};
    for (const auto& w : ws) sv.push_back(w);
    ctx->save_for_backward(sv);
    ctx->saved_data["cfg"] = std::vector<int64_t>{x0.size(0), x0.size(1), n_filters, rbf_kind, save_filters};
    ctx->saved_data["n_vars"] = (int64_t)(5 + ((p1.has_value() && p1->defined()) ? 1 : 0) + ws.size());
    ctx->saved_data["cutoff"] = cutoff;
    return std::get<0>(out);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto sv = ctx->get_saved_variables();
    const size_t n_ws = sv.size() - 7;
    check_eval_backward(ctx, "spk_hip::schnet", 4);
    auto cfg = ctx->saved_data["cfg"].toIntVector();
    std::vector<Tensor> ws(sv.begin() + 7, sv.end());
    Tensor gx = grads[0].defined() ? grads[0] : at::zeros({cfg[0], cfg[1]}, sv[0].options());
    auto res = call_schnet_backward(gx, sv[0], sv[1], sv[2], sv[5], sv[6], ws, cfg[2], cfg[3], sv[3], opt_of(sv[4]), ctx->saved_data["cutoff"].toDouble(),
                                    cfg[4] != 0, ctx->needs_input_grad(0));
    variable_list out(10 + n_ws);
    if (ctx->needs_input_grad(0)) out[0] = std::get<1>(res);
    if (ctx->needs_input_grad(1)) out[1] = std::get<0>(res);
    return out;
  }
};

// (E, scalar_representation) of the standard potential; first-order gradients w.r.t. the positions and the embedding rows
struct SchnetPotentialFn : public torch::autograd::Function<SchnetPotentialFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& x0, const Tensor& R, const Tensor& idx_i, const Tensor& idx_j, const Tensor& idx_m,
                               const Tensor& p0, const c10::optional<Tensor>& offsets, const c10::optional<Tensor>& p1, at::TensorList ws,
                               at::TensorList head, int64_t n_mol, int64_t n_filters, int64_t rbf_kind, double cutoff, int64_t head_act) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto out = call_potential_forward(x0, R, offsets, idx_i, idx_j, idx_m, n_mol, ws, head, n_filters, rbf_kind, p0, p1, cutoff, head_act);
    const bool has_off = offsets.has_value() && offsets->defined(), has_p1 = p1.has_value() && p1->defined();
    std::vector<Tensor> sv{x0, R, idx_i, idx_j, idx_m, p0, has_off ? *offsets : Tensor(), has_p1 ? *p1 : Tensor(), std::get<2>(out), std::get<3>(out)};
    for (const auto& w : ws) sv.push_back(w);
    for (const auto& w : head) sv.push_back(w);
    ctx->save_for_backward(sv);
    ctx->saved_data["cfg"] = std::vector<int64_t>{n_mol, n_filters, rbf_kind, head_act, (int64_t)ws.size(), (int64_t)head.size(), has_off, has_p1};
    ctx->saved_data["n_vars"] = (int64_t)(6 + (has_off ? 1 : 0) + (has_p1 ? 1 : 0) + ws.size() + head.size());
    ctx->saved_data["cutoff"] = cutoff;
    return {std::get<0>(out), std::get<1>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    check_eval_backward(ctx, "spk_hip::schnet_potential", 2);       // (before the saved tensors are touched: a pass that asks for
    auto sv = ctx->get_saved_variables();                           //  weight gradients gets THIS message)
    auto cfg = ctx->saved_data["cfg"].toIntVector();
    const size_t n_ws = (size_t)cfg[4], n_head = (size_t)cfg[5];
    std::vector<Tensor> ws(sv.begin() + 10, sv.begin() + 10 + n_ws), head(sv.begin() + 10 + n_ws, sv.begin() + 10 + n_ws + n_head);
    auto res = call_potential_backward(opt_of(grads[0]), opt_of(grads[1]), sv[0], sv[1], opt_of(sv[6]), sv[2], sv[3], sv[4], cfg[0], sv[8], sv[9], ws, head,
                                       cfg[1], cfg[2], sv[5], opt_of(sv[7]), ctx->saved_data["cutoff"].toDouble(), cfg[3], ctx->needs_input_grad(0));
    variable_list out(8 + n_ws + n_head + 5);
    if (ctx->needs_input_grad(0)) out[0] = std::get<1>(res);
    if (ctx->needs_input_grad(1)) out[1] = std::get<0>(res);
    return out;
  }
};

// Outputs of the no-autograd eval operators (schnet_potential_forces) pass through this node: no kernel, the value is an alias --
// but a backward pass that reaches it (parameter gradients of an eval-mode model) raises the eval-only message instead of
// "does not require grad".
struct EvalGuardFn : public torch::autograd::Function<EvalGuardFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& y, at::TensorList params) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->saved_data["n"] = (int64_t)params.size();
    return y.alias();
  }
  static variable_list backward(AutogradContext* ctx, variable_list) {
    TORCH_CHECK(false, "spk_hip::schnet_potential_forces / painn_potential_forces", kEvalOnly);
    return variable_list((size_t)ctx->saved_data["n"].toInt() + 1);
  }
};
Tensor eval_guard_ad(const Tensor& y, at::TensorList params) { return EvalGuardFn::apply(y, params); }
Tensor eval_guard_dev(const Tensor& y, at::TensorList) { return y.alias(); }

struct PaiNNFn : public torch::autograd::Function<PaiNNFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& q0, const Tensor& r_ij, const Tensor& idx_i, const Tensor& idx_j,
                               bool shared_filters, double eps, int64_t rbf_kind, const Tensor& p0, const c10::optional<Tensor>& p1,
                               double cutoff, at::TensorList ws) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto out = call_painn_forward(q0, r_ij, idx_i, idx_j, ws, shared_filters, eps, rbf_kind, p0, p1, cutoff);
    std::vector<Tensor> sv{r_ij, std::get<2>(out), std::get<3>(out), p0, (p1.has_value() && p1->defined()) ? *p1 : Tensor(), idx_i, idx_j};
    for (const auto& w : ws) sv.push_back(w);
    ctx->save_for_backward(sv);
    ctx->saved_data["cfg"] = std::vector<int64_t>{q0.size(0), shared_filters, rbf_kind};
    ctx->saved_data["n_vars"] = (int64_t)(5 + ((p1.has_value() && p1->defined()) ? 1 : 0) + ws.size());
    ctx->saved_data["cutoff"] = cutoff;
    ctx->saved_data["eps"] = eps;
    return {std::get<0>(out), std::get<1>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto sv = ctx->get_saved_variables();
    const size_t n_ws = sv.size() - 7;
    check_eval_backward(ctx, "spk_hip::painn", 4);
    auto cfg = ctx->saved_data["cfg"].toIntVector();
    std::vector<Tensor> ws(sv.begin() + 7, sv.end());
    auto res = call_painn_backward(opt_of(grads[0]), opt_of(grads[1]), sv[0], sv[1], sv[2], sv[5], sv[6], cfg[0], ws, cfg[1] != 0, ctx->saved_data["eps"].toDouble(),
                                   cfg[2], sv[3], opt_of(sv[4]), ctx->saved_data["cutoff"].toDouble(), ctx->needs_input_grad(0));
    variable_list out(10 + n_ws);
    if (ctx->needs_input_grad(0)) out[0] = std::get<1>(res);
    if (ctx->needs_input_grad(1)) out[1] = std::get<0>(res);
    return out;
  }
};

struct AtomwiseFn : public torch::autograd::Function<AtomwiseFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& x, const Tensor& w1, const c10::optional<Tensor>& b1, const Tensor& w2,
                               const c10::optional<Tensor>& b2, const Tensor& idx_m, int64_t n_mol, int64_t act) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto out = call_atomwise_forward(x, w1, b1, w2, b2, idx_m, n_mol, act);
    ctx->save_for_backward({std::get<2>(out), w1, w2, idx_m});
    ctx->saved_data["cfg"] = std::vector<int64_t>{n_mol, act};
    ctx->saved_data["n_vars"] = (int64_t)(4 + ((b1.has_value() && b1->defined()) ? 1 : 0) + ((b2.has_value() && b2->defined()) ? 1 : 0));
    return {std::get<0>(out), std::get<1>(out)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    check_eval_backward(ctx, "spk_hip::atomwise", 1);
    auto sv = ctx->get_saved_variables();
    auto cfg = ctx->saved_data["cfg"].toIntVector();
    Tensor gx;
    if (ctx->needs_input_grad(0)) gx = call_atomwise_backward(opt_of(grads[0]), opt_of(grads[1]), sv[0], sv[1], sv[2], sv[3], cfg[0], cfg[1]);
    return {gx, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// ------------------------------------------------------------------------------------------------ operator entry points
// --- Autograd key
Tensor scatter_add_ad(const Tensor& x, const Tensor& idx, int64_t dim_size, int64_t dim) { return ScatterAddFn::apply(x, idx, dim_size, dim); }
Tensor gather_ad(const Tensor& x, const Tensor& idx, int64_t dim) { return GatherFn::apply(x, idx, dim); }
Tensor pairwise_ad(const Tensor& R, const Tensor& ii, const Tensor& jj, const c10::optional<Tensor>& off) { return PairwiseFn::apply(R, ii, jj, off); }
Tensor pairwise_backward_ad(const Tensor& gr, const Tensor& ii, const Tensor& jj, int64_t n) { return PairwiseBwdFn::apply(gr, ii, jj, n); }
Tensor dense_ad(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, int64_t act) { return DenseFn::apply(x, w, b, act)[0]; }
std::tuple<Tensor, Tensor> radial_cutoff_ad(const Tensor& d, int64_t kind, const Tensor& p0, const c10::optional<Tensor>& p1, double cutoff,
                                            bool want_phi, bool want_cut) {
  auto r = RadialCutoffFn::apply(d, kind, p0, p1, cutoff, want_phi, want_cut);
  return {r[0], r[1]};
}
Tensor schnet_ad(const Tensor& x0, const Tensor& r_ij, const Tensor& idx_i, const Tensor& idx_j, at::TensorList ws, int64_t n_filters,
                 int64_t rbf_kind, const Tensor& p0, const c10::optional<Tensor>& p1, double cutoff) {
  // a gradient w.r.t. the geometry may be asked for: keep the raw filter outputs so that the backward runs the derivative GEMM only
  const bool save_filters = at::GradMode::is_enabled() && r_ij.requires_grad();
  return SchNetFn::apply(x0, r_ij, idx_i, idx_j, n_filters, rbf_kind, p0, p1, cutoff, save_filters, ws);
}
std::tuple<Tensor, Tensor> schnet_potential_ad(const Tensor& x0, const Tensor& R, const c10::optional<Tensor>& offsets, const Tensor& idx_i, const Tensor& idx_j,
                                               const Tensor& idx_m, int64_t n_mol, at::TensorList ws, at::TensorList head, int64_t n_filters, int64_t rbf_kind,
                                               const Tensor& p0, const c10::optional<Tensor>& p1, double cutoff, int64_t head_act) {
  TORCH_CHECK(!(offsets.has_value() && offsets->defined() && offsets->requires_grad()),
              "spk_hip::schnet_potential does not return a gradient w.r.t. the offsets (stress): run the modules separately for that");
  auto r = SchnetPotentialFn::apply(x0, R, idx_i, idx_j, idx_m, p0, offsets, p1, ws, head, n_mol, n_filters, rbf_kind, cutoff, head_act);
  return {r[0], r[1]};
}
std::tuple<Tensor, Tensor> schnet_potential_dev(const Tensor& x0, const Tensor& R, const c10::optional<Tensor>& offsets, const Tensor& idx_i, const Tensor& idx_j,
                                                const Tensor& idx_m, int64_t n_mol, at::TensorList ws, at::TensorList head, int64_t n_filters, int64_t rbf_kind,
                                                const Tensor& p0, const c10::optional<Tensor>& p1, double cutoff, int64_t head_act) {
  auto r = schnet_potential_forward_raw(x0, R, offsets, idx_i, idx_j, idx_m, n_mol, ws, head, n_filters, rbf_kind, p0, p1, cutoff, head_act);
  return {std::get<0>(r), std::get<1>(r)};
}
std::tuple<Tensor, Tensor> schnet_potential_meta(const Tensor& x0, const Tensor&, const c10::optional<Tensor>&, const Tensor&, const Tensor&, const Tensor&,
                                                 int64_t n_mol, at::TensorList, at::TensorList, int64_t, int64_t, const Tensor&, const c10::optional<Tensor>&,
                                                 double, int64_t) {
  return {at::empty({n_mol}, x0.options()), at::empty_like(x0)};
}
std::tuple<Tensor, Tensor, Tensor> schnet_potential_forces_meta(const c10::optional<Tensor>& x0, const c10::optional<Tensor>& emb, const Tensor&, const Tensor& R,
                                                                const c10::optional<Tensor>&, const Tensor&, const Tensor&, const Tensor&, int64_t n_mol,
                                                                at::TensorList, at::TensorList, int64_t, int64_t, const Tensor&, const c10::optional<Tensor>&,
                                                                double, int64_t) {
  const int64_t F = (x0.has_value() && x0->defined()) ? x0->size(1) : emb->size(1);
  return {at::empty({n_mol}, R.options()), at::empty_like(R), at::empty({R.size(0), F}, R.options())};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> painn_potential_forces_meta(const c10::optional<Tensor>& q0, const c10::optional<Tensor>& emb, const Tensor&, const Tensor& R,
                                                                       const c10::optional<Tensor>&, const Tensor&, const Tensor&, const Tensor&, int64_t n_mol,
                                                                       at::TensorList, at::TensorList, bool, double, int64_t, const Tensor&,
                                                                       const c10::optional<Tensor>&, double, int64_t) {
  const int64_t F = (q0.has_value() && q0->defined()) ? q0->size(1) : emb->size(1);
  return {at::empty({n_mol}, R.options()), at::empty_like(R), at::empty({R.size(0), F}, R.options()), at::empty({R.size(0), 3, F}, R.options())};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> schnet_potential_forward_meta(const Tensor& x0, const Tensor&, const c10::optional<Tensor>&, const Tensor& idx_i,
                                                                         const Tensor&, const Tensor&, int64_t n_mol, at::TensorList ws, at::TensorList head,
                                                                         int64_t nf, int64_t, const Tensor&, const c10::optional<Tensor>&, double, int64_t) {
  const int64_t N = x0.size(0), L = (int64_t)ws.size() / kSchnetPerLayer;
  return {at::empty({n_mol}, x0.options()), at::empty_like(x0), at::empty({L * (N * (nf + x0.size(1)) + (idx_i.size(0) / 2) * nf) + 1}, x0.options()),
          at::empty({N, head.size() ? head[0].size(0) : 0}, x0.options())};
}
std::tuple<Tensor, Tensor> schnet_potential_backward_meta(const c10::optional<Tensor>&, const c10::optional<Tensor>&, const Tensor& x0, const Tensor& R,
                                                          const c10::optional<Tensor>&, const Tensor&, const Tensor&, const Tensor&, int64_t, const Tensor&,
                                                          const Tensor&, at::TensorList, at::TensorList, int64_t, int64_t, const Tensor&,
                                                          const c10::optional<Tensor>&, double, int64_t, bool want_gx0) {
  return {at::empty_like(R), want_gx0 ? at::empty_like(x0) : at::empty({0}, x0.options())};
}
std::tuple<Tensor, Tensor> painn_ad(const Tensor& q0, const Tensor& r_ij, const Tensor& idx_i, const Tensor& idx_j, at::TensorList ws,
                                    bool shared_filters, double eps, int64_t rbf_kind, const Tensor& p0, const c10::optional<Tensor>& p1,
                                    double cutoff) {
  auto r = PaiNNFn::apply(q0, r_ij, idx_i, idx_j, shared_filters, eps, rbf_kind, p0, p1, cutoff, ws);
  return {r[0], r[1]};
}
std::tuple<Tensor, Tensor> atomwise_ad(const Tensor& x, const Tensor& w1, const c10::optional<Tensor>& b1, const Tensor& w2,
                                       const c10::optional<Tensor>& b2, const Tensor& idx_m, int64_t n_mol, int64_t act) {
  auto r = AtomwiseFn::apply(x, w1, b1, w2, b2, idx_m, n_mol, act);
  return {r[0], r[1]};
}

// --- device key (reached when no autograd is involved, e.g. inference mode)
Tensor dense_dev(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, int64_t act) { return std::get<0>(dense_raw(x, w, b, act)); }
Tensor schnet_dev(const Tensor& x0, const Tensor& r_ij, const Tensor& idx_i, const Tensor& idx_j, at::TensorList ws, int64_t n_filters,
                  int64_t rbf_kind, const Tensor& p0, const c10::optional<Tensor>& p1, double cutoff) {
  return std::get<0>(schnet_forward_raw(x0, r_ij, idx_i, idx_j, ws, n_filters, rbf_kind, p0, p1, cutoff, false));
}
std::tuple<Tensor, Tensor> painn_dev(const Tensor& q0, const Tensor& r_ij, const Tensor& idx_i, const Tensor& idx_j, at::TensorList ws,
                                     bool shared_filters, double eps, int64_t rbf_kind, const Tensor& p0, const c10::optional<Tensor>& p1,
                                     double cutoff) {
  auto r = painn_forward_raw(q0, r_ij, idx_i, idx_j, ws, shared_filters, eps, rbf_kind, p0, p1, cutoff, nullptr);
  return {std::get<0>(r), std::get<1>(r)};
}
std::tuple<Tensor, Tensor> atomwise_dev(const Tensor& x, const Tensor& w1, const c10::optional<Tensor>& b1, const Tensor& w2,
                                        const c10::optional<Tensor>& b2, const Tensor& idx_m, int64_t n_mol, int64_t act) {
  auto r = atomwise_forward_raw(x, w1, b1, w2, b2, idx_m, n_mol, act);
  return {std::get<0>(r), std::get<1>(r)};
}

// raw forward / backward launchers as operators of their own (stateless: saved / scratch are explicit tensors)
std::tuple<Tensor, Tensor, Tensor> schnet_forward_op(const Tensor& x0, const Tensor& r_ij, const Tensor& idx_i, const Tensor& idx_j, at::TensorList ws,
                                                     int64_t n_filters, int64_t rbf_kind, const Tensor& p0, const c10::optional<Tensor>& p1,
                                                     double cutoff, bool save_filters) {
  return schnet_forward_raw(x0, r_ij, idx_i, idx_j, ws, n_filters, rbf_kind, p0, p1, cutoff, save_filters);
}
// (the plan of the list is a cache hit: the forward of the same call built it; were it evicted, it is re-derived from r_ij)
std::tuple<Tensor, Tensor> schnet_backward_op(const Tensor& gx, const Tensor& r_ij, const Tensor& saved, const Tensor& scratch, const Tensor& idx_i,
                                              const Tensor& idx_j, at::TensorList ws, int64_t n_filters, int64_t rbf_kind, const Tensor& p0,
                                              const c10::optional<Tensor>& p1, double cutoff, bool saved_filters, bool want_gx0) {
  require_device(gx, "schnet_backward");
  auto plan = get_plan(idx_i, idx_j, gx.size(0), r_ij);
  decide_filter(*plan, r_ij, cutoff);
  auto res = schnet_backward_raw(gx, r_ij, saved, scratch, *plan, ws, gx.size(1), n_filters, rbf_kind, p0, p1, cutoff, saved_filters, want_gx0);
  return {std::get<0>(res), want_gx0 ? std::get<1>(res) : at::empty({0}, gx.options())};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> painn_forward_op(const Tensor& q0, const Tensor& r_ij, const Tensor& idx_i, const Tensor& idx_j,
                                                            at::TensorList ws, bool shared_filters, double eps, int64_t rbf_kind, const Tensor& p0,
                                                            const c10::optional<Tensor>& p1, double cutoff) {
  return painn_forward_raw(q0, r_ij, idx_i, idx_j, ws, shared_filters, eps, rbf_kind, p0, p1, cutoff, nullptr);
}
std::tuple<Tensor, Tensor> painn_backward_op(const c10::optional<Tensor>& gq, const c10::optional<Tensor>& gmu, const Tensor& r_ij, const Tensor& saved,
                                             const Tensor& scratch, const Tensor& idx_i, const Tensor& idx_j, int64_t n_atoms, at::TensorList ws,
                                             bool shared_filters, double eps, int64_t rbf_kind, const Tensor& p0, const c10::optional<Tensor>& p1,
                                             double cutoff, bool want_gq0) {
  require_device(r_ij, "painn_backward");
  auto plan = get_plan(idx_i, idx_j, n_atoms, r_ij);
  if (plan->filter_pairs < 0 && plan->n_edges >= (1 << 19)) decide_filter(*plan, r_ij, cutoff);
  const int64_t F = ws[0].size(0);     // interatomic_context_net.0.weight [F, F]
  auto res = painn_backward_raw((gq.has_value() && gq->defined()) ? *gq : Tensor(), (gmu.has_value() && gmu->defined()) ? *gmu : Tensor(), r_ij, saved, scratch,
                                *plan, ws, F, shared_filters, eps, rbf_kind, p0, p1, cutoff, want_gq0);
  return {std::get<0>(res), want_gq0 ? std::get<1>(res) : at::empty({0}, r_ij.options())};
}
Tensor atomwise_backward_op(const c10::optional<Tensor>& gE, const c10::optional<Tensor>& gy, const Tensor& pre, const Tensor& w1, const Tensor& w2,
                            const Tensor& idx_m, int64_t n_mol, int64_t act) {
  return atomwise_backward_raw((gE.has_value() && gE->defined()) ? *gE : Tensor(), (gy.has_value() && gy->defined()) ? *gy : Tensor(), pre, w1, w2, idx_m, n_mol, act);
}

// (rowptr, rev, half, flags[sorted, symmetric, n_half, filter_pairs, by-neighbour copy built]) of a list; also warms the cache outside a graph capture
// force_filter: -1 = decide from the geometry (cutoff > 0), 0 / 1 = switch the per-call pair compaction off / on for this list
std::tuple<Tensor, Tensor, Tensor, Tensor> edge_plan_op(const Tensor& idx_i, const Tensor& idx_j, int64_t n_atoms, const c10::optional<Tensor>& r_ij,
                                                        double cutoff, int64_t force_filter) {
  require_device(idx_i, "edge_plan");
  Tensor r = (r_ij.has_value() && r_ij->defined()) ? *r_ij : Tensor();
  auto p = get_plan(idx_i, idx_j, n_atoms, r);
  if (force_filter >= 0) p->filter_pairs = (force_filter != 0 && p->symmetric && p->n_edges > 0) ? 1 : 0;
  else if (r.defined() && cutoff > 0) decide_filter(*p, r, cutoff);
  Tensor flags = at::tensor(std::vector<int64_t>{p->sorted, p->symmetric, p->n_half, p->filter_pairs, p->has_transposed ? 1 : 0}, at::TensorOptions().dtype(at::kLong));
  auto iopt = at::TensorOptions().dtype(at::kInt).device(idx_i.device());
  return {p->rowptr, p->rev, p->half.defined() ? p->half : at::empty({0}, iopt), flags};
}

// Every array of the plan of a list (for raw C-ABI callers and tests: they wrap these tensors in a spk_graph_t instead of
// re-deriving them): (rowptr, rev, half, edge_pair, grp_atom0, grp_pair0, grp_tile0, meta) with meta = [sorted, symmetric,
// n_half, n_groups, max_group_atoms, max_group_pairs, filter_pairs (-1 undecided), n_tiles_grouped] (int64, host).
std::vector<Tensor> edge_plan_arrays_op(const Tensor& idx_i, const Tensor& idx_j, int64_t n_atoms, const c10::optional<Tensor>& r_ij) {
  require_device(idx_i, "edge_plan_arrays");
  Tensor r = (r_ij.has_value() && r_ij->defined()) ? *r_ij : Tensor();
  auto p = get_plan(idx_i, idx_j, n_atoms, r);
  auto iopt = at::TensorOptions().dtype(at::kInt).device(idx_i.device());
  auto or_empty = [&](const Tensor& t) { return t.defined() ? t : at::empty({0}, iopt); };
  Tensor meta = at::tensor(std::vector<int64_t>{p->sorted, p->symmetric, p->n_half, p->n_groups, p->max_group_atoms, p->max_group_pairs,
                                                p->filter_pairs, p->n_tiles_grouped}, at::TensorOptions().dtype(at::kLong));
  return {p->rowptr, p->rev, or_empty(p->half), or_empty(p->edge_pair), or_empty(p->grp_atom0), or_empty(p->grp_pair0), or_empty(p->grp_tile0), meta,
          p->idx_i, p->idx_j};
}

// Plan made on the host by the collate function (schnetpack_amd/data.py, DataLoader workers): put it into the cache under the
// key of the device index tensors -- no kernel, no device-to-host copy.  meta = [sorted, symmetric, n_half, n_groups,
// max_group_atoms, max_group_pairs, filter_pairs (-1 unknown), n_tiles_grouped].
void edge_plan_install_op(const Tensor& idx_i, const Tensor& idx_j, int64_t n_atoms, const Tensor& rowptr, const Tensor& rev, const Tensor& half,
                          const Tensor& edge_pair, const Tensor& grp_atom0, const Tensor& grp_pair0, at::IntArrayRef meta) {
  require_device(idx_i, "edge_plan_install");
  TORCH_CHECK(meta.size() >= 8, "edge_plan_install: meta needs 8 entries");
  auto p = std::make_shared<Plan>();
  p->src_i = idx_i;
  p->src_j = idx_j;
  p->idx_i = i64(idx_i, "edge_plan_install");
  p->idx_j = i64(idx_j, "edge_plan_install");
  p->n_atoms = n_atoms;
  p->n_edges = p->idx_i.size(0);
  p->has_r = true;
  auto i32 = [&](const Tensor& t, const char* what, int64_t want) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kInt && t.is_contiguous(), "edge_plan_install: ", what, " must be a contiguous int32 device tensor");
    TORCH_CHECK(want < 0 || t.numel() == want, "edge_plan_install: ", what, " has ", t.numel(), " entries, expected ", want);
    return t;
  };
  p->rowptr = i32(rowptr, "rowptr", n_atoms + 1);
  p->sorted = meta[0] != 0;
  p->symmetric = meta[1] != 0;
  p->rev = i32(rev, "rev", std::max<int64_t>(p->n_edges, 1));
  if (p->symmetric && p->n_edges > 0) {
    p->n_half = meta[2];
    TORCH_CHECK(2 * p->n_half == p->n_edges, "edge_plan_install: a symmetric list has n_edges = 2 n_half");
    p->half = i32(half, "half", p->n_half);
    p->edge_pair = i32(edge_pair, "edge_pair", p->n_edges);
    if (meta[3] > 0) {
      p->n_groups = (int32_t)meta[3];
      p->grp_atom0 = i32(grp_atom0, "grp_atom0", meta[3] + 1);
      p->grp_pair0 = i32(grp_pair0, "grp_pair0", meta[3] + 1);
      // tile offsets of the group-aligned tiling (only the experimental group-local pair kernels read them)
      Tensor tiles = at::floor_divide(at::diff(p->grp_pair0.to(at::kLong)) + 31, 32);
      p->grp_tile0 = at::cat({at::zeros({1}, tiles.options()), at::cumsum(tiles, 0)}).to(at::kInt).contiguous();
      p->max_group_atoms = (int32_t)meta[4];
      p->max_group_pairs = (int32_t)meta[5];
      p->n_tiles_grouped = meta[7];
    }
  }
  p->filter_pairs = (int)meta[6];
  std::vector<uint64_t> key{(uint64_t)idx_i.data_ptr(), (uint64_t)idx_j.data_ptr(), version_of(idx_i), version_of(idx_j),
                            (uint64_t)idx_i.size(0), (uint64_t)n_atoms, (uint64_t)idx_i.device().index(), 1, 1, (uint64_t)idx_i.scalar_type()};
  std::lock_guard<std::mutex> lock(g_mutex);
  g_plans.put(key, p);
}

// static-shape mode (training-step graph replays)
Tensor& static_err_of(int64_t owner, const at::Device& dev) {      // g_mutex held
  Tensor& e = g_static_err[owner];
  if (!e.defined()) e = at::zeros({1}, at::TensorOptions().dtype(at::kInt).device(dev));
  return e;
}
int64_t static_new_op() {
  std::lock_guard<std::mutex> lock(g_mutex);
  return g_static_next_owner++;
}
Tensor static_declare_op(const Tensor& idx, int64_t n_rows, int64_t owner) {
  require_device(idx, "static_declare");
  TORCH_CHECK(idx.scalar_type() == at::kLong && idx.is_contiguous(), "static_declare: needs a contiguous int64 tensor");
  std::lock_guard<std::mutex> lock(g_mutex);
  static_err_of(owner, idx.device());
  for (auto& e : g_static)
    if (e.idx.data_ptr() == idx.data_ptr() && e.n_rows == n_rows) {
      TORCH_CHECK(e.owner == owner, "static_declare: this index buffer is already declared by another StaticLists object");
      return e.rowptr;
    }
  g_static.push_back({idx, at::zeros({n_rows + 1}, at::TensorOptions().dtype(at::kInt).device(idx.device())), n_rows, owner});
  return g_static.back().rowptr;
}
void static_declare_range_op(const Tensor& idx, int64_t hi, int64_t owner) {
  require_device(idx, "static_declare_range");
  TORCH_CHECK(idx.scalar_type() == at::kLong && idx.is_contiguous(), "static_declare_range: needs a contiguous int64 tensor");
  std::lock_guard<std::mutex> lock(g_mutex);
  static_err_of(owner, idx.device());
  for (auto& e : g_static_ranges)
    if (e.idx.data_ptr() == idx.data_ptr() && e.owner == owner) { e.hi = hi; return; }
  g_static_ranges.push_back({idx, hi, owner});
}
void static_refresh_op(int64_t owner) {
  std::lock_guard<std::mutex> lock(g_mutex);
  auto it = g_static_err.find(owner);
  if (it == g_static_err.end()) return;
  int32_t* err = it->second.data_ptr<int32_t>();
  // every declared index of the owner in ONE launch (spk_index_jobs): row pointers + checks of the ascending ones, range checks of the others
  std::vector<spk_index_job_t> jobs;
  Tensor any;
  for (auto& e : g_static) {
    if (e.owner != owner) continue;
    jobs.push_back({e.idx.data_ptr<int64_t>(), e.idx.size(0), e.n_rows, e.rowptr.data_ptr<int32_t>()});
    any = e.idx;
  }
  for (auto& e : g_static_ranges) {
    if (e.owner != owner || e.idx.numel() == 0) continue;
    jobs.push_back({e.idx.data_ptr<int64_t>(), e.idx.numel(), e.hi, nullptr});
    any = e.idx;
  }
  if (jobs.empty()) return;
  c10::DeviceGuard guard(any.device());
  for (size_t k = 0; k < jobs.size(); k += SPK_INDEX_JOBS_MAX)
    check(spk_index_jobs(jobs.data() + k, (int32_t)std::min<size_t>(SPK_INDEX_JOBS_MAX, jobs.size() - k), err, stream_of(any)));
}
bool static_enable_op(bool on) {      // returns the PREVIOUS state (so that a scope can restore it)
  const bool prev = g_static_on;
  g_static_on = on;
  return prev;
}
int64_t static_check_op(int64_t owner) {
  Tensor e;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_static_err.find(owner);
    if (it == g_static_err.end()) return 0;
    e = it->second;
  }
  int64_t f = e.item<int32_t>();
  if (f) e.zero_();
  return f;
}
// drop the declarations of ONE owner (its StaticLists object is gone; so are the graphs that referenced its buffers)
void static_release_op(int64_t owner) {
  std::lock_guard<std::mutex> lock(g_mutex);
  g_static.erase(std::remove_if(g_static.begin(), g_static.end(), [&](const StaticEntry& e) { return e.owner == owner; }), g_static.end());
  g_static_ranges.erase(std::remove_if(g_static_ranges.begin(), g_static_ranges.end(), [&](const StaticRange& e) { return e.owner == owner; }),
                        g_static_ranges.end());
  g_static_err.erase(owner);
}
void static_clear_op() {      // everything, every owner (tests; no captured graph may be replayed afterwards)
  std::lock_guard<std::mutex> lock(g_mutex);
  g_static.clear();
  g_static_ranges.clear();
  g_static_err.clear();
  g_static_on = false;
}
void clear_caches_op() {
  std::lock_guard<std::mutex> lock(g_mutex);
  g_plans.clear();
  g_schnet.clear();
  g_painn.clear();
  g_heads.clear();
}
void weights_changed_op() {
  g_weight_generation.fetch_add(1, std::memory_order_acq_rel);
  spk_filter_table_clear();      // tabulated filters (opt-in experiment) are snapshots of the weights: all stale now
}

// --- CPU key: loud refusal (the dispatcher's own "no kernel" message does not say why); one boxed kernel for every operator
void no_cpu_boxed(const c10::OperatorHandle& op, c10::Stack*) {
  TORCH_CHECK(false, op.schema().name(), ": tensor on cpu -- schnetpack_amd runs on ROCm devices only (there is no CPU fallback)");
}

// --- Meta key: shapes only (tracing / export / shape checks without a device)
Tensor scatter_add_meta(const Tensor& x, const Tensor&, int64_t dim_size, int64_t dim) {
  auto shape = x.sizes().vec();
  shape[at::maybe_wrap_dim(dim, x.dim())] = dim_size;
  return at::empty(shape, x.options());
}
Tensor gather_meta(const Tensor& x, const Tensor& idx, int64_t dim) {
  auto shape = x.sizes().vec();
  shape[at::maybe_wrap_dim(dim, x.dim())] = idx.size(0);
  return at::empty(shape, x.options());
}
Tensor pairwise_meta(const Tensor& R, const Tensor& ii, const Tensor&, const c10::optional<Tensor>&) { return at::empty({ii.size(0), 3}, R.options()); }
Tensor pairwise_backward_meta(const Tensor& gr, const Tensor&, const Tensor&, int64_t n) { return at::empty({n, 3}, gr.options()); }
Tensor dense_meta(const Tensor& x, const Tensor& w, const c10::optional<Tensor>&, int64_t) {
  auto shape = x.sizes().vec();
  shape.back() = w.size(0);
  return at::empty(shape, x.options());
}
std::tuple<Tensor, Tensor> radial_cutoff_meta(const Tensor& d, int64_t, const Tensor& p0, const c10::optional<Tensor>&, double, bool want_phi, bool want_cut) {
  auto shape = d.sizes().vec();
  Tensor fc = want_cut ? at::empty(shape, d.options()) : at::empty({0}, d.options());
  shape.push_back(p0.size(0));
  return {want_phi ? at::empty(shape, d.options()) : at::empty({0}, d.options()), fc};
}
Tensor schnet_meta(const Tensor& x0, const Tensor&, const Tensor&, const Tensor&, at::TensorList, int64_t, int64_t, const Tensor&, const c10::optional<Tensor>&, double) {
  return at::empty_like(x0);
}
std::tuple<Tensor, Tensor> painn_meta(const Tensor& q0, const Tensor&, const Tensor&, const Tensor&, at::TensorList, bool, double, int64_t, const Tensor&,
                                      const c10::optional<Tensor>&, double) {
  return {at::empty_like(q0), at::empty({q0.size(0), 3, q0.size(1)}, q0.options())};
}
std::tuple<Tensor, Tensor> atomwise_meta(const Tensor& x, const Tensor&, const c10::optional<Tensor>&, const Tensor&, const c10::optional<Tensor>&, const Tensor&,
                                         int64_t n_mol, int64_t) {
  return {at::empty({n_mol}, x.options()), at::empty({x.size(0), 1}, x.options())};
}

std::tuple<Tensor, Tensor> dense_forward_meta(const Tensor& x, const Tensor& w, const OptT& b, int64_t act) {
  Tensor y = dense_meta(x, w, b, act);
  return {y, act != SPK_ACT_NONE ? at::empty_like(y) : at::empty({0}, x.options())};
}
std::tuple<Tensor, Tensor, Tensor> schnet_forward_meta(const Tensor& x0, const Tensor&, const Tensor&, const Tensor&, at::TensorList, int64_t, int64_t, const Tensor&,
                                                       const OptT&, double, bool) {
  return {at::empty_like(x0), at::empty({1}, x0.options()), at::empty({1}, x0.options())};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> painn_forward_meta(const Tensor& q0, const Tensor&, const Tensor&, const Tensor&, at::TensorList, bool, double, int64_t,
                                                              const Tensor&, const OptT&, double) {
  return {at::empty_like(q0), at::empty({q0.size(0), 3, q0.size(1)}, q0.options()), at::empty({1}, q0.options()), at::empty({1}, q0.options())};
}
std::tuple<Tensor, Tensor, Tensor> atomwise_forward_meta(const Tensor& x, const Tensor& w1, const OptT&, const Tensor&, const OptT&, const Tensor&, int64_t n_mol, int64_t) {
  return {at::empty({n_mol}, x.options()), at::empty({x.size(0), 1}, x.options()), at::empty({x.size(0), w1.size(0)}, x.options())};
}

Tensor dense_backward_input_meta(const Tensor& gy, const Tensor&, const Tensor& w, int64_t) {
  auto shape = gy.sizes().vec();
  shape.back() = w.size(1);
  return at::empty(shape, gy.options());
}
Tensor radial_cutoff_backward_meta(const Tensor& d, int64_t, const Tensor&, const OptT&, double, const OptT&, const OptT&) { return at::empty_like(d); }
std::tuple<Tensor, Tensor> schnet_backward_meta(const Tensor& gx, const Tensor& r, const Tensor&, const Tensor&, const Tensor&, const Tensor&, at::TensorList, int64_t, int64_t,
                                                const Tensor&, const OptT&, double, bool, bool) {
  return {at::empty_like(r), at::empty_like(gx)};
}
std::tuple<Tensor, Tensor> painn_backward_meta(const OptT&, const OptT&, const Tensor& r, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int64_t n_atoms,
                                               at::TensorList ws, bool, double, int64_t, const Tensor&, const OptT&, double, bool) {
  return {at::empty_like(r), at::empty({n_atoms, ws[0].size(0)}, r.options())};
}
Tensor atomwise_backward_meta(const OptT&, const OptT&, const Tensor& pre, const Tensor& w1, const Tensor&, const Tensor&, int64_t, int64_t) {
  return at::empty({pre.size(0), w1.size(1)}, pre.options());
}

}  // namespace

TORCH_LIBRARY(spk_hip, m) {
  // differentiable operators (what the module mirrors call)
  m.def("scatter_add(Tensor x, Tensor idx_i, int dim_size, int dim=0) -> Tensor");                        // nn/scatter.py:7-34
  m.def("gather(Tensor x, Tensor idx, int dim=0) -> Tensor");                                             // its transpose (x[idx_j], schnet.py:64)
  m.def("pairwise(Tensor R, Tensor idx_i, Tensor idx_j, Tensor? offsets) -> Tensor");                     // atomistic/distances.py:14-26
  m.def("pairwise_backward(Tensor gr, Tensor idx_i, Tensor idx_j, int n_atoms) -> Tensor");
  m.def("dense(Tensor x, Tensor weight, Tensor? bias, int act) -> Tensor");                               // nn/base.py:52-55
  m.def("radial_cutoff(Tensor d, int kind, Tensor p0, Tensor? p1, float cutoff, bool want_phi, bool want_cut) -> (Tensor, Tensor)");  // nn/radial.py, nn/cutoff.py
  m.def("schnet(Tensor x0, Tensor r_ij, Tensor idx_i, Tensor idx_j, Tensor[] weights, int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff) -> Tensor");  // representation/schnet.py:147-173
  m.def("painn(Tensor q0, Tensor r_ij, Tensor idx_i, Tensor idx_j, Tensor[] weights, bool shared_filters, float epsilon, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff) -> (Tensor, Tensor)");  // representation/painn.py:207-256
  m.def("atomwise(Tensor x, Tensor w1, Tensor? b1, Tensor w2, Tensor? b2, Tensor idx_m, int n_mol, int act) -> (Tensor, Tensor)");  // atomistic/atomwise.py:69-88
  // PairwiseDistances -> SchNet -> Atomwise(sum): (energy, scalar_representation); two launches where the list allows it
  m.def("schnet_potential(Tensor x0, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, Tensor[] weights, Tensor[] head, int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> (Tensor, Tensor)");
  m.def("schnet_potential_forces(Tensor? x0, Tensor? embedding, Tensor Z, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, Tensor[] weights, Tensor[] head, int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> (Tensor, Tensor, Tensor)");  // eval: (E, forces, scalar_representation), no autograd
  m.def("painn_potential_forces(Tensor? q0, Tensor? embedding, Tensor Z, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, Tensor[] weights, Tensor[] head, bool shared_filters, float epsilon, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> (Tensor, Tensor, Tensor, Tensor)");  // eval: (E, forces, scalar_representation, vector_representation), no autograd
  m.def("potential_plan(Tensor idx_i, Tensor idx_j, int n_atoms, Tensor idx_m, int n_mol) -> int");
  m.def("eval_guard(Tensor(a) y, Tensor[] params) -> Tensor(a)");      // alias of y whose backward raises the eval-only message
  m.def("schnet_potential_forward(Tensor x0, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, Tensor[] weights, Tensor[] head, int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("schnet_potential_backward(Tensor? gE, Tensor? gx, Tensor x0, Tensor R, Tensor? offsets, Tensor idx_i, Tensor idx_j, Tensor idx_m, int n_mol, Tensor saved, Tensor pre_h, Tensor[] weights, Tensor[] head, int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, int head_act, bool want_gx0) -> (Tensor, Tensor)");
  // raw launchers (no autograd): forward returns the tensors its backward consumes
  m.def("dense_forward(Tensor x, Tensor weight, Tensor? bias, int act) -> (Tensor, Tensor)");
  m.def("dense_backward_input(Tensor gy, Tensor pre, Tensor weight, int act) -> Tensor");
  m.def("radial_cutoff_backward(Tensor d, int kind, Tensor p0, Tensor? p1, float cutoff, Tensor? gphi, Tensor? gfcut) -> Tensor");
  m.def("schnet_forward(Tensor x0, Tensor r_ij, Tensor idx_i, Tensor idx_j, Tensor[] weights, int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, bool save_filters) -> (Tensor, Tensor, Tensor)");
  m.def("schnet_backward(Tensor gx, Tensor r_ij, Tensor saved, Tensor scratch, Tensor idx_i, Tensor idx_j, Tensor[] weights, int n_filters, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, bool saved_filters, bool want_gx0) -> (Tensor, Tensor)");
  m.def("painn_forward(Tensor q0, Tensor r_ij, Tensor idx_i, Tensor idx_j, Tensor[] weights, bool shared_filters, float epsilon, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("painn_backward(Tensor? gq, Tensor? gmu, Tensor r_ij, Tensor saved, Tensor scratch, Tensor idx_i, Tensor idx_j, int n_atoms, Tensor[] weights, bool shared_filters, float epsilon, int rbf_kind, Tensor rbf_p0, Tensor? rbf_p1, float cutoff, bool want_gq0) -> (Tensor, Tensor)");
  m.def("atomwise_forward(Tensor x, Tensor w1, Tensor? b1, Tensor w2, Tensor? b2, Tensor idx_m, int n_mol, int act) -> (Tensor, Tensor, Tensor)");
  m.def("atomwise_backward(Tensor? gE, Tensor? gy_atom, Tensor pre, Tensor w1, Tensor w2, Tensor idx_m, int n_mol, int act) -> Tensor");
  m.def("edge_plan(Tensor idx_i, Tensor idx_j, int n_atoms, Tensor? r_ij, float cutoff=0.0, int force_filter=-1) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("edge_plan_arrays(Tensor idx_i, Tensor idx_j, int n_atoms, Tensor? r_ij) -> Tensor[]");
  m.def("edge_plan_install(Tensor idx_i, Tensor idx_j, int n_atoms, Tensor rowptr, Tensor rev, Tensor half, Tensor edge_pair, Tensor grp_atom0, Tensor grp_pair0, int[] meta) -> ()");
  // static-shape mode + cache control (host-side state)
  m.def("static_new() -> int", static_new_op);
  m.def("static_declare(Tensor idx, int n_rows, int owner=0) -> Tensor");
  m.def("static_declare_range(Tensor idx, int hi, int owner=0) -> ()");
  m.def("static_refresh(int owner=0) -> ()", static_refresh_op);
  m.def("static_enable(bool on) -> bool", static_enable_op);
  m.def("static_check(int owner=0) -> int", static_check_op);
  m.def("static_release(int owner) -> ()", static_release_op);
  m.def("static_clear() -> ()", static_clear_op);
  m.def("clear_caches() -> ()", clear_caches_op);
  m.def("weights_changed() -> ()", weights_changed_op);
  // training regime: operators closed under differentiation (spk_torch_train.h)
  train_defs(m);
  fm_defs(m);
}

TORCH_LIBRARY_IMPL(spk_hip, CUDA, m) {   // "CUDA" is the dispatch key of ROCm devices in PyTorch-ROCm
  m.impl("scatter_add", scatter_add_raw);
  m.impl("gather", gather_raw);
  m.impl("pairwise", pairwise_raw);
  m.impl("pairwise_backward", pairwise_bwd_raw);
  m.impl("dense", dense_dev);
  m.impl("radial_cutoff", radial_cutoff_raw);
  m.impl("schnet", schnet_dev);
  m.impl("painn", painn_dev);
  m.impl("atomwise", atomwise_dev);
  m.impl("schnet_potential", schnet_potential_dev);
  m.impl("schnet_potential_forward", schnet_potential_forward_raw);
  m.impl("schnet_potential_forces", schnet_potential_forces_raw);
  m.impl("painn_potential_forces", painn_potential_forces_raw);
  m.impl("eval_guard", eval_guard_dev);
  m.impl("potential_plan", potential_plan_op);
  m.impl("schnet_potential_backward", schnet_potential_backward_raw);
  m.impl("dense_forward", dense_raw);
  m.impl("dense_backward_input", dense_bwd_input_raw);
  m.impl("radial_cutoff_backward", radial_cutoff_bwd_raw);
  m.impl("schnet_forward", schnet_forward_op);
  m.impl("schnet_backward", schnet_backward_op);
  m.impl("painn_forward", painn_forward_op);
  m.impl("painn_backward", painn_backward_op);
  m.impl("atomwise_forward", atomwise_forward_raw);
  m.impl("atomwise_backward", atomwise_backward_op);
  m.impl("edge_plan", edge_plan_op);
  m.impl("edge_plan_arrays", edge_plan_arrays_op);
  m.impl("static_declare", static_declare_op);
  m.impl("static_declare_range", static_declare_range_op);
  m.impl("edge_plan_install", edge_plan_install_op);
  train_impl_device(m);
  fm_impl_device(m);
}

TORCH_LIBRARY_IMPL(spk_hip, Autograd, m) {
  m.impl("scatter_add", scatter_add_ad);
  m.impl("gather", gather_ad);
  m.impl("pairwise", pairwise_ad);
  m.impl("pairwise_backward", pairwise_backward_ad);
  m.impl("dense", dense_ad);
  m.impl("radial_cutoff", radial_cutoff_ad);
  m.impl("schnet", schnet_ad);
  m.impl("painn", painn_ad);
  m.impl("atomwise", atomwise_ad);
  m.impl("schnet_potential", schnet_potential_ad);
  m.impl("eval_guard", eval_guard_ad);
  train_impl_autograd(m);
  fm_impl_autograd(m);
}

TORCH_LIBRARY_IMPL(spk_hip, CPU, m) {
  for (const char* name : {"scatter_add", "gather", "pairwise", "pairwise_backward", "dense", "radial_cutoff", "schnet", "painn", "atomwise",
                           "dense_forward", "dense_backward_input", "radial_cutoff_backward", "schnet_forward", "schnet_backward", "painn_forward",
                           "painn_backward", "atomwise_forward", "atomwise_backward", "edge_plan", "static_declare", "static_declare_range", "schnet_potential",
                           "schnet_potential_forward", "schnet_potential_backward", "schnet_potential_forces", "painn_potential_forces", "potential_plan"})
    m.impl(name, torch::CppFunction::makeFromBoxedFunction<&no_cpu_boxed>());
  for (const char* name : kTrainOps) m.impl(name, torch::CppFunction::makeFromBoxedFunction<&no_cpu_boxed>());
  for (const char* name : kFmOps) m.impl(name, torch::CppFunction::makeFromBoxedFunction<&no_cpu_boxed>());
}

TORCH_LIBRARY_IMPL(spk_hip, Meta, m) {
  m.impl("scatter_add", scatter_add_meta);
  m.impl("gather", gather_meta);
  m.impl("pairwise", pairwise_meta);
  m.impl("pairwise_backward", pairwise_backward_meta);
  m.impl("dense", dense_meta);
  m.impl("radial_cutoff", radial_cutoff_meta);
  m.impl("schnet", schnet_meta);
  m.impl("painn", painn_meta);
  m.impl("atomwise", atomwise_meta);
  m.impl("dense_forward", dense_forward_meta);
  m.impl("schnet_forward", schnet_forward_meta);
  m.impl("painn_forward", painn_forward_meta);
  m.impl("atomwise_forward", atomwise_forward_meta);
  m.impl("dense_backward_input", dense_backward_input_meta);
  m.impl("radial_cutoff_backward", radial_cutoff_backward_meta);
  m.impl("schnet_backward", schnet_backward_meta);
  m.impl("painn_backward", painn_backward_meta);
  m.impl("atomwise_backward", atomwise_backward_meta);
  m.impl("schnet_potential", schnet_potential_meta);
  m.impl("schnet_potential_forward", schnet_potential_forward_meta);
  m.impl("schnet_potential_forces", schnet_potential_forces_meta);
  m.impl("painn_potential_forces", painn_potential_forces_meta);
  m.impl("eval_guard", eval_guard_dev);
  m.impl("schnet_potential_backward", schnet_potential_backward_meta);
  train_impl_meta(m);
  fm_impl_meta(m);
}
