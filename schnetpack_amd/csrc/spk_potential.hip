// Deployment runtime: a self-contained force call on a flat weight file (include/spk_hip.h, "deployment runtime").
// Host C++ around the C-ABI entry points of this library -- what schnetpack_amd/{model,atomistic,ops}.py do with
// torch tensors is done here with hipMalloc'ed buffers, so that an MD code can link libspk_hip.so alone.
// Replaces: spkdeploy:16-40 (export), pair_schnetpack.cpp:128 (load), :196-283 (inputs), :328-350 (forward, outputs).
#include "spk_common.h"
#include <cstring>
#include <cmath>
#include <rocprim/rocprim.hpp>
#include <map>
#include <string>
#include <vector>

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return SPK_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 4 + 256;     // grow-only with head room: MD lists fluctuate by a few per cent
    SPK_HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return SPK_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return (T*)p; }
};

#pragma pack(push, 1)
struct FileHeader {
  char magic[8];
  int32_t version, kind, F, nf, L, n_rbf, rbf_kind, head_hidden, head_act, emb_rows, extensive, n_atomref, n_tensors, r0, r1, r2;
  float cutoff, eps, e_mean, rf;
};
struct FileEntry { char name[32]; int64_t n_floats, offset; };
#pragma pack(pop)

__global__ void k_fill_f32(float* __restrict__ p, float v, int64_t n) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) p[k] = v;
}
__global__ void k_half_flags(const int32_t* __restrict__ rev, int64_t n, unsigned char* __restrict__ flags) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    flags[e] = rev[e] > (int32_t)e ? 1 : 0;
}

}  // namespace

struct spk_potential {
  FileHeader h;
  hipStream_t stream = nullptr;
  float* d_w = nullptr;                                  // all weights (+ transposed copies) in one allocation
  std::map<std::string, const float*> t;                 // device pointer of every tensor
  std::vector<float> atomref;
  std::vector<spk_schnet_layer_t> sl;
  std::vector<spk_painn_layer_t> pl;
  spk_schnet_t sm;
  spk_painn_t pm;
  float* wpack = nullptr;
  spk_radial_t rb;
  // per-call device buffers
  DevBuf z, R, ii, jj, off, idxm, rij, rowptr, rev, half, scr, x0, xo, mu, saved, scratch, pre, gx, gE, gr, gR, E, flags, tmp,
      cell, pbc, nblws, nblrow, shifts;
  // current list
  spk_graph_t g;
  bool cell_list = false;
  int64_t list_atoms = -1, list_mol = -1;
  float list_skin = -1.f;
  std::vector<float> R_ref, cell_ref;
  std::vector<int64_t> idxm_ref;
  std::vector<uint8_t> pbc_ref;
  // host staging for unsorted explicit lists
  std::vector<int64_t> si, sj, cnt;
  std::vector<float> soff;
  std::vector<float> hE, hF;
  // what is currently in the device copies of z / idx_m
  std::vector<int64_t> z_up, m_up;
  int64_t m_up_n = -1;
  bool m_up_valid = false;
};

namespace {

int parse_and_upload(spk_potential* p, const unsigned char* blob, int64_t n) {
  SPK_CHECK_ARG(blob && n >= (int64_t)sizeof(FileHeader), "spk_potential: file too short");
  memcpy(&p->h, blob, sizeof(FileHeader));
  const FileHeader& h = p->h;
  SPK_CHECK_ARG(memcmp(h.magic, "SPKHIP01", 8) == 0, "spk_potential: bad magic (not a file written by schnetpack_amd.deploy)");
  SPK_CHECK_ARG(h.version == 1, "spk_potential: unsupported file version %d", h.version);
  SPK_CHECK_ARG(h.kind == 0 || h.kind == 1, "spk_potential: unknown representation kind %d", h.kind);
  SPK_CHECK_ARG(h.F > 0 && h.nf > 0 && h.L >= 0 && h.L <= 64 && h.n_rbf > 0 && h.emb_rows > 0 && h.n_tensors > 0 && h.n_tensors < 4096,
                "spk_potential: implausible header");
  SPK_CHECK_ARG(h.cutoff > 0.f, "spk_potential: cutoff must be positive");
  const int64_t table_end = (int64_t)sizeof(FileHeader) + (int64_t)h.n_tensors * (int64_t)sizeof(FileEntry);
  SPK_CHECK_ARG(n >= table_end, "spk_potential: truncated tensor table");
  const int64_t data0 = (table_end + 63) / 64 * 64;
  std::vector<FileEntry> ent(h.n_tensors);
  memcpy(ent.data(), blob + sizeof(FileHeader), (size_t)h.n_tensors * sizeof(FileEntry));
  std::map<std::string, const FileEntry*> by_name;
  int64_t total = 0;
  for (auto& e : ent) {
    e.name[31] = 0;
    SPK_CHECK_ARG(e.n_floats >= 0 && e.offset >= 0 && data0 + 4 * (e.offset + e.n_floats) <= n, "spk_potential: tensor %s outside the file", e.name);
    by_name[e.name] = &e;
    total = std::max(total, e.offset + e.n_floats);
  }
  // expected tensors and shapes
  struct Need { std::string name; int64_t n; bool transpose; int rows, cols; };
  std::vector<Need> need;
  const int64_t F = h.F, nf = h.nf, K = h.n_rbf, H = h.head_hidden;
  need.push_back({"embedding", (int64_t)h.emb_rows * F, false, 0, 0});
  need.push_back({"rbf_p0", K, false, 0, 0});
  need.push_back({"rbf_p1", K, false, 0, 0});
  need.push_back({"head_w1", H * F, false, 0, 0});
  need.push_back({"head_b1", H, false, 0, 0});
  need.push_back({"head_w2", H, false, 0, 0});
  need.push_back({"head_b2", 1, false, 0, 0});
  for (int l = 0; l < h.L; ++l) {
    const std::string pre = "l" + std::to_string(l) + ".";
    if (h.kind == 0) {
      need.push_back({pre + "in2f_w", nf * F, true, (int)nf, (int)F});
      need.push_back({pre + "fn_w1", nf * K, false, 0, 0});
      need.push_back({pre + "fn_b1", nf, false, 0, 0});
      need.push_back({pre + "fn_w2", nf * nf, false, 0, 0});
      need.push_back({pre + "fn_b2", nf, false, 0, 0});
      need.push_back({pre + "f2out_w1", F * nf, true, (int)F, (int)nf});
      need.push_back({pre + "f2out_b1", F, false, 0, 0});
      need.push_back({pre + "f2out_w2", F * F, true, (int)F, (int)F});
      need.push_back({pre + "f2out_b2", F, false, 0, 0});
    } else {
      need.push_back({pre + "ctx_w1", F * F, true, (int)F, (int)F});
      need.push_back({pre + "ctx_b1", F, false, 0, 0});
      need.push_back({pre + "ctx_w2", 3 * F * F, true, (int)(3 * F), (int)F});
      need.push_back({pre + "ctx_b2", 3 * F, false, 0, 0});
      need.push_back({pre + "mix_w", 2 * F * F, true, (int)(2 * F), (int)F});
      need.push_back({pre + "ictx_w1", F * 2 * F, true, (int)F, (int)(2 * F)});
      need.push_back({pre + "ictx_b1", F, false, 0, 0});
      need.push_back({pre + "ictx_w2", 3 * F * F, true, (int)(3 * F), (int)F});
      need.push_back({pre + "ictx_b2", 3 * F, false, 0, 0});
    }
  }
  if (h.kind == 1) {
    need.push_back({"filt_w", (int64_t)h.L * 3 * F * K, false, 0, 0});
    need.push_back({"filt_b", (int64_t)h.L * 3 * F, false, 0, 0});
  }
  int64_t extra = 0;
  for (auto& nd : need) {
    auto it = by_name.find(nd.name);
    SPK_CHECK_ARG(it != by_name.end(), "spk_potential: tensor %s missing from the file", nd.name.c_str());
    SPK_CHECK_ARG(it->second->n_floats == nd.n, "spk_potential: tensor %s has %lld floats, expected %lld", nd.name.c_str(),
                  (long long)it->second->n_floats, (long long)nd.n);
    if (nd.transpose) extra += (nd.n + 15) / 16 * 16;
  }
  if (h.n_atomref > 0) {
    auto it = by_name.find("atomref");
    SPK_CHECK_ARG(it != by_name.end() && it->second->n_floats == h.n_atomref, "spk_potential: atomref table missing / wrong size");
    p->atomref.resize(h.n_atomref);
    memcpy(p->atomref.data(), blob + data0 + 4 * it->second->offset, 4 * (size_t)h.n_atomref);
  }
  SPK_CHECK_ARG(spk_atomwise_supported(h.F, h.head_hidden, h.head_act), "spk_potential: head %d -> %d (activation %d) has no fused kernel",
                h.F, h.head_hidden, h.head_act);
  // host image = file data + transposed copies, one upload
  total = (total + 15) / 16 * 16;
  std::vector<float> img((size_t)(total + extra), 0.f);
  memcpy(img.data(), blob + data0, (size_t)std::min<int64_t>(4 * total, n - data0));
  SPK_HIP_TRY(hipMalloc((void**)&p->d_w, img.size() * sizeof(float) + 256));
  for (auto& e : ent) p->t[e.name] = p->d_w + e.offset;
  int64_t cur = total;
  for (auto& nd : need) {
    if (!nd.transpose) continue;
    const float* src = img.data() + by_name[nd.name]->offset;
    float* dst = img.data() + cur;
    for (int r = 0; r < nd.rows; ++r)
      for (int c = 0; c < nd.cols; ++c) dst[(size_t)c * nd.rows + r] = src[(size_t)r * nd.cols + c];
    p->t[nd.name + "T"] = p->d_w + cur;
    cur += (nd.n + 15) / 16 * 16;
  }
  SPK_HIP_TRY(hipMemcpy(p->d_w, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
  // parameter blocks
  auto T = [&](const std::string& s) { return p->t.at(s); };
  p->rb.kind = h.rbf_kind; p->rb.n_rbf = h.n_rbf; p->rb.p0 = T("rbf_p0"); p->rb.p1 = T("rbf_p1"); p->rb.cutoff = h.cutoff;
  int64_t n_pack = 0;
  if (h.kind == 0) {
    p->sl.resize(std::max(1, h.L));
    for (int l = 0; l < h.L; ++l) {
      const std::string pre = "l" + std::to_string(l) + ".";
      spk_schnet_layer_t& y = p->sl[l];
      y.in2f_w = T(pre + "in2f_w"); y.fn_w1 = T(pre + "fn_w1"); y.fn_b1 = T(pre + "fn_b1"); y.fn_w2 = T(pre + "fn_w2");
      y.fn_b2 = T(pre + "fn_b2"); y.f2out_w1 = T(pre + "f2out_w1"); y.f2out_b1 = T(pre + "f2out_b1");
      y.f2out_w2 = T(pre + "f2out_w2"); y.f2out_b2 = T(pre + "f2out_b2");
      y.in2f_wT = T(pre + "in2f_wT"); y.f2out_w1T = T(pre + "f2out_w1T"); y.f2out_w2T = T(pre + "f2out_w2T");
    }
    p->sm.n_atom_basis = h.F; p->sm.n_filters = h.nf; p->sm.n_interactions = h.L; p->sm.reserved = 0;
    p->sm.layers = p->sl.data(); p->sm.wpack = nullptr;
    n_pack = h.L > 0 ? spk_schnet_packed_floats(&p->sm) : 0;
  } else {
    p->pl.resize(std::max(1, h.L));
    for (int l = 0; l < h.L; ++l) {
      const std::string pre = "l" + std::to_string(l) + ".";
      spk_painn_layer_t& y = p->pl[l];
      y.ctx_w1 = T(pre + "ctx_w1"); y.ctx_b1 = T(pre + "ctx_b1"); y.ctx_w2 = T(pre + "ctx_w2"); y.ctx_b2 = T(pre + "ctx_b2");
      y.filt_w = T("filt_w") + (int64_t)l * 3 * F * K; y.filt_b = T("filt_b") + (int64_t)l * 3 * F;
      y.mix_w = T(pre + "mix_w"); y.ictx_w1 = T(pre + "ictx_w1"); y.ictx_b1 = T(pre + "ictx_b1");
      y.ictx_w2 = T(pre + "ictx_w2"); y.ictx_b2 = T(pre + "ictx_b2");
      y.ctx_w1T = T(pre + "ctx_w1T"); y.ctx_w2T = T(pre + "ctx_w2T"); y.mix_wT = T(pre + "mix_wT");
      y.ictx_w1T = T(pre + "ictx_w1T"); y.ictx_w2T = T(pre + "ictx_w2T");
    }
    p->pm.n_atom_basis = h.F; p->pm.n_interactions = h.L; p->pm.epsilon = h.eps; p->pm.reserved = 0;
    p->pm.layers = p->pl.data(); p->pm.wpack = nullptr;
    n_pack = h.L > 0 ? spk_painn_packed_floats(&p->pm) : 0;
  }
  SPK_HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  if (n_pack > 0) {
    SPK_HIP_TRY(hipMalloc((void**)&p->wpack, (size_t)n_pack * sizeof(float)));
    int rc = h.kind == 0 ? spk_schnet_pack_weights_f32(&p->sm, p->wpack, p->stream) : spk_painn_pack_weights_f32(&p->pm, p->wpack, p->stream);
    if (rc) return rc;
    SPK_HIP_TRY(hipStreamSynchronize(p->stream));
    if (h.kind == 0) p->sm.wpack = p->wpack; else p->pm.wpack = p->wpack;
  }
  memset(&p->g, 0, sizeof(p->g));
  return SPK_OK;
}

int create(const unsigned char* blob, int64_t n, spk_potential_t** out) {
  SPK_CHECK_ARG(out, "spk_potential: null output handle");
  *out = nullptr;
  spk_potential* p = new spk_potential();
  int rc = parse_and_upload(p, blob, n);
  if (rc) { spk_potential_free(p); return rc; }
  *out = p;
  return SPK_OK;
}

// plan of the list now in p->ii / p->jj / p->rij  (what ops.EdgePlan does on the Python side)
int plan_list(spk_potential* p, int64_t N, int64_t E, bool want_filter) {
  hipStream_t s = p->stream;
  int rc;
  if ((rc = p->rowptr.ensure((size_t)(N + 1) * 4))) return rc;
  if ((rc = p->rev.ensure((size_t)std::max<int64_t>(E, 1) * 4))) return rc;
  if ((rc = p->scr.ensure(64))) return rc;
  int32_t fl[4] = {0, 0, 0, 0};
  rc = spk_edge_plan(p->ii.as<int64_t>(), p->jj.as<int64_t>(), E > 0 ? p->rij.as<float>() : nullptr, E, N, p->rowptr.as<int32_t>(),
                     p->rev.as<int32_t>(), p->scr.as<int32_t>(), fl, s);
  if (rc) return rc;
  const bool sorted = fl[0] != 0;
  bool symmetric = fl[2] != 0;
  int64_t n_half = 0;
  if (symmetric && E > 0) {
    if ((rc = p->half.ensure((size_t)E * 4 + 16))) return rc;     // list [<= E] + count
    if ((rc = p->flags.ensure((size_t)E))) return rc;
    int32_t* cnt = p->half.as<int32_t>() + E;
    hipLaunchKernelGGL(k_half_flags, dim3(spk_grid_for(E, 256, 4096)), dim3(256), 0, s, p->rev.as<int32_t>(), E, p->flags.as<unsigned char>());
    SPK_LAUNCH_CHECK();
    size_t tmp_bytes = 0;
    rocprim::counting_iterator<int32_t> ids(0);
    SPK_HIP_TRY(rocprim::select(nullptr, tmp_bytes, ids, p->flags.as<unsigned char>(), p->half.as<int32_t>(), cnt, (size_t)E, s));
    if ((rc = p->tmp.ensure(tmp_bytes + 16))) return rc;
    SPK_HIP_TRY(rocprim::select(p->tmp.p, tmp_bytes, ids, p->flags.as<unsigned char>(), p->half.as<int32_t>(), cnt, (size_t)E, s));
    int32_t c = 0;
    SPK_HIP_TRY(hipMemcpyAsync(&c, cnt, 4, hipMemcpyDeviceToHost, s));
    SPK_HIP_TRY(hipStreamSynchronize(s));
    n_half = c;
    if (2 * n_half != E) { symmetric = false; n_half = 0; }
  }
  spk_graph_t& g = p->g;
  memset(&g, 0, sizeof(g));
  g.n_atoms = N; g.n_edges = E; g.idx_i = p->ii.as<int64_t>(); g.idx_j = p->jj.as<int64_t>();
  g.rowptr = sorted ? p->rowptr.as<int32_t>() : nullptr;
  g.sorted = sorted; g.symmetric = symmetric;
  g.rev = symmetric ? p->rev.as<int32_t>() : nullptr;
  g.half = (symmetric && n_half > 0) ? p->half.as<int32_t>() : nullptr;
  g.n_half = n_half;
  // SchNet: per-call compaction of the pair list; PaiNN: hint that the list has a skin (keeps the row kernels, which mask dead pairs)
  g.filter_pairs = (want_filter && (p->h.kind == 1 || (symmetric && n_half > 0))) ? 1 : 0;
  return SPK_OK;
}

// inputs are on the device (z, R, ii, jj, off, idxm); runs the model and downloads energy / forces
int run_model(spk_potential* p, int64_t N, int64_t E, int64_t M, bool new_list, bool want_filter, bool have_off,
              const int64_t* host_z, const int64_t* host_idx_m, float* host_energy, float* host_forces) {
  const FileHeader& h = p->h;
  hipStream_t s = p->stream;
  const int F = h.F, H = h.head_hidden;
  int rc;
  if ((rc = p->rij.ensure((size_t)std::max<int64_t>(E, 1) * 12))) return rc;
  if (E > 0) {
    rc = spk_pairwise_f32(p->R.as<float>(), p->ii.as<int64_t>(), p->jj.as<int64_t>(), have_off ? p->off.as<float>() : nullptr, E, p->rij.as<float>(), s);
    if (rc) return rc;
  }
  if (new_list && (rc = plan_list(p, N, E, want_filter))) return rc;
  const spk_graph_t* g = &p->g;
  if ((rc = p->x0.ensure((size_t)N * F * 4))) return rc;
  if ((rc = p->xo.ensure((size_t)N * F * 4))) return rc;
  if ((rc = p->gx.ensure((size_t)N * F * 4))) return rc;
  if ((rc = p->pre.ensure((size_t)N * H * 4))) return rc;
  if ((rc = p->E.ensure((size_t)M * 4))) return rc;
  if ((rc = p->gE.ensure((size_t)M * 4))) return rc;
  if ((rc = p->gr.ensure((size_t)std::max<int64_t>(E, 1) * 12))) return rc;
  if ((rc = p->gR.ensure((size_t)N * 12))) return rc;
  if ((rc = spk_embedding_f32(p->t.at("embedding"), p->z.as<int64_t>(), N, F, p->x0.as<float>(), s))) return rc;
  int64_t n_saved, n_scratch;
  if (h.kind == 0) {
    p->sm.reserved = 1;
    n_saved = spk_schnet_saved_floats_graph(&p->sm, g, &p->rb);
    n_scratch = spk_schnet_scratch_floats(&p->sm, N);
  } else {
    n_saved = spk_painn_saved_floats(&p->pm, N);
    n_scratch = spk_painn_scratch_floats(&p->pm, N);
    if ((rc = p->mu.ensure((size_t)N * 3 * F * 4))) return rc;
  }
  if ((rc = p->saved.ensure((size_t)std::max<int64_t>(n_saved, 1) * 4))) return rc;
  if ((rc = p->scratch.ensure((size_t)std::max<int64_t>(n_scratch, 1) * 4))) return rc;
  if (h.kind == 0)
    rc = spk_schnet_forward_f32(&p->sm, g, &p->rb, p->x0.as<float>(), p->rij.as<float>(), p->xo.as<float>(), p->saved.as<float>(), p->scratch.as<float>(), s);
  else
    rc = spk_painn_forward_f32(&p->pm, g, &p->rb, p->x0.as<float>(), p->rij.as<float>(), p->xo.as<float>(), p->mu.as<float>(), p->saved.as<float>(),
                               p->scratch.as<float>(), s);
  if (rc) return rc;
  rc = spk_atomwise_fwd_f32(p->xo.as<float>(), p->t.at("head_w1"), p->t.at("head_b1"), p->t.at("head_w2"), p->t.at("head_b2"), p->idxm.as<int64_t>(),
                            N, F, H, h.head_act, M, p->pre.as<float>(), nullptr, p->E.as<float>(), s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_fill_f32, dim3(spk_grid_for(M, 256, 1024)), dim3(256), 0, s, p->gE.as<float>(), 1.0f, M);   // grad_outputs = ones (atomistic/response.py:63)
  SPK_LAUNCH_CHECK();
  rc = spk_atomwise_bwd_f32(p->gE.as<float>(), nullptr, p->pre.as<float>(), p->t.at("head_w1"), p->t.at("head_w2"), p->idxm.as<int64_t>(), N, F, H,
                            h.head_act, M, p->gx.as<float>(), s);
  if (rc) return rc;
  if (h.kind == 0)
    rc = spk_schnet_backward_f32(&p->sm, g, &p->rb, p->gx.as<float>(), p->rij.as<float>(), p->saved.as<float>(), p->scratch.as<float>(), p->gr.as<float>(),
                                 nullptr, s);
  else
    rc = spk_painn_backward_f32(&p->pm, g, &p->rb, p->gx.as<float>(), nullptr, p->rij.as<float>(), p->saved.as<float>(), p->scratch.as<float>(),
                                p->gr.as<float>(), nullptr, s);
  if (rc) return rc;
  if ((rc = spk_pairwise_bwd_graph_f32(p->gr.as<float>(), g, p->gR.as<float>(), s))) return rc;
  p->hE.resize((size_t)M);
  p->hF.resize((size_t)N * 3);
  SPK_HIP_TRY(hipMemcpyAsync(p->hE.data(), p->E.p, (size_t)M * 4, hipMemcpyDeviceToHost, s));
  SPK_HIP_TRY(hipMemcpyAsync(p->hF.data(), p->gR.p, (size_t)N * 12, hipMemcpyDeviceToHost, s));
  SPK_HIP_TRY(hipStreamSynchronize(s));
  for (int64_t k = 0; k < 3 * N; ++k) host_forces[k] = -p->hF[k];          // forces = -dE/dR (atomistic/response.py:76)
  // AddOffsets (transform/atomistic.py:300-324), fp32 like the deployed reference model (spkdeploy:24-26)
  std::vector<float> n_at((size_t)M, 0.f), y0((size_t)M, 0.f);
  for (int64_t a = 0; a < N; ++a) {
    const int64_t m = host_idx_m ? host_idx_m[a] : 0;
    n_at[m] += 1.f;
    if (!p->atomref.empty()) y0[m] += p->atomref[host_z[a]];
  }
  for (int64_t m = 0; m < M; ++m) {
    float e = p->hE[m];
    if (h.e_mean != 0.f) e += h.extensive ? h.e_mean * n_at[m] : h.e_mean;
    if (!p->atomref.empty()) e += h.extensive ? y0[m] : (n_at[m] > 0.f ? y0[m] / n_at[m] : 0.f);
    host_energy[m] = e;
  }
  return SPK_OK;
}

int check_atoms(const spk_potential* p, int64_t N, const int64_t* z, const float* R, int64_t M, const int64_t* idx_m, const char* who) {
  SPK_CHECK_ARG(p, "%s: null handle", who);
  SPK_CHECK_ARG(N > 0 && N < (1LL << 31) && M > 0 && M <= N, "%s: n_atoms = %lld, n_mol = %lld", who, (long long)N, (long long)M);
  SPK_CHECK_ARG(z && R, "%s: null atomic numbers / positions", who);
  SPK_CHECK_ARG(idx_m || M == 1, "%s: idx_m is required for n_mol > 1", who);
  const int64_t zmax = p->atomref.empty() ? p->h.emb_rows : std::min<int64_t>(p->h.emb_rows, (int64_t)p->atomref.size());
  for (int64_t a = 0; a < N; ++a) {
    SPK_CHECK_ARG(z[a] >= 0 && z[a] < zmax, "%s: atomic number %lld of atom %lld outside the embedding table [0, %lld)", who, (long long)z[a],
                  (long long)a, (long long)zmax);
    if (idx_m) SPK_CHECK_ARG(idx_m[a] >= 0 && idx_m[a] < M && (a == 0 || idx_m[a] >= idx_m[a - 1]), "%s: idx_m must be ascending in [0, n_mol)", who);
  }
  return SPK_OK;
}

int upload_atoms(spk_potential* p, int64_t N, const int64_t* z, const float* R, const int64_t* idx_m) {
  int rc;
  hipStream_t s = p->stream;
  if (p->z.cap < (size_t)N * 8 || p->idxm.cap < (size_t)N * 8) { p->z_up.clear(); p->m_up_valid = false; }   // buffers move: re-upload
  if ((rc = p->z.ensure((size_t)N * 8))) return rc;
  if ((rc = p->R.ensure((size_t)N * 12))) return rc;
  if ((rc = p->idxm.ensure((size_t)N * 8))) return rc;
  SPK_HIP_TRY(hipMemcpyAsync(p->R.p, R, (size_t)N * 12, hipMemcpyHostToDevice, s));
  // atomic numbers and system indices rarely change between the calls of an MD run: upload them when they do
  const bool same_z = p->z_up.size() == (size_t)N && memcmp(p->z_up.data(), z, (size_t)N * 8) == 0;
  const bool same_m = p->m_up_valid && (idx_m ? (p->m_up.size() == (size_t)N && memcmp(p->m_up.data(), idx_m, (size_t)N * 8) == 0)
                                             : (p->m_up.empty() && p->m_up_n == N));
  if (!same_z) {
    SPK_HIP_TRY(hipMemcpyAsync(p->z.p, z, (size_t)N * 8, hipMemcpyHostToDevice, s));
    p->z_up.assign(z, z + N);
  }
  if (!same_m) {
    if (idx_m) { SPK_HIP_TRY(hipMemcpyAsync(p->idxm.p, idx_m, (size_t)N * 8, hipMemcpyHostToDevice, s)); p->m_up.assign(idx_m, idx_m + N); }
    else { if ((rc = spk_zero_async(p->idxm.p, (size_t)N * 8, s))) return rc; p->m_up.clear(); }
    p->m_up_n = N; p->m_up_valid = true;
  }
  return SPK_OK;
}

}  // namespace

extern "C" int spk_potential_from_memory(const void* blob, int64_t n_bytes, spk_potential_t** out) {
  return create((const unsigned char*)blob, n_bytes, out);
}

extern "C" int spk_potential_load(const char* path, spk_potential_t** out) {
  SPK_CHECK_ARG(path && out, "spk_potential_load: null argument");
  FILE* f = fopen(path, "rb");
  SPK_CHECK_ARG(f, "spk_potential_load: cannot open %s", path);
  std::vector<unsigned char> buf;
  unsigned char chunk[1 << 16];
  size_t got;
  while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
  fclose(f);
  return create(buf.data(), (int64_t)buf.size(), out);
}

extern "C" void spk_potential_free(spk_potential_t* p) {
  if (!p) return;
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  DevBuf* bufs[] = {&p->z, &p->R, &p->ii, &p->jj, &p->off, &p->idxm, &p->rij, &p->rowptr, &p->rev, &p->half, &p->scr, &p->x0, &p->xo, &p->mu,
                    &p->saved, &p->scratch, &p->pre, &p->gx, &p->gE, &p->gr, &p->gR, &p->E, &p->flags, &p->tmp, &p->cell, &p->pbc, &p->nblws,
                    &p->nblrow, &p->shifts};
  for (DevBuf* b : bufs) b->release();
  if (p->wpack) (void)hipFree(p->wpack);
  if (p->d_w) (void)hipFree(p->d_w);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

extern "C" int spk_potential_info(const spk_potential_t* p, int32_t* info, float* cutoff) {
  SPK_CHECK_ARG(p, "spk_potential_info: null handle");
  if (info) {
    info[0] = p->h.kind; info[1] = p->h.F; info[2] = p->h.L; info[3] = p->h.n_rbf; info[4] = p->h.rbf_kind; info[5] = p->h.nf;
    info[6] = p->h.head_hidden; info[7] = p->h.emb_rows;
  }
  if (cutoff) *cutoff = p->h.cutoff;
  return SPK_OK;
}

extern "C" int spk_potential_compute(spk_potential_t* p, int64_t N, const int64_t* z, const float* R, int64_t E, const int64_t* idx_i,
                                     const int64_t* idx_j, const float* offsets, int64_t M, const int64_t* idx_m, float* energy,
                                     float* forces) {
  int rc = check_atoms(p, N, z, R, M, idx_m, "spk_potential_compute");
  if (rc) return rc;
  SPK_CHECK_ARG(E >= 0 && E < (1LL << 31) && (E == 0 || (idx_i && idx_j)), "spk_potential_compute: bad neighbour list (n_edges = %lld)", (long long)E);
  SPK_CHECK_ARG(energy && forces, "spk_potential_compute: null output");
  bool sorted = true;
  for (int64_t e = 0; e < E; ++e) {
    if (idx_i[e] < 0 || idx_i[e] >= N || idx_j[e] < 0 || idx_j[e] >= N) {
      spk_set_error("spk_potential_compute: neighbour index out of range [0, %lld) at edge %lld", (long long)N, (long long)e);
      return SPK_ERR_INDEX;
    }
    if (e > 0 && idx_i[e] < idx_i[e - 1]) sorted = false;
  }
  if (!sorted) {
    // stable counting sort by the centre atom: the order the reference's own lists have (neighborlist.py:450-453) and
    // the one the segmented kernels want; the result does not depend on the edge order (sums over neighbours)
    p->cnt.assign((size_t)N + 1, 0);
    for (int64_t e = 0; e < E; ++e) p->cnt[(size_t)idx_i[e] + 1]++;
    for (int64_t a = 0; a < N; ++a) p->cnt[(size_t)a + 1] += p->cnt[(size_t)a];
    p->si.resize((size_t)E); p->sj.resize((size_t)E);
    if (offsets) p->soff.resize((size_t)E * 3);
    for (int64_t e = 0; e < E; ++e) {
      const int64_t d = p->cnt[(size_t)idx_i[e]]++;
      p->si[(size_t)d] = idx_i[e]; p->sj[(size_t)d] = idx_j[e];
      if (offsets) { p->soff[3 * d] = offsets[3 * e]; p->soff[3 * d + 1] = offsets[3 * e + 1]; p->soff[3 * d + 2] = offsets[3 * e + 2]; }
    }
    idx_i = p->si.data(); idx_j = p->sj.data();
    if (offsets) offsets = p->soff.data();
  }
  if ((rc = upload_atoms(p, N, z, R, idx_m))) return rc;
  hipStream_t s = p->stream;
  const size_t Eb = (size_t)std::max<int64_t>(E, 1);
  if ((rc = p->ii.ensure(Eb * 8))) return rc;
  if ((rc = p->jj.ensure(Eb * 8))) return rc;
  if (offsets && (rc = p->off.ensure(Eb * 12))) return rc;
  if (E > 0) {
    SPK_HIP_TRY(hipMemcpyAsync(p->ii.p, idx_i, (size_t)E * 8, hipMemcpyHostToDevice, s));
    SPK_HIP_TRY(hipMemcpyAsync(p->jj.p, idx_j, (size_t)E * 8, hipMemcpyHostToDevice, s));
    if (offsets) SPK_HIP_TRY(hipMemcpyAsync(p->off.p, offsets, (size_t)E * 12, hipMemcpyHostToDevice, s));
  }
  p->cell_list = false;
  return run_model(p, N, E, M, true, false, offsets != nullptr, z, idx_m, energy, forces);
}

extern "C" int spk_potential_compute_cell(spk_potential_t* p, int64_t N, const int64_t* z, const float* R, int64_t M, const int64_t* idx_m,
                                          const float* cell, const uint8_t* pbc, float skin, float* energy, float* forces,
                                          int64_t* stats) {
  int rc = check_atoms(p, N, z, R, M, idx_m, "spk_potential_compute_cell");
  if (rc) return rc;
  SPK_CHECK_ARG(energy && forces, "spk_potential_compute_cell: null output");
  SPK_CHECK_ARG(skin >= 0.f && std::isfinite(skin), "spk_potential_compute_cell: skin must be >= 0");
  bool periodic = false;
  if (pbc) for (int64_t k = 0; k < 3 * M; ++k) periodic = periodic || pbc[k];
  SPK_CHECK_ARG(!periodic || cell, "spk_potential_compute_cell: periodic directions need a cell");
  // keep the list?  (md/neighborlist_md.py:80-90: rebuild when an atom moved more than half the skin)
  bool rebuild = !p->cell_list || p->list_atoms != N || p->list_mol != M || p->list_skin != skin || skin == 0.f;
  if (!rebuild) {
    if (idx_m ? (p->idxm_ref.size() != (size_t)N || memcmp(p->idxm_ref.data(), idx_m, (size_t)N * 8) != 0) : !p->idxm_ref.empty()) rebuild = true;
    if (cell ? (p->cell_ref.size() != (size_t)M * 9 || memcmp(p->cell_ref.data(), cell, (size_t)M * 36) != 0) : !p->cell_ref.empty()) rebuild = true;
    if (pbc ? (p->pbc_ref.size() != (size_t)M * 3 || memcmp(p->pbc_ref.data(), pbc, (size_t)M * 3) != 0) : !p->pbc_ref.empty()) rebuild = true;
  }
  if (!rebuild) {
    const float lim = 0.25f * skin * skin;
    for (int64_t a = 0; a < N && !rebuild; ++a) {
      const float dx = R[3 * a] - p->R_ref[3 * a], dy = R[3 * a + 1] - p->R_ref[3 * a + 1], dz = R[3 * a + 2] - p->R_ref[3 * a + 2];
      if (!(dx * dx + dy * dy + dz * dz <= lim)) rebuild = true;
    }
  }
  if ((rc = upload_atoms(p, N, z, R, idx_m))) return rc;
  hipStream_t s = p->stream;
  int64_t E = p->g.n_edges;
  if (rebuild) {
    p->cell_list = false;
    if (cell) {
      if ((rc = p->cell.ensure((size_t)M * 36))) return rc;
      SPK_HIP_TRY(hipMemcpyAsync(p->cell.p, cell, (size_t)M * 36, hipMemcpyHostToDevice, s));
    }
    if (pbc) {
      if ((rc = p->pbc.ensure((size_t)M * 3))) return rc;
      SPK_HIP_TRY(hipMemcpyAsync(p->pbc.p, pbc, (size_t)M * 3, hipMemcpyHostToDevice, s));
    }
    if ((rc = p->nblws.ensure((size_t)spk_nbl_workspace_bytes(N, M)))) return rc;
    if ((rc = p->nblrow.ensure((size_t)(N + 1) * 4))) return rc;
    const float rc_list = p->h.cutoff + skin;
    const int64_t* dm = idx_m ? p->idxm.as<int64_t>() : nullptr;
    E = 0;
    rc = spk_nbl_count_f32(p->R.as<float>(), dm, cell ? p->cell.as<float>() : nullptr, pbc ? p->pbc.as<uint8_t>() : nullptr, N, M, rc_list,
                           p->nblws.p, p->nblrow.as<int32_t>(), &E, s);
    if (rc) return rc;
    const size_t Eb = (size_t)std::max<int64_t>(E, 1);
    if ((rc = p->ii.ensure(Eb * 8))) return rc;
    if ((rc = p->jj.ensure(Eb * 8))) return rc;
    if ((rc = p->off.ensure(Eb * 12))) return rc;
    if (E > 0) {
      rc = spk_nbl_fill_f32(p->R.as<float>(), dm, N, M, rc_list, p->nblws.p, p->nblrow.as<int32_t>(), E, p->ii.as<int64_t>(), p->jj.as<int64_t>(),
                            nullptr, p->off.as<float>(), s);
      if (rc) return rc;
    }
    p->R_ref.assign(R, R + 3 * N);
    if (idx_m) p->idxm_ref.assign(idx_m, idx_m + N); else p->idxm_ref.clear();
    if (cell) p->cell_ref.assign(cell, cell + 9 * M); else p->cell_ref.clear();
    if (pbc) p->pbc_ref.assign(pbc, pbc + 3 * M); else p->pbc_ref.clear();
  }
  rc = run_model(p, N, E, M, rebuild, skin > 0.f, true, z, idx_m, energy, forces);
  if (rc) { p->cell_list = false; return rc; }
  p->cell_list = true; p->list_atoms = N; p->list_mol = M; p->list_skin = skin;
  if (stats) { stats[0] = E; stats[1] = rebuild ? 1 : 0; }
  return SPK_OK;
}
