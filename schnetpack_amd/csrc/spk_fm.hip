// Force-matching gradient engine on the device: the HIP instantiation of spk_fm_engine.h (host orchestration of the four passes)
// and spk_fm_kernels.h (their element-wise / row kernels), with
//   Dense / input-gradient GEMMs    spk_dense_f32 / spk_dense_bwd_input_f32 (fp32 MFMA, spk_dense.hip) on [value ; tangent]-stacked rows
//   weight-gradient GEMMs           spk_gemm_tn_nb_f32 (v_mfma_f32_32x32x2_f32 over the stacked sample dimension, spk_gemm_tn.h)
//   by-neighbour CSR                 stable rocPRIM radix sort of (idx_j, pair id)
// Replaces the second-order autograd graph of the reference's training step (atomistic/response.py:59-68, task.py:166-185).
#include <string.h>
#include <vector>
#include "spk_common.h"
#include <rocprim/rocprim.hpp>
#define SPK_FM_PAINN_UNCOND 1
#include "spk_fm_engine.h"

// ------------------------------------------------------------------------------------------------ by-neighbour CSR
__global__ void k_tp_keys(const int64_t* __restrict__ jj, int64_t E, int64_t N, int* __restrict__ keys, int* __restrict__ vals) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = jj[e];
    keys[e] = (uint64_t)j < (uint64_t)N ? (int)j : (int)N;
    vals[e] = (int)e;
  }
}
// colptr [rows + 1] of ascending int keys in [0, rows)
__global__ void k_tp_colptr(const int* __restrict__ keys, int64_t E, int64_t rows, int* __restrict__ colptr) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e <= E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t prev = e > 0 ? keys[e - 1] : -1;
    const int64_t cur = e < E ? keys[e] : rows;
    for (int64_t r = prev + 1; r <= cur; ++r) colptr[r] = (int)e;
  }
}
static size_t tp_align(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t tp_sort_tmp_bytes(int64_t n) {
  static size_t cached[64] = {0};
  int b = 0;
  while (((int64_t)1 << b) < n && b < 62) ++b;
  if (cached[b] == 0) {
    size_t bytes = 0;
    int* k = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, (size_t)1 << b, 0, 32, (hipStream_t)0);
    if (e != hipSuccess || bytes == 0) bytes = ((size_t)32 << b) + (1 << 20);     // no device to ask (build box): a generous bound
    cached[b] = bytes;
  }
  return cached[b];
}
extern "C" int64_t spk_transpose_plan_bytes(int64_t E, int64_t N) {
  (void)N;
  if (E < 1) E = 1;
  return (int64_t)(3 * tp_align((size_t)E * 4) + tp_align(tp_sort_tmp_bytes(E)));
}
extern "C" int spk_transpose_plan(const int64_t* idx_j, int64_t E, int64_t N, int32_t* colptr, int32_t* perm, void* tmp, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(E >= 0 && N >= 0 && E < (1ll << 31) && N < (1ll << 31) - 1 && colptr && (E == 0 || (idx_j && perm && tmp)), "spk_transpose_plan: bad arguments");
  SpkProfScope prof("transpose_plan", stream);
  char* p = (char*)tmp;
  int* keys = (int*)p;
  int* keys_s = (int*)(p + tp_align((size_t)(E > 0 ? E : 1) * 4));
  int* vals = (int*)(p + 2 * tp_align((size_t)(E > 0 ? E : 1) * 4));
  void* sort_tmp = p + 3 * tp_align((size_t)(E > 0 ? E : 1) * 4);
  if (E > 0) {
    hipLaunchKernelGGL(k_tp_keys, dim3(spk_grid_for(E, 256, 4096)), dim3(256), 0, stream, idx_j, E, N, keys, vals);
    size_t bytes = tp_sort_tmp_bytes(E);
    int bits = 1;
    while (((int64_t)1 << bits) <= N && bits < 31) ++bits;
    SPK_HIP_TRY(rocprim::radix_sort_pairs(sort_tmp, bytes, keys, keys_s, vals, perm, (size_t)E, 0, bits, stream));
  }
  hipLaunchKernelGGL(k_tp_colptr, dim3(spk_grid_for(E + 1, 256, 4096)), dim3(256), 0, stream, keys_s, E, N + 1, colptr);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

__global__ void k_tp_fill(const int64_t* __restrict__ idx_i, const int64_t* __restrict__ idx_j, const int* __restrict__ perm, int64_t E,
                          int64_t* __restrict__ ti, int64_t* __restrict__ tj) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < E; k += (int64_t)gridDim.x * blockDim.x) {
    const int e = perm[k];
    ti[k] = idx_j[e];
    tj[k] = idx_i[e];
  }
}
extern "C" int spk_transposed_build(const int64_t* idx_i, const int64_t* idx_j, int64_t E, int64_t N, int64_t* t_idx_i, int64_t* t_idx_j, int32_t* rowptr,
                                    int32_t* perm, void* tmp, void* stream_) {
  SPK_CHECK_ARG(E == 0 || (idx_i && t_idx_i && t_idx_j), "spk_transposed_build: null argument");
  int rc = spk_transpose_plan(idx_j, E, N, rowptr, perm, tmp, stream_);
  if (rc || E == 0) return rc;
  hipLaunchKernelGGL(k_tp_fill, dim3(spk_grid_for(E, 256, 4096)), dim3(256), 0, (hipStream_t)stream_, idx_i, idx_j, perm, E, t_idx_i, t_idx_j);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------ batched weight-gradient GEMMs
// Every weight gradient of a pass is G = U^T X over [value ; tangent]-stacked rows (spk_gemm_tn.h).  The ~20 problems of a step are
// independent of each other and of the rest of pass D, and each is tiny (a few hundred rows): as separate launches they cost ~19 us
// apiece (latency), together they fill the chip once.  Descriptors travel as kernel arguments (< 4 KB), so nothing is staged in memory.
#include "spk_gemm_tn.h"
#include "spk_fm_chain.h"
#define FM_TN_MAX 24
struct GemmTnBatch {
  int n;
  int xcd_walk;                // 1: XCD-contiguous (slice, tile) order inside a problem (SPK_XCD_WALK=0 switches it off)
  int prefix[FM_TN_MAX + 1];   // first workgroup of every problem
  GemmTnArgs a[FM_TN_MAX];
};
__global__ __launch_bounds__(64 * TN_WAVES) void k_gemm_tn_batched(GemmTnBatch b_) {
  // The problem index is dynamic: indexing the by-value argument would make the compiler copy all of it (2.8 KB) into scratch per lane
  // (first version: 227 us for what is 15 us of work).  Read the descriptor of this workgroup's problem from the kernel-argument segment
  // itself -- constant memory, uniform address, scalar loads.
  typedef const __attribute__((address_space(4))) GemmTnBatch* BatchPtr;      // keep the constant address space: uniform index => s_load
  BatchPtr b = (BatchPtr)__builtin_amdgcn_kernarg_segment_ptr();
  (void)b_;
  int p = 0;
  const int n = b->n;
  while (p + 1 < n && (int)blockIdx.x >= b->prefix[p + 1]) ++p;
  const int local = (int)blockIdx.x - b->prefix[p];
  GemmTnArgs a;
  a.U = b->a[p].U; a.X = b->a[p].X; a.n = b->a[p].n; a.O = b->a[p].O; a.K = b->a[p].K; a.tiles_k = b->a[p].tiles_k; a.S = b->a[p].S; a.n_tiles = b->a[p].n_tiles;
  a.rows_per_slice = b->a[p].rows_per_slice; a.rows_per_wave = b->a[p].rows_per_wave; a.nb = b->a[p].nb;
  a.G = b->a[p].G; a.gb = b->a[p].gb; a.ws = b->a[p].ws; a.wsb = b->a[p].wsb; a.tickets = b->a[p].tickets; a.defer = b->a[p].defer;
  // Which (tile, slice) this workgroup takes.  Workgroups go to the XCDs round robin (blockIdx % 8), and every tile of one slice reads the
  // same rows of U / X (1 024 rows x (O + K) floats: ~1 MB for 128 x 128) -- in plain order the 16 tiles of a slice sit on 8 different L2s
  // and the operands (2 x 42 MB at 128 frames: far beyond a 4 MB L2) are fetched from the memory side four times over.  With the pairs
  // enumerated so that one XCD gets a CONTIGUOUS range of (slice, tile) pairs, all tiles of a slice share one L2.
  int pair = local;
  const int total = a.n_tiles * a.S;
  if (a.S > 1 && total % 8 == 0 && b->xcd_walk) pair = (local & 7) * (total >> 3) + (local >> 3);
  gemm_tn_block(a, pair % a.n_tiles, pair / a.n_tiles);
}
// the sliced problems of a batch (S > 1, deferred): one workgroup per output tile adds the partial tiles up; prefix[] counts tiles here
__global__ __launch_bounds__(64 * TN_WAVES) void k_gemm_tn_batched_reduce(GemmTnBatch b_) {
  typedef const __attribute__((address_space(4))) GemmTnBatch* BatchPtr;
  BatchPtr b = (BatchPtr)__builtin_amdgcn_kernarg_segment_ptr();
  (void)b_;
  int p = 0;
  const int n = b->n;
  while (p + 1 < n && (int)blockIdx.x >= b->prefix[p + 1]) ++p;
  GemmTnArgs a;
  a.O = b->a[p].O; a.K = b->a[p].K; a.tiles_k = b->a[p].tiles_k; a.S = b->a[p].S; a.n_tiles = b->a[p].n_tiles;
  a.G = b->a[p].G; a.gb = b->a[p].gb; a.ws = b->a[p].ws; a.wsb = b->a[p].wsb;
  gemm_tn_reduce_block(a, (int)blockIdx.x - b->prefix[p]);
}

// ------------------------------------------------------------------------------------------------ device backend of the engine
// Side streams of the engine -- EXPERIMENT, off by default (SPK_FM_STREAMS=1).  Launches of a training step are latency bound (~100
// launches of 2-5 us of work each) and some chains are independent (the filter networks of the interactions depend on the geometry
// only), so they were put on side streams beside the atom chain; inside a stream capture the event record / wait pairs become graph
// edges, i.e. the captured step gets parallel branches.  MEASURED (scripts/gpu_fm_streams_ab.sh, one box, alternating runs, identical
// losses): SchNet 0.610 -> 0.851 ms per step, PaiNN 0.891 -> 1.062 ms -- every cross-queue dependency of a replayed HIP graph costs
// more than the short launch it lets overlap.  The switch stays for re-measurement on other runtimes; per device, created at the first
// EAGER call (a captured training step is preceded by eager warm-up steps).
struct FmSideStreams {
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t fork_ev[2] = {nullptr, nullptr};
  hipEvent_t done_ev[16] = {};
  bool ok = false;
};
static FmSideStreams* fm_side_streams(hipStream_t main) {
  static FmSideStreams per_dev[64];
  static const bool on = [] { const char* e = getenv("SPK_FM_STREAMS"); return e && e[0] == '1'; }();
  if (!on) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  FmSideStreams& s = per_dev[dev];
  if (!s.ok) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;     // never create inside a capture
    for (int k = 0; k < 2; ++k) {
      if (hipStreamCreateWithFlags(&s.side[k], hipStreamNonBlocking) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&s.fork_ev[k], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    for (int k = 0; k < 16; ++k)
      if (hipEventCreateWithFlags(&s.done_ev[k], hipEventDisableTiming) != hipSuccess) return nullptr;
    s.ok = true;
  }
  return &s;
}

// row chains (experiment): 1 = record them, 0 / -1 = launch by launch (the default); SPK_FM_CHAIN sets the initial value
static int g_fm_chain_mode = [] { const char* e = getenv("SPK_FM_CHAIN"); return e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1)) : -1; }();
extern "C" void spk_fm_set_chain(int32_t mode) { g_fm_chain_mode = mode < 0 ? -1 : (mode ? 1 : 0); }

struct FmDeviceBackend {
  hipStream_t stream;
  hipStream_t main_stream;
  FmSideStreams* ss = nullptr;
  float* gws = nullptr;
  uint32_t* tickets = nullptr;
  int max_blocks;
  GemmTnBatch batch, rbatch;
  int64_t ws_used = 0;
  int tickets_used = 0;
  explicit FmDeviceBackend(hipStream_t s) : stream(s), main_stream(s), max_blocks(spk_num_cus() * 32) { batch.n = 0; batch.prefix[0] = 0; chain.n_stages = 0; }
  // ---- row chains (spk_fm_chain.h): between chain_begin() and chain_end() the atom-local launches are RECORDED as stages and leave as one
  // launch; anything else that launches flushes the recorded stages first, so the issue order of the engine is kept.  SPK_FM_CHAIN=0 / 1
  // switches the recording off / on for every size (default: batches of at most FM_CHAIN_MAX_ATOMS atoms).
  FmChainDesc chain;
  bool recording = false;
  int chain_max_k = 0, chain_max_nw = 0, chain_max_r = 0;
  static int chain_mode() { return g_fm_chain_mode; }
  void chain_begin(int64_t N) {
    const int m = chain_mode();
    // EXPERIMENT, default OFF (measured slower: profiles/r05_row_chains.md): recorded only on request (SPK_FM_CHAIN=1 / spk_fm_set_chain(1))
    recording = m == 1 && N > 0;
    chain.n_stages = 0; chain.N = N; chain_max_k = chain_max_nw = chain_max_r = 0;
  }
  int chain_end() { const int rc = chain_flush(); recording = false; return rc; }
  int chain_flush() {
    if (chain.n_stages == 0) return SPK_OK;
    const int n = chain.n_stages;
    chain.n_stages = 0;                       // (first: the launch below must not re-enter through pre_launch)
    FmChainDesc d = chain;
    d.n_stages = n;
    // per stage: the split of K over the waves, and whether Y is handed to the next stage through LDS (Dense -> Dense on the same rows)
    const int nwaves = FM_CHAIN_THREADS / 64;
    int buf = 0;
    for (int i = 0; i < n; ++i) {
      if (d.st[i].is_ew) continue;
      FmGemmStage& g = d.st[i].g;
      const int n_cg = (g.NW + 63) / 64;
      int ks = 1;
      while (n_cg * ks * 2 <= nwaves && (g.K % (ks * 2 * FM_CHAIN_CHUNK)) == 0) ks *= 2;
      g.ks = ks;
      g.keep_y = 0;
      g.x_from_lds = 0;
      const int ns_in = g.mode == FM_G_TANGENT ? 1 : g.ns;
      const int R = ns_in * g.rpa * FM_CHAIN_ATOMS;
      const int need_x = R * (g.K + 4), need_p = ks * R * (g.NW + 4);
      if (need_x > buf) buf = need_x;
      if (need_p > buf) buf = need_p;
    }
    for (int i = 0; i + 1 < n; ++i) {
      if (d.st[i].is_ew || d.st[i + 1].is_ew) continue;
      FmGemmStage &a = d.st[i].g, &b = d.st[i + 1].g;
      const int a_ns_out = a.mode == FM_G_TANGENT ? 1 : a.ns, b_ns_in = b.mode == FM_G_TANGENT ? 1 : b.ns;
      const bool prologue = b.mode == FM_G_BWD_INPUT && b.pre_in != nullptr;
      if (b.X == a.Y && b.K == a.NW && a.rpa == b.rpa && a_ns_out == b_ns_in && !prologue) { a.keep_y = 1; b.x_from_lds = 1; }
    }
    // operands that an earlier stage of this chain writes (any overlap of the address ranges is enough to call it internal)
    for (int i = 0; i < n; ++i) {
      if (d.st[i].is_ew) continue;
      FmGemmStage& g = d.st[i].g;
      auto internal = [&](const float* p) {
        if (!p) return true;
        for (int j = 0; j < i; ++j) {
          if (d.st[j].is_ew) { for (int q = 0; q < 3; ++q) if (d.st[j].e.out[q] && chain_near(d.st[j].e.out[q], p)) return true; }
          else if (chain_near(d.st[j].g.Y, p) || (d.st[j].g.pre_out && chain_near(d.st[j].g.pre_out, p))) return true;
        }
        return false;
      };
      g.ext = (internal(g.X) ? 0 : 1) | (internal(g.res) ? 0 : 2) | (internal(g.pre_in) ? 0 : 4);
    }
    d.buf_floats = (buf + 3) & ~3;
    static const bool dbg = getenv("SPK_FM_CHAIN_DEBUG") != nullptr;
    if (dbg) {
      fprintf(stderr, "[fm_chain] N=%lld stages=%d lds=%d B:", (long long)chain.N, n, 2 * d.buf_floats * 4);
      for (int i = 0; i < n; ++i) {
        if (d.st[i].is_ew) fprintf(stderr, " ew%d", d.st[i].e.kind);
        else fprintf(stderr, " g(m%d %dx%d R%d ks%d%s%s%s)", d.st[i].g.mode, d.st[i].g.K, d.st[i].g.NW, (d.st[i].g.mode == FM_G_TANGENT ? 1 : d.st[i].g.ns) * d.st[i].g.rpa * FM_CHAIN_ATOMS,
                     d.st[i].g.ks, d.st[i].g.trans ? " T" : "", d.st[i].g.x_from_lds ? " <lds" : "", d.st[i].g.keep_y ? " >lds" : "");
      }
      fprintf(stderr, "\n");
    }
    const size_t lds = (size_t)2 * d.buf_floats * sizeof(float);
    static SpkPerDevice lds_set;
    int dev_;
    if (lds_set.pending(&dev_)) {
      SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_fm_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096));      // (the kernel keeps its descriptor, < 4 KB, in static LDS)
      lds_set.mark(dev_);
    }
    SPK_CHECK_ARG(lds <= 160 * 1024 - 4096, "fm engine: row chain needs %zu bytes of LDS", lds);
    const int64_t blocks = (chain.N + FM_CHAIN_ATOMS - 1) / FM_CHAIN_ATOMS;
    static const bool stamps = getenv("SPK_FM_CHAIN_STAMPS") != nullptr;       // debugging aid: cycle stamps of workgroup 0, printed per launch (synchronises!)
    static unsigned long long* dbg_dev = nullptr;
    d.dbg = nullptr;
    // timing experiment only, results are WRONG: bit 0 no epilogue loads / stores, 1 no weight loads, 2 no X loads, 3 no element-wise stages, 4 no warm-up
    static const int dry = [] { const char* e = getenv("SPK_FM_CHAIN_DRY"); return e ? atoi(e) : 0; }();
    d.dry = dry;
    if (stamps) {
      if (!dbg_dev) SPK_HIP_TRY(hipMalloc((void**)&dbg_dev, 128 * sizeof(unsigned long long)));
      SPK_HIP_TRY(hipMemsetAsync(dbg_dev, 0, 128 * sizeof(unsigned long long), stream));
      d.dbg = dbg_dev;
    }
    {
      SpkProfScope prof("fm_chain", stream);
      hipLaunchKernelGGL(k_fm_chain, dim3((unsigned)blocks), dim3(FM_CHAIN_THREADS), lds, stream, d);
    }
    if (stamps) {
      unsigned long long h[128];
      SPK_HIP_TRY(hipStreamSynchronize(stream));
      SPK_HIP_TRY(hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[fm_chain stamps] warm %llu |", h[1] - h[0]);
      unsigned long long prev = h[1];
      for (int i = 0; i < n; ++i) {
        const unsigned long long* q = h + 2 + 8 * i;
        if (d.st[i].is_ew) { fprintf(stderr, " ew:%llu |", q[6] - prev); prev = q[6]; }
        else {
          fprintf(stderr, " g[x %llu bar %llu mm %llu bar %llu ld %llu st %llu bar %llu] |", q[0] - prev, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5]);
          prev = q[6];
        }
      }
      fprintf(stderr, " total %llu\n", prev - h[0]);
    }
    chain_max_k = chain_max_nw = chain_max_r = 0;
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  // p lies inside (or right behind: stacked halves are addressed as base + offset) the tensor that starts at `base`: tensors of a workspace are at
  // most 12 N F floats long
  bool chain_near(const float* base, const float* p) const {
    const int64_t span = 12ll * chain.N * FM_CHAIN_MAX_W;
    return p >= base && p < base + span;
  }
  void pre_launch() { if (recording && chain.n_stages) (void)chain_flush(); }
  // a Dense-type stage joins the chain when the shapes fit (rows = 1, 2, 3 or 6 per atom; widths multiples of 4 up to FM_CHAIN_MAX_W; 16-byte aligned rows)
  bool chain_gemm(int mode, const float* X, const float* W, const float* b, const float* res, const float* pre_in, float* Y, float* pre_out, int64_t m_rows, int K, int NW,
                  int act, int trans, int ns_hint) {
    if (!recording || chain.N <= 0 || m_rows % chain.N) return false;
    const int64_t mpa = m_rows / chain.N;                 // rows per atom over all stacks of X
    int ns, rpa;
    if (ns_hint == 2) { if (mpa != 2 && mpa != 6) return false; ns = 2; rpa = (int)mpa / 2; }
    else if (mpa == 1 || mpa == 3) { ns = 1; rpa = (int)mpa; }
    else if (mpa == 2 || mpa == 6) { ns = 2; rpa = (int)mpa / 2; }
    else return false;
    if (mode == FM_G_TANGENT && ns != 1) return false;
    if (K < FM_CHAIN_CHUNK || NW < 4 || (K % FM_CHAIN_CHUNK) || (NW & 3) || K > FM_CHAIN_MAX_W || NW > FM_CHAIN_MC * FM_CHAIN_CW) return false;
    if (!al16(X) || !al16(W) || !al16(Y) || !al16(pre_in) || !al16(pre_out) || !al16(res)) return false;
    if (chain.n_stages == FM_CHAIN_MAX_STAGES && chain_flush()) return false;
    FmChainStage& st = chain.st[chain.n_stages++];
    st.is_ew = 0;
    st.g.X = X; st.g.W = W; st.g.b = b; st.g.res = res; st.g.pre_in = pre_in; st.g.Y = Y; st.g.pre_out = pre_out;
    st.g.K = K; st.g.NW = NW; st.g.act = act; st.g.mode = mode; st.g.trans = trans; st.g.ns = ns; st.g.rpa = rpa;
    const int R = ns * rpa * FM_CHAIN_ATOMS;
    if (K > chain_max_k) chain_max_k = K;
    if (NW > chain_max_nw) chain_max_nw = NW;
    if (R > chain_max_r) chain_max_r = R;
    return true;
  }
  // the atom-local element-wise kernels of the PaiNN mixing block (FmEwArgs): a stage of the chain, or one generic launch
  void ew(const FmEwArgs<float>& a) {
    if (recording && a.N == chain.N) {
      if (chain.n_stages == FM_CHAIN_MAX_STAGES) (void)chain_flush();
      FmChainStage& st = chain.st[chain.n_stages++];
      st.is_ew = 1;
      st.e = a;
      return;
    }
    static const char* const tags[FM_EW_KINDS] = {"fm_painn_mix", "fm_painn_mix_t", "fm_painn_update", "fm_painn_update_t", "fm_painn_update_bwd", "fm_painn_mix_bwd",
                                                  "fm_painn_update_dual_bwd", "fm_painn_mix_dual_bwd"};
    flat(tags[a.kind], k_fm_ew<float>, a.N * a.F, a);
  }
  // fork(s): later launches go to side stream s (which first waits for everything issued on the main stream so far); back(k): record
  // "done" event k there and return to the main stream; wait(k): the main stream waits for event k.  Without side streams all three are
  // no-ops and the work stays in issue order on the main stream.
  bool can_fork(int n_events) { if (!ss) ss = fm_side_streams(main_stream); return ss != nullptr && n_events <= 16; }
  void fork(int s) {
    if (!ss) return;
    pre_launch();
    (void)hipEventRecord(ss->fork_ev[s], main_stream);
    (void)hipStreamWaitEvent(ss->side[s], ss->fork_ev[s], 0);
    stream = ss->side[s];
  }
  void back(int k) {
    if (!ss) return;
    pre_launch();
    (void)hipEventRecord(ss->done_ev[k], stream);
    stream = main_stream;
    pending |= 1u << k;
  }
  void wait(int k) {
    if (!ss || !(pending & (1u << k))) return;
    pre_launch();
    (void)hipStreamWaitEvent(main_stream, ss->done_ev[k], 0);
    pending &= ~(1u << k);
  }
  void set_gemm_ws(float* w, uint32_t* t) { gws = w; tickets = t; }
  unsigned pending = 0;      // "done" events recorded on a side stream and not yet waited for by the main stream
  int gemm_flush() {
    pre_launch();
    if (batch.n == 0) return SPK_OK;
    for (int k = 0; k < 16; ++k)       // the batch reads operands that side streams may still be producing
      if (pending & (1u << k)) wait(k);
    batch.xcd_walk = spk_xcd_walk_default();
    {
      SpkProfScope prof("gemm_tn_batched", stream);
      hipLaunchKernelGGL(k_gemm_tn_batched, dim3(batch.prefix[batch.n]), dim3(64 * TN_WAVES), 0, stream, batch);
    }
    rbatch.n = 0;
    rbatch.prefix[0] = 0;
    for (int p = 0; p < batch.n; ++p)
      if (batch.a[p].S > 1) {
        rbatch.a[rbatch.n] = batch.a[p];
        rbatch.prefix[rbatch.n + 1] = rbatch.prefix[rbatch.n] + batch.a[p].n_tiles;
        ++rbatch.n;
      }
    if (rbatch.n > 0) {
      SpkProfScope prof("gemm_tn_batched_reduce", stream);
      hipLaunchKernelGGL(k_gemm_tn_batched_reduce, dim3(rbatch.prefix[rbatch.n]), dim3(64 * TN_WAVES), 0, stream, rbatch);
    }
    batch.n = 0;
    ws_used = 0;
    tickets_used = 0;
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  // A workgroup walks its rows in batches of 32 per wave, one exposed memory round trip after the other: the pair-row problems of a step
  // (5 120 rows, 20 batches per wave) took 45 us while the atom-row problems of the same launch were done after 5.  Rows are therefore cut
  // into slices of ~1 024; the partial tiles of the slices are added up by a SECOND launch (k_gemm_tn_batched_reduce), not inside the first:
  // meeting through memory behind __threadfence() costs an L2 write-back per workgroup on this part (first batched version: 167 us for
  // 15 us of work), a kernel boundary costs ~4 us once.
  static void tn_plan(int64_t n, int O, int K, int32_t* S, int64_t* wsf, int32_t* tiles) {
    *tiles = ((O + 31) / 32) * ((K + 31) / 32);
    int64_t s = (n + 512) / 1024;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    *S = (int32_t)s;
    *wsf = s > 1 ? s * (int64_t)*tiles * (1024 + 32) : 0;
  }
  int64_t gemm_tn_ws_floats(int64_t n, int O, int K) {
    int32_t S, tiles;
    int64_t wsf = 0;
    tn_plan(n, O, K, &S, &wsf, &tiles);
    return wsf;
  }
  size_t transpose_tmp_bytes(int64_t E, int64_t N) { return (size_t)spk_transpose_plan_bytes(E, N); }
  int zero_u32(uint32_t* p, int64_t n) { pre_launch(); return spk_zero_async(p, (size_t)n * 4, stream); }
  int rowptr(const int64_t* idx, int64_t n, int64_t rows, int32_t* out, int32_t* err) { pre_launch(); return spk_segment_rowptr_i32(idx, n, rows, out, err, stream); }
  int rowptr2(const int64_t* ia, int64_t na, int64_t ra, int32_t* oa, const int64_t* ib, int64_t nb, int64_t rb, int32_t* ob, int32_t* err) {
    pre_launch();
    const spk_index_job_t jobs[2] = {{ia, na, ra, oa}, {ib, nb, rb, ob}};
    return spk_index_jobs(jobs, 2, err, stream);
  }
  int transpose_plan(const int64_t* jj, int64_t E, int64_t N, int32_t* colptr, int32_t* perm, void* tmp) { pre_launch(); return spk_transpose_plan(jj, E, N, colptr, perm, tmp, stream); }
  int dense(const float* x, const float* w, const float* b, const float* res, float* y, float* pre, int64_t m, int k, int n_out, int act) {
    if (chain_gemm(FM_G_DENSE, x, w, b, res, nullptr, y, pre, m, k, n_out, act, 0, 0)) return SPK_OK;
    pre_launch();
    return spk_dense_f32(x, w, b, res, y, pre, m, k, n_out, act, stream);
  }
  int dense_bwd_input(const float* dy, const float* pre, const float* w, const float* res, float* dx, int64_t m, int k, int n_out, int act) {
    if ((pre || act == FM_ACT_NONE) && chain_gemm(FM_G_BWD_INPUT, dy, w, nullptr, res, pre, dx, nullptr, m, n_out, k, act, 1, 0)) return SPK_OK;
    pre_launch();
    return spk_dense_bwd_input_f32(dy, pre, w, res, dx, m, k, n_out, act, stream);
  }
  // Dense layers on (value, tangent) pairs: ONE launch each (spk_dense_dual_f32) at training sizes; problems beyond the pair kernel's tile
  // budget (and SPK_FM_NO_DUAL=1, the A/B switch) run as the two Dense launches and the element-wise launch they replace.
  // MEASURED per use (scripts/gpu_fm_mask.sh, 8-frame aspirin step, SPK_FM_DUAL_MASK): the forward pair pays (SchNet 0.529 -> 0.507 ms:
  // three launches of which two are Dense become one); the tangent alone is neutral (a Dense launch absorbs a ~1.5 us element-wise one);
  // the reverse of the pair LOSES (SchNet +3 us, PaiNN +18 us per step: its epilogue evaluates act' and act'' behind the transposed-weight
  // loads, longer than the Dense launch plus the short element-wise launch it replaces) -- default mask 3, the reverse stays separate.
  static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
  static bool dual_enabled(int kind) {      // kind: 1 forward pair, 2 tangent alone, 4 reverse of the pair (SPK_FM_DUAL_MASK: tuning)
    static const int mask = [] {
      const char* e = getenv("SPK_FM_NO_DUAL");
      if (e && e[0] == '1') return 0;
      const char* m = getenv("SPK_FM_DUAL_MASK");
      return m ? atoi(m) : 3;
    }();
    return (mask & kind) != 0;
  }
  bool dual_ok(int64_t M, int KC, int NW, int kind = 1) const { return dual_enabled(kind) && spk_dense_dual_supported(M, KC, NW); }
  int dense_dual(const float* x2, const float* w, const float* b, float* y2, float* pre2, int64_t M, int k, int n_out, int act, const float* fc, const float* fc1) {
    const float* xt = x2 + M * k;
    float* yt = y2 + M * n_out;
    float* pt = pre2 ? pre2 + M * n_out : nullptr;
    if (!fc && chain_gemm(FM_G_DUAL_FWD, x2, w, b, nullptr, nullptr, y2, pre2, 2 * M, k, n_out, act, 0, 2)) return SPK_OK;
    pre_launch();
    if (dual_enabled(1) && spk_dense_dual_fwd_supported(M, k, n_out) && al16(xt) && al16(yt) && al16(pt)) {      // (one-tile-per-workgroup kernel up to 4 tiles per CU, grid-stride kernel beyond)
      spk_dense_dual_t d = {};
      d.x_v = x2; d.x_t = xt; d.w = w; d.b = b; d.fc = fc; d.fc1 = fc1; d.y_v = y2; d.y_t = yt; d.pre_v = pre2; d.pre_t = pt;
      d.m = M; d.k_in = k; d.n_out = n_out; d.act = act; d.mode = SPK_DD_FWD; d.trans = 0;
      return spk_dense_dual_f32(&d, stream);
    }
    int rc;
    if (fc) {
      if ((rc = dense(x2, w, b, nullptr, y2, nullptr, M, k, n_out, FM_ACT_NONE))) return rc;
      if ((rc = dense(xt, w, nullptr, nullptr, yt, nullptr, M, k, n_out, FM_ACT_NONE))) return rc;
      flat("fm_filter_fc", k_fm_filter_fc<float>, M * n_out, y2, fc, fc1, M, n_out);
      return SPK_OK;
    }
    SPK_CHECK_ARG(pre2 || act == FM_ACT_NONE, "fm engine: a Dense pair with an activation keeps its pre-activations");
    if ((rc = dense(x2, w, b, nullptr, y2, pre2, M, k, n_out, act))) return rc;
    if (act == FM_ACT_NONE) return dense(xt, w, nullptr, nullptr, yt, pt, M, k, n_out, FM_ACT_NONE);
    if ((rc = dense(xt, w, nullptr, nullptr, pt, nullptr, M, k, n_out, FM_ACT_NONE))) return rc;
    flat("fm_act_t", k_fm_act_tangent<float>, M * n_out, (const float*)pre2, (const float*)pt, M * n_out, act, yt);
    return SPK_OK;
  }
  int dense_tangent(const float* xt, const float* w, const float* pre_v, float* yt, float* pre_t, int64_t M, int KC, int NW, int act, bool trans) {
    if (chain_gemm(FM_G_TANGENT, xt, w, nullptr, nullptr, pre_v, yt, pre_t, M, KC, NW, act, trans ? 1 : 0, 1)) return SPK_OK;
    pre_launch();
    if (dual_ok(M, KC, NW, 2) && al16(xt) && al16(pre_v) && al16(yt) && al16(pre_t)) {
      spk_dense_dual_t d = {};
      d.x_t = xt; d.w = w; d.pre_v_in = pre_v; d.y_t = yt; d.pre_t = pre_t;
      d.m = M; d.k_in = KC; d.n_out = NW; d.act = act; d.mode = SPK_DD_TANGENT; d.trans = trans ? 1 : 0;
      return spk_dense_dual_f32(&d, stream);
    }
    int rc = trans ? dense_bwd_input(xt, nullptr, w, nullptr, pre_t, M, NW, KC, FM_ACT_NONE) : dense(xt, w, nullptr, nullptr, pre_t, nullptr, M, KC, NW, FM_ACT_NONE);
    if (rc) return rc;
    flat("fm_act_t", k_fm_act_tangent<float>, M * NW, pre_v, (const float*)pre_t, M * NW, act, yt);
    return SPK_OK;
  }
  int dense_dual_bwd(const float* g2, const float* w, const float* pre2, float* gx2, float* tmp2, int64_t M, int k, int n_out, int act) {
    if (chain_gemm(FM_G_DUAL_BWD, g2, w, nullptr, nullptr, pre2, gx2, nullptr, 2 * M, n_out, k, act, 1, 2)) return SPK_OK;
    pre_launch();
    if (dual_ok(M, n_out, k, 4) && al16(g2 + M * n_out) && al16(pre2 + M * k) && al16(gx2 + M * k)) {
      spk_dense_dual_t d = {};
      d.x_v = g2; d.x_t = g2 + M * n_out; d.w = w; d.pre_v_in = pre2; d.pre_t_in = pre2 + M * k; d.y_v = gx2; d.y_t = gx2 + M * k;
      d.m = M; d.k_in = n_out; d.n_out = k; d.act = act; d.mode = SPK_DD_DUAL_BWD; d.trans = 1;
      return spk_dense_dual_f32(&d, stream);
    }
    int rc = dense_bwd_input(g2, nullptr, w, nullptr, tmp2, 2 * M, k, n_out, FM_ACT_NONE);
    if (rc) return rc;
    flat("fm_act_dual_bwd", k_fm_act_dual_bwd<float>, M * k, (const float*)tmp2, pre2, M * k, act, gx2);
    return SPK_OK;
  }
  // deferred: the problem joins the batch that gemm_flush() launches (the engine keeps U and X intact until then)
  int gemm_tn(const float* U, const float* X, int64_t n, int O, int K, float* G, float* gb, int64_t n_bias) {
    int32_t S, tiles;
    int64_t wsf;
    int rc;
    tn_plan(n, O, K, &S, &wsf, &tiles);
    SPK_CHECK_ARG(tiles <= 4096 && U && X && G && n_bias <= n, "fm engine: bad weight-gradient problem");
    if (batch.n == FM_TN_MAX || tickets_used + tiles > 4096) {
      if ((rc = gemm_flush())) return rc;
    }
    GemmTnArgs a = spk_gemm_tn_args(U, X, n, O, K, S, tiles, G, gb, gws ? gws + ws_used : nullptr, tickets + tickets_used);
    a.nb = n_bias;
    a.defer = 1;
    batch.a[batch.n] = a;
    batch.prefix[batch.n + 1] = batch.prefix[batch.n] + tiles * S;
    ++batch.n;
    ws_used += wsf;
    tickets_used += tiles;
    return SPK_OK;
  }
  template <class... KA, class... A>
  void flat(const char* tag, void (*k)(KA...), int64_t total, A... a) {
    if (total <= 0) return;
    pre_launch();
    SpkProfScope prof(tag, stream);
    hipLaunchKernelGGL(k, dim3(spk_grid_for(total, 256, max_blocks)), dim3(256), 0, stream, static_cast<KA>(a)...);
  }
  template <class... KA, class... A>
  void slotted(const char* tag, void (*k)(KA...), int64_t total, A... a) {      // FM_FOR_SLOTTED kernels: 64 items per workgroup, four waves share each item's row
    if (total <= 0) return;
    pre_launch();
    SpkProfScope prof(tag, stream);
    hipLaunchKernelGGL(k, dim3(spk_grid_for(total, 64, max_blocks)), dim3(256), 0, stream, static_cast<KA>(a)...);
  }
  template <class... KA, class... A>
  void rows(const char* tag, void (*k)(KA...), int64_t n_rows, A... a) {      // one wavefront per row, four per workgroup
    if (n_rows <= 0) return;
    pre_launch();
    SpkProfScope prof(tag, stream);
    hipLaunchKernelGGL(k, dim3(spk_grid_for(n_rows, 4, max_blocks)), dim3(256), 0, stream, static_cast<KA>(a)...);
  }
};
typedef FmEngine<float, FmDeviceBackend> FmDev;

// ------------------------------------------------------------------------------------------------ C ABI -> engine descriptions
static int fm_common_check(const spk_head_t* head, const spk_radial_t* rb, const char* who) {
  SPK_CHECK_ARG(head && rb, "%s: null head / radial description", who);
  SPK_CHECK_ARG(head->w1 && head->w2 && head->n_hidden >= 1 && (head->act == SPK_ACT_SSP || head->act == SPK_ACT_SILU), "%s: bad head description", who);
  SPK_CHECK_ARG((rb->kind == SPK_RBF_GAUSSIAN || rb->kind == SPK_RBF_BESSEL) && rb->n_rbf >= 1 && rb->n_rbf <= 1024 && rb->p0 &&
                (rb->kind == SPK_RBF_BESSEL || rb->p1) && rb->cutoff > 0.f, "%s: bad radial description", who);
  return SPK_OK;
}
static int fm_batch_check(const spk_fm_batch_t* b, const char* who) {
  SPK_CHECK_ARG(b != nullptr, "%s: null batch", who);
  SPK_CHECK_ARG(b->n_atoms >= 1 && b->n_edges >= 0 && b->n_mol >= 1 && b->n_edges < (1ll << 31) && b->n_atoms < (1ll << 31) - 2, "%s: bad batch sizes", who);
  SPK_CHECK_ARG(b->Z && b->idx_m && b->R && b->embedding && b->n_types >= 1 && (b->n_edges == 0 || (b->idx_i && b->idx_j)), "%s: null batch pointer", who);
  return SPK_OK;
}
static FmHead<float> fm_head(const spk_head_t* h) { return FmHead<float>{h->w1, h->b1, h->w2, h->b2, h->n_hidden, h->act}; }
static FmRadial<float> fm_radial(const spk_radial_t* rb) { return FmRadial<float>{rb->kind, rb->n_rbf, rb->p0, rb->p1, rb->cutoff}; }
static FmBatch<float> fm_batch(const spk_fm_batch_t* b) {
  return FmBatch<float>{b->n_atoms, b->n_edges, b->n_mol, b->Z, b->idx_i, b->idx_j, b->idx_m, b->R, b->offsets, b->embedding, b->n_types};
}

static int fm_schnet_model(const spk_schnet_t* m, std::vector<FmSchnetLayer<float>>& lay, FmSchnetModel<float>& out, const char* who) {
  SPK_CHECK_ARG(m && m->n_atom_basis >= 1 && m->n_filters >= 1 && m->n_interactions >= 1 && m->layers, "%s: bad model description (at least one interaction)", who);
  lay.resize((size_t)m->n_interactions);
  for (int l = 0; l < m->n_interactions; ++l) {
    const spk_schnet_layer_t& P = m->layers[l];
    SPK_CHECK_ARG(P.in2f_w && P.fn_w1 && P.fn_b1 && P.fn_w2 && P.fn_b2 && P.f2out_w1 && P.f2out_b1 && P.f2out_w2 && P.f2out_b2, "%s: null weight in interaction %d", who, l);
    lay[(size_t)l] = FmSchnetLayer<float>{P.in2f_w, P.fn_w1, P.fn_b1, P.fn_w2, P.fn_b2, P.f2out_w1, P.f2out_b1, P.f2out_w2, P.f2out_b2};
  }
  out = FmSchnetModel<float>{m->n_atom_basis, m->n_filters, m->n_interactions, lay.data()};
  return SPK_OK;
}
static int fm_painn_model(const spk_painn_t* m, const spk_radial_t* rb, std::vector<FmPainnLayer<float>>& lay, FmPainnModel<float>& out, const char* who) {
  SPK_CHECK_ARG(m && m->n_atom_basis >= 1 && m->n_interactions >= 1 && m->layers, "%s: bad model description", who);
  const int L = m->n_interactions, F = m->n_atom_basis;
  lay.resize((size_t)L);
  const bool shared = L > 1 && m->layers[1].filt_w == m->layers[0].filt_w;
  for (int l = 0; l < L; ++l) {
    const spk_painn_layer_t& P = m->layers[l];
    SPK_CHECK_ARG(P.ctx_w1 && P.ctx_b1 && P.ctx_w2 && P.ctx_b2 && P.mix_w && P.ictx_w1 && P.ictx_b1 && P.ictx_w2 && P.ictx_b2 && P.filt_w && P.filt_b,
                  "%s: null weight in interaction %d", who, l);
    // the filter rows of all interactions must be ONE matrix (filter_net.weight, painn.py:179-189): slices in interaction order, or one shared slice
    const int64_t row0 = shared ? 0 : 3ll * F * l;
    SPK_CHECK_ARG(P.filt_w == m->layers[0].filt_w + row0 * rb->n_rbf && P.filt_b == m->layers[0].filt_b + row0, "%s: filter_net rows of interaction %d are not slice %d of one matrix",
                  who, l, l);
    lay[(size_t)l] = FmPainnLayer<float>{P.ctx_w1, P.ctx_b1, P.ctx_w2, P.ctx_b2, P.mix_w, P.ictx_w1, P.ictx_b1, P.ictx_w2, P.ictx_b2};
  }
  out = FmPainnModel<float>{F, L, shared ? 1 : 0, m->epsilon, lay.data(), m->layers[0].filt_w, m->layers[0].filt_b};
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------ SchNet
extern "C" int64_t spk_schnet_fm_workspace_bytes(const spk_schnet_t* m, const spk_head_t* head, const spk_radial_t* rb, int64_t N, int64_t E, int64_t M, int32_t n_types) {
  if (!m || !head || !rb || N < 1 || E < 0 || M < 1 || n_types < 1) return -1;
  FmDeviceBackend be(nullptr);
  FmDev eng(be);
  FmDev::SchnetWs w;
  FmSchnetModel<float> mm{m->n_atom_basis, m->n_filters, m->n_interactions, nullptr};
  eng.schnet_carve(nullptr, mm, rb->n_rbf, head->n_hidden, N, E, M, n_types, w);
  return (int64_t)w.bytes;
}
extern "C" int64_t spk_schnet_fm_grad_floats(const spk_schnet_t* m, const spk_head_t* head, const spk_radial_t* rb, int32_t n_types) {
  if (!m || !head || !rb) return -1;
  return fm_schnet_grad_floats(m->n_atom_basis, m->n_filters, m->n_interactions, rb->n_rbf, head->n_hidden, n_types);
}
extern "C" int spk_schnet_fm_forward_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_radial_t* rb, const spk_fm_batch_t* batch, void* workspace, float* E,
                                         float* F, int32_t* err, void* stream_) {
  int rc = fm_common_check(head, rb, "spk_schnet_fm_forward_f32");
  if (rc) return rc;
  if ((rc = fm_batch_check(batch, "spk_schnet_fm_forward_f32"))) return rc;
  SPK_CHECK_ARG(workspace && E, "spk_schnet_fm_forward_f32: null workspace / output");
  std::vector<FmSchnetLayer<float>> lay;
  FmSchnetModel<float> mm;
  if ((rc = fm_schnet_model(m, lay, mm, "spk_schnet_fm_forward_f32"))) return rc;
  FmDeviceBackend be((hipStream_t)stream_);
  FmDev eng(be);
  rc = eng.schnet_forward(mm, fm_head(head), fm_radial(rb), fm_batch(batch), workspace, E, F, err);
  if (rc) return rc;
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
extern "C" int spk_schnet_fm_backward_f32(const spk_schnet_t* m, const spk_head_t* head, const spk_radial_t* rb, const spk_fm_batch_t* batch, void* workspace,
                                          const float* gE, const float* gF, float* grads, void* stream_) {
  int rc = fm_common_check(head, rb, "spk_schnet_fm_backward_f32");
  if (rc) return rc;
  if ((rc = fm_batch_check(batch, "spk_schnet_fm_backward_f32"))) return rc;
  SPK_CHECK_ARG(workspace && gE && gF && grads, "spk_schnet_fm_backward_f32: null pointer");
  std::vector<FmSchnetLayer<float>> lay;
  FmSchnetModel<float> mm;
  if ((rc = fm_schnet_model(m, lay, mm, "spk_schnet_fm_backward_f32"))) return rc;
  FmDeviceBackend be((hipStream_t)stream_);
  FmDev eng(be);
  rc = eng.schnet_backward(mm, fm_head(head), fm_radial(rb), fm_batch(batch), workspace, gE, gF, grads);
  if (rc) return rc;
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ------------------------------------------------------------------------------------------------ PaiNN
extern "C" int64_t spk_painn_fm_workspace_bytes(const spk_painn_t* m, const spk_head_t* head, const spk_radial_t* rb, int64_t N, int64_t E, int64_t M, int32_t n_types) {
  if (!m || !head || !rb || N < 1 || E < 0 || M < 1 || n_types < 1 || m->n_interactions < 1 || !m->layers) return -1;
  FmDeviceBackend be(nullptr);
  FmDev eng(be);
  FmDev::PainnWs w;
  const bool shared = m->n_interactions > 1 && m->layers[1].filt_w == m->layers[0].filt_w;
  FmPainnModel<float> mm{m->n_atom_basis, m->n_interactions, shared ? 1 : 0, m->epsilon, nullptr, nullptr, nullptr};
  eng.painn_carve(nullptr, mm, rb->n_rbf, head->n_hidden, N, E, M, n_types, w);
  return (int64_t)w.bytes;
}
extern "C" int64_t spk_painn_fm_grad_floats(const spk_painn_t* m, const spk_head_t* head, const spk_radial_t* rb, int32_t n_types) {
  if (!m || !head || !rb || m->n_interactions < 1 || !m->layers) return -1;
  const bool shared = m->n_interactions > 1 && m->layers[1].filt_w == m->layers[0].filt_w;
  return fm_painn_grad_floats(m->n_atom_basis, m->n_interactions, rb->n_rbf, head->n_hidden, n_types, shared ? 1 : 0);
}
extern "C" int spk_painn_fm_forward_f32(const spk_painn_t* m, const spk_head_t* head, const spk_radial_t* rb, const spk_fm_batch_t* batch, void* workspace, float* E,
                                        float* F, int32_t* err, void* stream_) {
  int rc = fm_common_check(head, rb, "spk_painn_fm_forward_f32");
  if (rc) return rc;
  if ((rc = fm_batch_check(batch, "spk_painn_fm_forward_f32"))) return rc;
  SPK_CHECK_ARG(workspace && E, "spk_painn_fm_forward_f32: null workspace / output");
  std::vector<FmPainnLayer<float>> lay;
  FmPainnModel<float> mm;
  if ((rc = fm_painn_model(m, rb, lay, mm, "spk_painn_fm_forward_f32"))) return rc;
  FmDeviceBackend be((hipStream_t)stream_);
  FmDev eng(be);
  rc = eng.painn_forward(mm, fm_head(head), fm_radial(rb), fm_batch(batch), workspace, E, F, err);
  if (rc) return rc;
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
extern "C" int spk_painn_fm_backward_f32(const spk_painn_t* m, const spk_head_t* head, const spk_radial_t* rb, const spk_fm_batch_t* batch, void* workspace,
                                         const float* gE, const float* gF, float* grads, void* stream_) {
  int rc = fm_common_check(head, rb, "spk_painn_fm_backward_f32");
  if (rc) return rc;
  if ((rc = fm_batch_check(batch, "spk_painn_fm_backward_f32"))) return rc;
  SPK_CHECK_ARG(workspace && gE && gF && grads, "spk_painn_fm_backward_f32: null pointer");
  std::vector<FmPainnLayer<float>> lay;
  FmPainnModel<float> mm;
  if ((rc = fm_painn_model(m, rb, lay, mm, "spk_painn_fm_backward_f32"))) return rc;
  FmDeviceBackend be((hipStream_t)stream_);
  FmDev eng(be);
  rc = eng.painn_backward(mm, fm_head(head), fm_radial(rb), fm_batch(batch), workspace, gE, gF, grads);
  if (rc) return rc;
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
