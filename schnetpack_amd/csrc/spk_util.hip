// Library plumbing + the HBM-bound index kernels: neighbour-list plan, segmented scatter_add,
// gather, radial basis / cutoff expansion, embedding lookup.
#include "spk_common.h"
#include <string.h>

// ---------------------------------------------------------------- error / info
static thread_local char g_err[512] = "";
static int g_variant = SPK_VARIANT_AUTO;
static int g_num_cus = 0;

void spk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int spk_num_cus() {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_num_cus = prop.multiProcessorCount;
    else
      g_num_cus = 256;
  }
  return g_num_cus;
}

extern "C" int spk_version(void) { return 100; }
extern "C" const char* spk_last_error(void) { return g_err; }
extern "C" void spk_set_variant(int v) { g_variant = v; }
extern "C" int spk_get_variant(void) { return g_variant; }
static int g_split = -1;
extern "C" void spk_set_split(int on) { g_split = on ? 1 : 0; }
extern "C" int spk_get_split(void) {
  if (g_split < 0) { const char* e = getenv("SPK_SPLIT"); g_split = (e && e[0] == '0') ? 0 : 1; }
  return g_split;
}

extern "C" int spk_device_info(int32_t* host_info) {
  SPK_CHECK_ARG(host_info != nullptr, "spk_device_info: null output");
  int dev = 0;
  SPK_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  SPK_HIP_TRY(hipGetDeviceProperties(&prop, dev));
  host_info[0] = prop.multiProcessorCount;
  host_info[1] = prop.warpSize;
  host_info[2] = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
  int arch = 0;
  const char* p = strstr(prop.gcnArchName, "gfx");
  if (p) arch = (int)strtol(p + 3, nullptr, 16) == 0x950 ? 950 : (int)strtol(p + 3, nullptr, 10);
  host_info[3] = arch;
  return SPK_OK;
}

// ---------------------------------------------------------------- HIP-event profiling
#include <vector>
#include <string>
#include <map>
struct ProfRec { std::string tag; hipEvent_t a, b; };
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
static std::string g_report;

bool spk_prof_enabled() { return g_prof; }
void spk_prof_begin(const char* tag, hipStream_t stream) {
  ProfRec r; r.tag = tag;
  if (!g_pool.empty()) { r.a = g_pool.back().first; r.b = g_pool.back().second; g_pool.pop_back(); }
  else { if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return; }
  (void)hipEventRecord(r.a, stream);
  g_recs.push_back(r);
}
void spk_prof_end(hipStream_t stream) {
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, stream);
}
extern "C" void spk_profile_enable(int on) { g_prof = on != 0; }
// Synchronises the device, folds all recorded (begin,end) pairs into per-tag totals and returns a
// text table "tag count total_ms\n..." (valid until the next call); clears the records.
extern "C" const char* spk_profile_report(void) {
  (void)hipDeviceSynchronize();
  std::map<std::string, std::pair<long, double>> acc;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { acc[r.tag].first += 1; acc[r.tag].second += ms; }
    g_pool.push_back({r.a, r.b});
  }
  g_recs.clear();
  g_report.clear();
  char line[256];
  for (auto& kv : acc) {
    snprintf(line, sizeof(line), "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    g_report += line;
  }
  return g_report.c_str();
}

// ---------------------------------------------------------------- neighbour-list plan
__global__ void k_plan_flags(const int64_t* __restrict__ idx_i, const int64_t* __restrict__ idx_j,
                             int64_t E, int64_t N, int32_t* flags) {
  int bad_sort = 0, bad_range = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = idx_i[e], j = idx_j[e];
    if (i < 0 || i >= N || j < 0 || j >= N) bad_range = 1;
    if (e > 0 && idx_i[e - 1] > i) bad_sort = 1;
  }
  if (bad_sort) atomicOr(&flags[0], 1);
  if (bad_range) atomicOr(&flags[1], 1);
}

// rowptr[r] = first edge e with idx_i[e] >= r  (requires ascending idx_i)
__global__ void k_rowptr(const int64_t* __restrict__ idx_i, int64_t E, int64_t N,
                         int32_t* __restrict__ rowptr) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e <= E;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t prev = (e > 0) ? idx_i[e - 1] : -1;
    int64_t cur = (e < E) ? idx_i[e] : N;
    if (prev < -1) prev = -1;
    if (cur > N) cur = N;
    for (int64_t r = prev + 1; r <= cur; ++r) rowptr[r] = (int32_t)e;
  }
}

// CSR row pointers of an ascending index on the device only (no host round trip: usable inside a HIP graph).
// err[0] |= 1 if the index is not ascending, |= 2 if an entry is outside [0, n_rows).
__global__ void k_rowptr_checked(const int64_t* __restrict__ idx, int64_t E, int64_t N, int32_t* __restrict__ rowptr,
                                 int32_t* __restrict__ err) {
  int bad = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e <= E; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t prev = (e > 0) ? idx[e - 1] : -1;
    int64_t cur = (e < E) ? idx[e] : N;
    if (e < E && (cur < 0 || cur >= N)) bad |= 2;
    if (e > 0 && e < E && prev > cur) bad |= 1;
    if (prev < -1) prev = -1;
    if (cur > N) cur = N;
    for (int64_t r = prev + 1; r <= cur; ++r) rowptr[r] = (int32_t)e;
  }
  if (bad && err) atomicOr(err, bad);
}

extern "C" int spk_segment_rowptr_i32(const int64_t* idx, int64_t n, int64_t n_rows, int32_t* rowptr, int32_t* err,
                                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n >= 0 && n_rows >= 0 && n < (1LL << 31) && rowptr != nullptr && (n == 0 || idx != nullptr),
                "spk_segment_rowptr_i32: bad input");
  hipLaunchKernelGGL(k_rowptr_checked, dim3(spk_grid_for(n + 1, 256, 4096)), dim3(256), 0, stream, idx, n, n_rows, rowptr, err);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// err[0] |= 2 if an entry of idx is outside [0, hi)  (device only: neighbour indices / atomic numbers of a static-shape step)
__global__ void k_range_checked(const int64_t* __restrict__ idx, int64_t n, int64_t hi, int32_t* __restrict__ err) {
  int bad = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    if ((uint64_t)idx[e] >= (uint64_t)hi) bad = 2;
  if (bad) atomicOr(err, bad);
}

extern "C" int spk_index_range_check(const int64_t* idx, int64_t n, int64_t hi, int32_t* err, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n >= 0 && hi >= 0 && err != nullptr && (n == 0 || idx != nullptr), "spk_index_range_check: bad input");
  if (n == 0) return SPK_OK;
  hipLaunchKernelGGL(k_range_checked, dim3(spk_grid_for(n, 256, 4096)), dim3(256), 0, stream, idx, n, hi, err);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// Several index jobs in ONE launch (blockIdx.y = job): CSR row pointers of an ascending index with the checks of k_rowptr_checked, or (rowptr
// NULL) the range check of k_range_checked.  A static-shape training step validates four index arrays and the force-matching engine derives two
// row-pointer arrays per step: as separate launches that is six of the ~100 launches of a step for a few microseconds of work.
struct SpkIndexJobs { int n; spk_index_job_t job[SPK_INDEX_JOBS_MAX]; };
__global__ void k_index_jobs(SpkIndexJobs J, int32_t* __restrict__ err) {
  const spk_index_job_t jb = J.job[blockIdx.y];
  const int64_t* __restrict__ idx = jb.idx;
  const int64_t E = jb.n, N = jb.rows;
  int bad = 0;
  if (jb.rowptr) {
    int32_t* __restrict__ rowptr = jb.rowptr;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e <= E; e += (int64_t)gridDim.x * blockDim.x) {
      int64_t prev = (e > 0) ? idx[e - 1] : -1;
      int64_t cur = (e < E) ? idx[e] : N;
      if (e < E && (cur < 0 || cur >= N)) bad |= 2;
      if (e > 0 && e < E && prev > cur) bad |= 1;
      if (prev < -1) prev = -1;
      if (cur > N) cur = N;
      for (int64_t r = prev + 1; r <= cur; ++r) rowptr[r] = (int32_t)e;
    }
  } else {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x)
      if ((uint64_t)idx[e] >= (uint64_t)N) bad = 2;
  }
  if (bad && err) atomicOr(err, bad);
}
extern "C" int spk_index_jobs(const spk_index_job_t* jobs, int32_t n_jobs, int32_t* err, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(n_jobs >= 0 && n_jobs <= SPK_INDEX_JOBS_MAX && (n_jobs == 0 || jobs != nullptr), "spk_index_jobs: at most %d jobs", SPK_INDEX_JOBS_MAX);
  if (n_jobs == 0) return SPK_OK;
  SpkIndexJobs J;
  J.n = n_jobs;
  int64_t longest = 1;
  for (int k = 0; k < n_jobs; ++k) {
    SPK_CHECK_ARG(jobs[k].n >= 0 && jobs[k].rows >= 0 && jobs[k].n < (1LL << 31) && (jobs[k].n == 0 || jobs[k].idx != nullptr), "spk_index_jobs: bad job %d", k);
    SPK_CHECK_ARG(jobs[k].rowptr != nullptr || err != nullptr, "spk_index_jobs: a range check needs the error word");
    J.job[k] = jobs[k];
    if (jobs[k].n + 1 > longest) longest = jobs[k].n + 1;
  }
  hipLaunchKernelGGL(k_index_jobs, dim3(spk_grid_for(longest, 256, 1024), n_jobs), dim3(256), 0, stream, J, err);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// every edge (i<-j, r) must have a partner (j<-i, -r) in row j
__global__ void k_symmetry(const int64_t* __restrict__ idx_i, const int64_t* __restrict__ idx_j,
                           const float* __restrict__ rij, const int32_t* __restrict__ rowptr,
                           int64_t E, int32_t* flags, int32_t* __restrict__ rev) {
  int asym = 0;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = idx_i[e], j = idx_j[e];
    float rx = rij[3 * e], ry = rij[3 * e + 1], rz = rij[3 * e + 2];
    int32_t found = -1;
    for (int32_t q = rowptr[j]; q < rowptr[j + 1]; ++q) {
      if (idx_j[q] == i && rij[3 * (int64_t)q] == -rx && rij[3 * (int64_t)q + 1] == -ry &&
          rij[3 * (int64_t)q + 2] == -rz) { found = q; break; }
    }
    if (rev) rev[e] = found;
    if (found < 0 || found == e) asym = 1;
  }
  if (asym) atomicOr(&flags[2], 1);
}

extern "C" int spk_edge_plan(const int64_t* idx_i, const int64_t* idx_j, const float* r_ij,
                             int64_t E, int64_t N, int32_t* rowptr, int32_t* rev, int32_t* scratch,
                             int32_t* host_flags, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(E >= 0 && N >= 0 && N < (1LL << 31) && E < (1LL << 31),
                "spk_edge_plan: sizes out of range (N=%lld E=%lld)", (long long)N, (long long)E);
  SPK_CHECK_ARG(rowptr && scratch && host_flags, "spk_edge_plan: null pointer");
  SPK_CHECK_ARG(E == 0 || (idx_i && idx_j), "spk_edge_plan: null index arrays");
  { int _zr = spk_zero_async(scratch, 4 * sizeof(int32_t), stream); if (_zr) return _zr; }
  int32_t f[4] = {0, 0, 0, 0};
  if (E > 0) {
    int grid = spk_grid_for(E, 256, 4096);
    hipLaunchKernelGGL(k_plan_flags, dim3(grid), dim3(256), 0, stream, idx_i, idx_j, E, N, scratch);
    SPK_LAUNCH_CHECK();
    SPK_HIP_TRY(hipMemcpyAsync(f, scratch, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    SPK_HIP_TRY(hipStreamSynchronize(stream));
  }
  int sorted = !f[0], in_range = !f[1], symmetric = 0;
  if (sorted && in_range) {
    int grid = spk_grid_for(E + 1, 256, 4096);
    hipLaunchKernelGGL(k_rowptr, dim3(grid), dim3(256), 0, stream, idx_i, E, N, rowptr);
    SPK_LAUNCH_CHECK();
    if (r_ij && E > 0) {
      hipLaunchKernelGGL(k_symmetry, dim3(spk_grid_for(E, 256, 4096)), dim3(256), 0, stream, idx_i,
                         idx_j, r_ij, rowptr, E, scratch, rev);
      SPK_LAUNCH_CHECK();
      SPK_HIP_TRY(hipMemcpyAsync(f, scratch, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
      SPK_HIP_TRY(hipStreamSynchronize(stream));
      symmetric = !f[2];
    } else if (E == 0) {
      symmetric = 1;
    }
  }
  host_flags[0] = sorted;
  host_flags[1] = in_range;
  host_flags[2] = symmetric;
  if (!in_range) {
    spk_set_error("spk_edge_plan: neighbour index out of range [0, %lld)", (long long)N);
    return SPK_ERR_INDEX;
  }
  return SPK_OK;
}

// ---------------------------------------------------------------- scatter_add (nn/scatter.py)
// Segmented sum over CSR rows: one thread per (outer, row, VEC-chunk of inner); consecutive
// threads read consecutive addresses of every edge row (coalesced), y written exactly once.
// UNR entries of the segment are in flight per thread, the tail as one more PREDICATED batch (clamped address, value masked) -- never a loop
// of single dependent loads; the sums run in entry order whatever UNR is, so the result does not depend on it.  Streaming from DRAM the kernel is
// bound by the loads it keeps outstanding (round 5: SPK_SEGSUM_UNROLL = 4 / 8 / 16 selects the instance for A/B runs; default 8).
template <int VEC, int UNR>
__global__ void k_segsum(const float* __restrict__ x, const int32_t* __restrict__ rowptr,
                         int64_t outer, int64_t E, int64_t inner, int64_t N,
                         float* __restrict__ y) {
  const int64_t cpr = inner / VEC;  // chunks per row
  const int64_t total = outer * N * cpr;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t c = t % cpr;
    int64_t k = (t / cpr) % N;
    int64_t o = t / (cpr * N);
    int32_t e0 = rowptr[k], e1 = rowptr[k + 1];
    const float* xp = x + (o * E) * inner + c * VEC;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int32_t e = e0; e < e1; e += UNR) {
      float tmp[UNR][VEC];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const bool ok = e + u < e1;
        const float* p = xp + (int64_t)(ok ? e + u : e1 - 1) * inner;
        if (VEC == 4) { f32x4 q = *(const f32x4*)p; tmp[u][0] = ok ? q.x : 0.f; tmp[u][1 % VEC] = ok ? q.y : 0.f; tmp[u][2 % VEC] = ok ? q.z : 0.f; tmp[u][3 % VEC] = ok ? q.w : 0.f; }
        else { tmp[u][0] = ok ? p[0] : 0.f; }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += tmp[u][v];
    }
    float* yp = y + (o * N + k) * inner + c * VEC;
    if (VEC == 4) { f32x4 q; q.x = acc[0]; q.y = acc[1 % VEC]; q.z = acc[2 % VEC]; q.w = acc[3 % VEC]; *(f32x4*)yp = q; }
    else { yp[0] = acc[0]; }
  }
}

// Few, long rows (e.g. per-atom energies of ONE 32 k-atom system summed into one molecule): one WAVEFRONT per
// (outer, row, channel), lanes stride over the row's entries, wave reduction at the end -- the one-thread-per-row
// kernel above would walk such a row serially.
__global__ void k_segsum_wave(const float* __restrict__ x, const int32_t* __restrict__ rowptr, int64_t outer, int64_t E,
                              int64_t inner, int64_t N, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t total = outer * N * inner;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; t < total; t += nwaves) {
    const int64_t c = t % inner, k = (t / inner) % N, o = t / (inner * N);
    const int32_t e0 = rowptr[k], e1 = rowptr[k + 1];
    const float* xp = x + (o * E) * inner + c;
    float acc = 0.f;
    for (int32_t e = e0 + lane; e < e1; e += 64) acc += xp[(int64_t)e * inner];
    acc = spk_wave_sum(acc);
    if (lane == 0) y[(o * N + k) * inner + c] = acc;
  }
}

__global__ void k_scatter_atomic(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                 int64_t outer, int64_t E, int64_t inner, int64_t N,
                                 float* __restrict__ y) {
  const int64_t total = outer * E * inner;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t c = t % inner;
    int64_t e = (t / inner) % E;
    int64_t o = t / (inner * E);
    int64_t k = idx[e];
    if (k >= 0 && k < N) unsafeAtomicAdd(&y[(o * N + k) * inner + c], x[t]);
  }
}

extern "C" int spk_scatter_add_f32(const float* x, const int64_t* idx, const int32_t* rowptr,
                                   int64_t outer, int64_t E, int64_t inner, int64_t N, float* y,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(outer >= 0 && E >= 0 && inner >= 0 && N >= 0, "spk_scatter_add_f32: negative size");
  int64_t out_elems = outer * N * inner;
  if (out_elems == 0) return SPK_OK;
  SPK_CHECK_ARG(y != nullptr, "spk_scatter_add_f32: null output");
  if (E == 0) { { int _zr = spk_zero_async(y, out_elems * sizeof(float), stream); if (_zr) return _zr; } return SPK_OK; }
  SPK_CHECK_ARG(x != nullptr && (idx != nullptr || rowptr != nullptr), "spk_scatter_add_f32: null input");
  const int maxb = spk_num_cus() * 16;
  if (rowptr) {
    bool vec4 = (inner % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    if (outer * N * inner <= 4096 && E >= 256 * N) {   // few long rows
      hipLaunchKernelGGL(k_segsum_wave, dim3(spk_grid_for(outer * N * inner * 64, 256, maxb)), dim3(256), 0, stream, x, rowptr, outer, E, inner, N, y);
    } else if (vec4) {
      int grid = spk_grid_for(outer * N * (inner / 4), 256, maxb);
      static const int unr = [] { const char* e = getenv("SPK_SEGSUM_UNROLL"); const int v = e ? atoi(e) : 8; return v == 4 || v == 16 ? v : 8; }();
      SpkProfScope prof("scatter_add_segsum", stream);
      if (unr == 4) hipLaunchKernelGGL((k_segsum<4, 4>), dim3(grid), dim3(256), 0, stream, x, rowptr, outer, E, inner, N, y);
      else if (unr == 16) hipLaunchKernelGGL((k_segsum<4, 16>), dim3(grid), dim3(256), 0, stream, x, rowptr, outer, E, inner, N, y);
      else hipLaunchKernelGGL((k_segsum<4, 8>), dim3(grid), dim3(256), 0, stream, x, rowptr, outer, E, inner, N, y);
    } else {
      int grid = spk_grid_for(outer * N * inner, 256, maxb);
      hipLaunchKernelGGL((k_segsum<1, 8>), dim3(grid), dim3(256), 0, stream, x, rowptr, outer, E, inner, N, y);
    }
    SPK_LAUNCH_CHECK();
  } else {
    { int _zr = spk_zero_async(y, out_elems * sizeof(float), stream); if (_zr) return _zr; }
    int grid = spk_grid_for(outer * E * inner, 256, maxb);
    hipLaunchKernelGGL(k_scatter_atomic, dim3(grid), dim3(256), 0, stream, x, idx, outer, E, inner, N, y);
    SPK_LAUNCH_CHECK();
  }
  return SPK_OK;
}

template <int VEC>
__global__ void k_gather(const float* __restrict__ x, const int64_t* __restrict__ idx,
                         int64_t outer, int64_t R, int64_t E, int64_t inner,
                         float* __restrict__ y) {
  const int64_t cpr = inner / VEC;
  const int64_t total = outer * E * cpr;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t c = t % cpr;
    int64_t e = (t / cpr) % E;
    int64_t o = t / (cpr * E);
    int64_t k = idx[e];
    const bool ok = (uint64_t)k < (uint64_t)R;      // an index out of range reads nothing (rows of zeros), never out of bounds
    const float* xp = x + (o * R + (ok ? k : 0)) * inner + c * VEC;
    float* yp = y + (o * E + e) * inner + c * VEC;
    if (VEC == 4) *(f32x4*)yp = ok ? *(const f32x4*)xp : f32x4{0.f, 0.f, 0.f, 0.f};
    else yp[0] = ok ? xp[0] : 0.f;
  }
}

extern "C" int spk_gather_f32(const float* x, const int64_t* idx, int64_t outer, int64_t R,
                              int64_t E, int64_t inner, float* y, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(outer >= 0 && E >= 0 && inner >= 0 && R >= 0, "spk_gather_f32: negative size");
  if (outer * E * inner == 0) return SPK_OK;
  SPK_CHECK_ARG(x && idx && y, "spk_gather_f32: null pointer");
  const int maxb = spk_num_cus() * 16;
  bool vec4 = (inner % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
  if (vec4)
    hipLaunchKernelGGL(k_gather<4>, dim3(spk_grid_for(outer * E * (inner / 4), 256, maxb)), dim3(256), 0, stream, x, idx, outer, R, E, inner, y);
  else
    hipLaunchKernelGGL(k_gather<1>, dim3(spk_grid_for(outer * E * inner, 256, maxb)), dim3(256), 0, stream, x, idx, outer, R, E, inner, y);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ---------------------------------------------------------------- radial basis x cutoff
__global__ void k_radial_cutoff(const float* __restrict__ d, int64_t n, RadialDev rb,
                                float* __restrict__ phi, float* __restrict__ fcut) {
  const int K = rb.n_rbf;
  const int64_t total = n * (int64_t)(K + 1);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t e = t / (K + 1);
    int k = (int)(t % (K + 1));
    float dd = d[e];
    if (k < K) {
      if (phi) { float p, dp; spk_rbf_eval(rb, k, dd, p, dp); phi[e * K + k] = p; }
    } else if (fcut) {
      float f, df; spk_cutoff_eval(rb.cutoff, dd, f, df); fcut[e] = f;
    }
  }
}

__global__ void k_radial_cutoff_bwd(const float* __restrict__ d, int64_t n, RadialDev rb,
                                    const float* __restrict__ gphi, const float* __restrict__ gfcut,
                                    float* __restrict__ gd) {
  const int K = rb.n_rbf;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    float dd = d[e];
    float acc = 0.f;
    if (gphi)
      for (int k = 0; k < K; ++k) { float p, dp; spk_rbf_eval(rb, k, dd, p, dp); acc += gphi[e * K + k] * dp; }
    if (gfcut) { float f, df; spk_cutoff_eval(rb.cutoff, dd, f, df); acc += gfcut[e] * df; }
    gd[e] = acc;
  }
}

static int check_radial(const spk_radial_t* rb, const char* who) {
  SPK_CHECK_ARG(rb != nullptr, "%s: null radial description", who);
  SPK_CHECK_ARG(rb->kind == SPK_RBF_GAUSSIAN || rb->kind == SPK_RBF_BESSEL, "%s: unknown rbf kind %d", who, rb->kind);
  SPK_CHECK_ARG(rb->n_rbf >= 1 && rb->n_rbf <= 1024, "%s: n_rbf=%d unsupported", who, rb->n_rbf);
  SPK_CHECK_ARG(rb->p0 != nullptr && (rb->kind == SPK_RBF_BESSEL || rb->p1 != nullptr), "%s: null rbf parameters", who);
  SPK_CHECK_ARG(rb->cutoff > 0.f, "%s: cutoff must be positive", who);
  return SPK_OK;
}

extern "C" int spk_radial_cutoff_f32(const float* d, int64_t n, const spk_radial_t* rb, float* phi,
                                     float* fcut, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_radial(rb, "spk_radial_cutoff_f32");
  if (rc) return rc;
  if (n == 0) return SPK_OK;
  SPK_CHECK_ARG(d != nullptr && n > 0, "spk_radial_cutoff_f32: bad input");
  int grid = spk_grid_for(n * (rb->n_rbf + 1), 256, spk_num_cus() * 16);
  hipLaunchKernelGGL(k_radial_cutoff, dim3(grid), dim3(256), 0, stream, d, n, spk_radial_dev(rb), phi, fcut);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_radial_cutoff_bwd_f32(const float* d, int64_t n, const spk_radial_t* rb,
                                         const float* gphi, const float* gfcut, float* gd,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_radial(rb, "spk_radial_cutoff_bwd_f32");
  if (rc) return rc;
  if (n == 0) return SPK_OK;
  SPK_CHECK_ARG(d != nullptr && gd != nullptr && n > 0, "spk_radial_cutoff_bwd_f32: bad input");
  int grid = spk_grid_for(n, 256, spk_num_cus() * 16);
  hipLaunchKernelGGL(k_radial_cutoff_bwd, dim3(grid), dim3(256), 0, stream, d, n, spk_radial_dev(rb), gphi, gfcut, gd);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

__global__ void k_edge_norm(const float* __restrict__ rij, int64_t E, float* __restrict__ d,
                            float* __restrict__ u) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E;
       e += (int64_t)gridDim.x * blockDim.x) {
    float x = rij[3 * e], y = rij[3 * e + 1], z = rij[3 * e + 2];
    float dd = sqrtf(x * x + y * y + z * z);
    if (d) d[e] = dd;
    if (u) { u[3 * e] = x / dd; u[3 * e + 1] = y / dd; u[3 * e + 2] = z / dd; }
  }
}

extern "C" int spk_edge_norm_f32(const float* r_ij, int64_t E, float* d, float* u, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (E == 0) return SPK_OK;
  SPK_CHECK_ARG(r_ij != nullptr && E > 0, "spk_edge_norm_f32: bad input");
  hipLaunchKernelGGL(k_edge_norm, dim3(spk_grid_for(E, 256, spk_num_cus() * 16)), dim3(256), 0, stream, r_ij, E, d, u);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ---------------------------------------------------------------- zero fill (kernel, never a memset node)
__global__ void k_zero_words(uint32_t* __restrict__ p, size_t n) {
  const size_t n4 = n / 4;
  uint4 z = {0u, 0u, 0u, 0u};
  for (size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x; s < n4; s += (size_t)gridDim.x * blockDim.x) ((uint4*)p)[s] = z;
  for (size_t s = 4 * n4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; s < n; s += (size_t)gridDim.x * blockDim.x) p[s] = 0u;
}
__global__ void k_zero_words_unaligned(uint32_t* __restrict__ p, size_t n) {
  for (size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x; s < n; s += (size_t)gridDim.x * blockDim.x) p[s] = 0u;
}
int spk_zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return SPK_OK;
  SPK_CHECK_ARG(p != nullptr && bytes % 4 == 0 && ((uintptr_t)p % 4) == 0, "spk_zero_async: needs a 4-byte aligned buffer of whole words");
  const size_t n = bytes / 4;
  const int grid = spk_grid_for((int64_t)((n + 3) / 4), 256, spk_num_cus() * 8);
  if (((uintptr_t)p % 16) == 0) hipLaunchKernelGGL(k_zero_words, dim3(grid), dim3(256), 0, stream, (uint32_t*)p, n);
  else hipLaunchKernelGGL(k_zero_words_unaligned, dim3(spk_grid_for((int64_t)n, 256, spk_num_cus() * 8)), dim3(256), 0, stream, (uint32_t*)p, n);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ---------------------------------------------------------------- pairwise vectors (distances.py)
__global__ void k_pairwise(const float* __restrict__ R, const int64_t* __restrict__ idx_i,
                           const int64_t* __restrict__ idx_j, const float* __restrict__ off,
                           int64_t E, int64_t N, float* __restrict__ rij) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = idx_i[e], j = idx_j[e];
    if (N > 0) {   // a malformed list must not read out of bounds: clamp here, spk_edge_plan reports the index error
      i = i < 0 ? 0 : (i >= N ? N - 1 : i);
      j = j < 0 ? 0 : (j >= N ? N - 1 : j);
    }
    // same operation order as the reference: (R[j] - R[i]) + offsets  => r_ji == -r_ij bit-exactly
    float x = R[3 * j] - R[3 * i], y = R[3 * j + 1] - R[3 * i + 1], z = R[3 * j + 2] - R[3 * i + 2];
    if (off) { x += off[3 * e]; y += off[3 * e + 1]; z += off[3 * e + 2]; }
    rij[3 * e] = x; rij[3 * e + 1] = y; rij[3 * e + 2] = z;
  }
}

__global__ void k_pairwise_bwd(const float* __restrict__ gr, const int64_t* __restrict__ idx_i,
                               const int64_t* __restrict__ idx_j, int64_t E, int64_t N, float* __restrict__ gR) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx_i[e], j = idx_j[e];
    if ((uint64_t)i >= (uint64_t)N || (uint64_t)j >= (uint64_t)N) continue;      // never scatter out of bounds
    const float x = gr[3 * e], y = gr[3 * e + 1], z = gr[3 * e + 2];
    unsafeAtomicAdd(&gR[3 * j], x); unsafeAtomicAdd(&gR[3 * j + 1], y); unsafeAtomicAdd(&gR[3 * j + 2], z);
    unsafeAtomicAdd(&gR[3 * i], -x); unsafeAtomicAdd(&gR[3 * i + 1], -y); unsafeAtomicAdd(&gR[3 * i + 2], -z);
  }
}

// Symmetric + sorted lists: every edge e = (i <- j) has its reverse rev[e] = (j <- i) in the list, so
//   gR[a] = sum_{e: idx_j[e]==a} gr[e] - sum_{e: idx_i[e]==a} gr[e] = sum_{e in row(a)} (gr[rev[e]] - gr[e]):
// a segmented sum over the CSR row of the atom -- no atomics, no memset, deterministic.  16 lanes per
// atom (molecular lists have ~15 neighbours, bulk water ~52).
__global__ void k_pairwise_bwd_row(const float* __restrict__ gr, const int32_t* __restrict__ rowptr,
                                   const int32_t* __restrict__ rev, int64_t N, float* __restrict__ gR) {
  const int sub = threadIdx.x & 15;
  for (int64_t a = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4; a < N;
       a += ((int64_t)gridDim.x * blockDim.x) >> 4) {
    const int e0 = rowptr[a], e1 = rowptr[a + 1];
    float x = 0.f, y = 0.f, z = 0.f;
    for (int e = e0 + sub; e < e1; e += 16) {
      const int q = rev[e];
      x += gr[3 * (int64_t)q] - gr[3 * (int64_t)e];
      y += gr[3 * (int64_t)q + 1] - gr[3 * (int64_t)e + 1];
      z += gr[3 * (int64_t)q + 2] - gr[3 * (int64_t)e + 2];
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
      x += __shfl_xor(x, m, 64); y += __shfl_xor(y, m, 64); z += __shfl_xor(z, m, 64);
    }
    if (sub == 0) { gR[3 * a] = x; gR[3 * a + 1] = y; gR[3 * a + 2] = z; }
  }
}

extern "C" int spk_pairwise_f32(const float* R, const int64_t* idx_i, const int64_t* idx_j,
                                const float* offsets, int64_t E, float* r_ij, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (E == 0) return SPK_OK;
  SPK_CHECK_ARG(R && idx_i && idx_j && r_ij && E > 0, "spk_pairwise_f32: bad input");
  hipLaunchKernelGGL(k_pairwise, dim3(spk_grid_for(E, 256, spk_num_cus() * 16)), dim3(256), 0, stream, R, idx_i, idx_j, offsets, E, (int64_t)0, r_ij);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// the same with the number of atoms known: indices outside [0, n_atoms) are clamped for the read (no out-of-bounds access);
// the index error itself is reported by spk_edge_plan of the list
extern "C" int spk_pairwise_n_f32(const float* R, const int64_t* idx_i, const int64_t* idx_j, const float* offsets, int64_t E,
                                  int64_t n_atoms, float* r_ij, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (E == 0) return SPK_OK;
  SPK_CHECK_ARG(R && idx_i && idx_j && r_ij && E > 0 && n_atoms > 0, "spk_pairwise_n_f32: bad input");
  hipLaunchKernelGGL(k_pairwise, dim3(spk_grid_for(E, 256, spk_num_cus() * 16)), dim3(256), 0, stream, R, idx_i, idx_j, offsets, E, n_atoms, r_ij);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_pairwise_bwd_f32(const float* gr, const int64_t* idx_i, const int64_t* idx_j,
                                    int64_t E, int64_t N, float* gR, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) return SPK_OK;
  SPK_CHECK_ARG(gR != nullptr && N > 0 && E >= 0, "spk_pairwise_bwd_f32: bad input");
  { int _zr = spk_zero_async(gR, (size_t)N * 3 * sizeof(float), stream); if (_zr) return _zr; }
  if (E == 0) return SPK_OK;
  SPK_CHECK_ARG(gr && idx_i && idx_j, "spk_pairwise_bwd_f32: null pointer");
  hipLaunchKernelGGL(k_pairwise_bwd, dim3(spk_grid_for(E, 256, spk_num_cus() * 16)), dim3(256), 0, stream, gr, idx_i, idx_j, E, N, gR);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_pairwise_bwd_graph_f32(const float* gr, const spk_graph_t* g, float* gR, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(g != nullptr, "spk_pairwise_bwd_graph_f32: null graph");
  if (!(g->sorted && g->symmetric && g->rowptr && g->rev) || g->n_edges == 0)
    return spk_pairwise_bwd_f32(gr, g->idx_i, g->idx_j, g->n_edges, g->n_atoms, gR, stream_);
  SPK_CHECK_ARG(gr && gR && g->n_atoms > 0, "spk_pairwise_bwd_graph_f32: bad input");
  SpkProfScope prof("pairwise_bwd_row", stream);
  hipLaunchKernelGGL(k_pairwise_bwd_row, dim3(spk_grid_for(g->n_atoms * 16, 256, spk_num_cus() * 16)), dim3(256), 0, stream,
                     gr, g->rowptr, g->rev, g->n_atoms, gR);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ---------------------------------------------------------------- embedding / add
__global__ void k_embedding(const float* __restrict__ table, const int64_t* __restrict__ z,
                            int64_t n, int F, float* __restrict__ out) {
  const int64_t total = n * (int64_t)F;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t a = t / F;
    int f = (int)(t % F);
    out[t] = table[z[a] * F + f];
  }
}

extern "C" int spk_embedding_f32(const float* table, const int64_t* z, int64_t n, int32_t F,
                                 float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return SPK_OK;
  SPK_CHECK_ARG(table && z && out && F > 0 && n > 0, "spk_embedding_f32: bad input");
  hipLaunchKernelGGL(k_embedding, dim3(spk_grid_for(n * F, 256, spk_num_cus() * 16)), dim3(256), 0, stream, table, z, n, F, out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

__global__ void k_add(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                      float* __restrict__ y) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n;
       t += (int64_t)gridDim.x * blockDim.x)
    y[t] = a[t] + b[t];
}

extern "C" int spk_add_f32(const float* a, const float* b, int64_t n, float* y, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return SPK_OK;
  SPK_CHECK_ARG(a && b && y && n > 0, "spk_add_f32: bad input");
  hipLaunchKernelGGL(k_add, dim3(spk_grid_for(n, 256, spk_num_cus() * 16)), dim3(256), 0, stream, a, b, n, y);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
