// EXPERIMENT (round 3, SURVEY.md section 8(d): "cfconv as a whole is MFMA-bound unless the filter MLP is tabulated"):
// continuous-filter convolution (representation/schnet.py:60-67) with the filter  W_l(d) f_c(d)  read from a TABLE instead of
// being evaluated by the filter network.  The filter is a smooth function of ONE variable per channel, so a cubic-Hermite spline
// over n_knots equidistant knots (value + slope, built in float64 from the weights whenever they change) replaces the
// 2 (n_rbf nf + nf nf) = 37.9 kFLOP per edge-message of the MLP by ~10 FLOP and one 2-KB table row per edge.  Eval-only, default
// OFF: the fp32 MFMA kernels stay the contract path; scripts/tab_filter_experiment.py measures time and error of this kernel
// beside them (profiles/r03_tabulated_filter_experiment.json).
//
//   y[i, c] = sum_{e in row(i)} h[j(e), c] * T_c(d_e),   T_c(d) = Hermite(table[n], table[n + 1], t),  n = floor(d / step), t = frac
//
// table: [n_knots][nf][4] floats = (value, slope * step, value of the next knot - value [formed in float64], slope * step of the
// next knot) at d_n = n * step -- one 16-byte read per channel and interval; entries beyond the cutoff are zero.  One wavefront per
// centre atom (CSR row, sorted idx_i), a lane owns two channels: a table row is one coalesced 1-KB burst, the neighbour row
// 512 bytes; no atomics; row sums in registers.  Bound: L2 / HBM gather bandwidth (2.5 KB per edge).
#include "spk_common.h"

typedef float tf2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_cfconv_tab(const float* __restrict__ h, const float* __restrict__ rij, const int64_t* __restrict__ idx_j,
                                                    const int32_t* __restrict__ rowptr, const float* __restrict__ table, int n_knots, float inv_step,
                                                    float cutoff, int64_t N, float* __restrict__ y, float* __restrict__ dy_dd /* [E] or null */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t atom = (int64_t)blockIdx.x * 4 + wv; atom < N; atom += (int64_t)gridDim.x * 4) {
    const int32_t e0 = rowptr[atom], e1 = rowptr[atom + 1];
    tf2 acc = {0.f, 0.f};
    for (int32_t cs = e0; cs < e1; cs += 64) {
      // lanes = edges: geometry of up to 64 edges of the row at once
      const int32_t em = cs + lane;
      const bool ev = em < e1;
      const int32_t emc = ev ? em : (e1 - 1);
      const int jl = (int)idx_j[emc];
      const float rx = rij[3 * (int64_t)emc], ry = rij[3 * (int64_t)emc + 1], rz = rij[3 * (int64_t)emc + 2];
      const float dl = sqrtf(rx * rx + ry * ry + rz * rz);
      uint64_t live = __ballot(ev && dl < cutoff);
      while (live) {
        const int t = __ffsll((long long)live) - 1;
        live &= live - 1;
        const int64_t j = __builtin_amdgcn_readlane(jl, t);
        const float d = spk_readlane_f(dl, t);
        const float u = d * inv_step;
        int n = (int)u;
        n = n < n_knots - 2 ? n : n_knots - 2;
        const float s = u - (float)n;
        const f32x4 ka = *(const f32x4*)(table + ((size_t)n * 128 + 2 * lane) * 4);          // (v_n, m_n, v_n+1 - v_n, m_n+1) of channel 2 lane
        const f32x4 kb = *(const f32x4*)(table + ((size_t)n * 128 + 2 * lane + 1) * 4);      // ... of channel 2 lane + 1
        const tf2 hj = *(const tf2*)(h + j * 128 + 2 * lane);
        const float s2 = s * s, s3 = s2 * s;
        const float h10 = s3 - 2.f * s2 + s, h01 = -2.f * s3 + 3.f * s2, h11 = s3 - s2;
        const tf2 W = {ka.x + h10 * ka.y + h01 * ka.z + h11 * ka.w, kb.x + h10 * kb.y + h01 * kb.z + h11 * kb.w};
        acc += W * hj;
      }
    }
    *(tf2*)(y + atom * 128 + 2 * lane) = acc;
  }
}

// First-order backward of the table-driven convolution on SYMMETRIC sorted lists, one row pass, no atomics:
//   gh[i, c]  = sum_{e in row(i)} gy[j(e), c] T_c(d_e)                      (the transposed sum through the reverse edge: T_ij = T_ji)
//   gr[e]    (+)= ( sum_c gy[i, c] h[j(e), c] T_c'(d_e) ) r_e / d_e          (every directed edge exactly once per call)
// T_c' comes from the SAME spline (derivative of the Hermite basis / step).
__global__ __launch_bounds__(256) void k_cfconv_tab_bwd(const float* __restrict__ h, const float* __restrict__ gy, const float* __restrict__ rij,
                                                        const int64_t* __restrict__ idx_j, const int32_t* __restrict__ rowptr,
                                                        const float* __restrict__ table, int n_knots, float inv_step, float cutoff, int64_t N,
                                                        float* __restrict__ gh /* null: not wanted */, float* __restrict__ gr, int assign) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t atom = (int64_t)blockIdx.x * 4 + wv; atom < N; atom += (int64_t)gridDim.x * 4) {
    const int32_t e0 = rowptr[atom], e1 = rowptr[atom + 1];
    const tf2 gyi = *(const tf2*)(gy + atom * 128 + 2 * lane);
    tf2 acc = {0.f, 0.f};
    for (int32_t cs = e0; cs < e1; cs += 64) {
      const int32_t em = cs + lane;
      const bool ev = em < e1;
      const int32_t emc = ev ? em : (e1 - 1);
      const int jl = (int)idx_j[emc];
      const float rx = rij[3 * (int64_t)emc], ry = rij[3 * (int64_t)emc + 1], rz = rij[3 * (int64_t)emc + 2];
      const float dl = sqrtf(rx * rx + ry * ry + rz * rz);
      float sl = 0.f;                                   // d/dd sum of this lane's edge
      uint64_t live = __ballot(ev && dl < cutoff);
      while (live) {
        const int t = __ffsll((long long)live) - 1;
        live &= live - 1;
        const int64_t j = __builtin_amdgcn_readlane(jl, t);
        const float d = spk_readlane_f(dl, t);
        const float u = d * inv_step;
        int n = (int)u;
        n = n < n_knots - 2 ? n : n_knots - 2;
        const float s = u - (float)n;
        const f32x4 ka = *(const f32x4*)(table + ((size_t)n * 128 + 2 * lane) * 4);
        const f32x4 kb = *(const f32x4*)(table + ((size_t)n * 128 + 2 * lane + 1) * 4);
        const tf2 hj = *(const tf2*)(h + j * 128 + 2 * lane);
        const float s2 = s * s, s3 = s2 * s;
        if (gh) {
          const tf2 gyj = *(const tf2*)(gy + j * 128 + 2 * lane);
          const float h10 = s3 - 2.f * s2 + s, h01 = -2.f * s3 + 3.f * s2, h11 = s3 - s2;
          const tf2 W = {ka.x + h10 * ka.y + h01 * ka.z + h11 * ka.w, kb.x + h10 * kb.y + h01 * kb.z + h11 * kb.w};
          acc += W * gyj;
        }
        const float g10 = 3.f * s2 - 4.f * s + 1.f, g01 = 6.f * s - 6.f * s2, g11 = 3.f * s2 - 2.f * s;
        const tf2 dW = {g10 * ka.y + g01 * ka.z + g11 * ka.w, g10 * kb.y + g01 * kb.z + g11 * kb.w};
        const tf2 pv = gyi * hj * dW;
        float sum = pv.x + pv.y;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (lane == t) sl = sum * inv_step;
      }
      if (ev) {
        const float f = dl > 0.f ? sl / dl : 0.f;
        float* gp = gr + 3 * (int64_t)em;
        if (assign) { gp[0] = f * rx; gp[1] = f * ry; gp[2] = f * rz; }
        else { gp[0] += f * rx; gp[1] += f * ry; gp[2] += f * rz; }
      }
    }
    if (gh) *(tf2*)(gh + atom * 128 + 2 * lane) = acc;
  }
}

// ---- registry: tables are attached to a filter network by the device pointer of its filter_network.1.weight (the key the
// general SchNet driver hands to the cfconv launchers); spk_schnet_cfconv_* run the table kernels for registered layers
#include <mutex>
struct TabEntry { const float* key; const float* table; int n_knots; float d_max; uint64_t stamp; };      // stamp: version of the weights the table was built from (0 = untracked)
static TabEntry g_tabs[64];
static int g_ntabs = 0;
static std::mutex g_tab_mutex;

extern "C" int spk_filter_table_set(const float* key, const float* table, int32_t n_knots, float d_max) {
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  for (int i = 0; i < g_ntabs; ++i)
    if (g_tabs[i].key == key) {
      if (table) { g_tabs[i].table = table; g_tabs[i].n_knots = n_knots; g_tabs[i].d_max = d_max; g_tabs[i].stamp = 0; }
      else { g_tabs[i] = g_tabs[g_ntabs - 1]; --g_ntabs; }
      return SPK_OK;
    }
  if (!table) return SPK_OK;
  SPK_CHECK_ARG(key && n_knots >= 2 && d_max > 0.f, "spk_filter_table_set: bad arguments");
  SPK_CHECK_ARG(g_ntabs < 64, "spk_filter_table_set: more than 64 tabulated layers");
  g_tabs[g_ntabs++] = TabEntry{key, table, n_knots, d_max, 0};
  return SPK_OK;
}
// A table is a SNAPSHOT of the weights.  The caller that owns the weights records their version beside it (stamp) and asks, whenever
// it sees the weights again with a version of its own, whether the snapshot is still theirs: a stale table is dropped (the exact
// filter network runs again) instead of being served silently.  Returns 1 when an entry was dropped.
extern "C" int spk_filter_table_set_stamp(const float* key, uint64_t stamp) {
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  for (int i = 0; i < g_ntabs; ++i)
    if (g_tabs[i].key == key) { g_tabs[i].stamp = stamp; return SPK_OK; }
  return SPK_ERR_ARG;
}
extern "C" int spk_filter_table_drop_if_stale(const float* key, uint64_t stamp) {
  if (g_ntabs == 0) return 0;
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  for (int i = 0; i < g_ntabs; ++i)
    if (g_tabs[i].key == key) {
      if (g_tabs[i].stamp == 0 || g_tabs[i].stamp == stamp) return 0;
      g_tabs[i] = g_tabs[g_ntabs - 1];
      --g_ntabs;
      return 1;
    }
  return 0;
}
extern "C" void spk_filter_table_clear() {
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  g_ntabs = 0;
}
bool spk_filter_table_lookup(const float* key, const float** table, int* n_knots, float* d_max) {
  if (g_ntabs == 0) return false;
  std::lock_guard<std::mutex> lock(g_tab_mutex);
  for (int i = 0; i < g_ntabs; ++i)
    if (g_tabs[i].key == key) { *table = g_tabs[i].table; *n_knots = g_tabs[i].n_knots; *d_max = g_tabs[i].d_max; return true; }
  return false;
}
bool spk_filter_tables_active() { return g_ntabs > 0; }

int spk_cfconv_tab_bwd_internal(const spk_graph_t* g, const float* r_ij, const float* h, const float* gy, const float* table, int n_knots, float d_max,
                                float cutoff, float* gh, float* gr, bool gr_assign, hipStream_t stream) {
  const float inv_step = (float)(n_knots - 1) / d_max;
  SpkProfScope prof(gh ? "cfconv_tab_bwd" : "cfconv_tab_bwd_geom", stream);
  hipLaunchKernelGGL(k_cfconv_tab_bwd, dim3(spk_grid_for(g->n_atoms, 4, spk_num_cus() * 8)), dim3(256), 0, stream, h, gy, r_ij, g->idx_j, g->rowptr, table,
                     n_knots, inv_step, cutoff, g->n_atoms, gh, gr, gr_assign ? 1 : 0);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// table: [n_knots, 128, 4]; the list must be sorted (rowptr given); nf = 128 only (experiment).  y [N, 128] is overwritten.
extern "C" int spk_cfconv_tab_f32(const spk_graph_t* g, const float* r_ij, const float* h, const float* table, int32_t n_knots, float d_max,
                                  float cutoff, int32_t nf, float* y, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPK_CHECK_ARG(g && g->sorted && g->rowptr && g->idx_j, "spk_cfconv_tab_f32: needs a sorted list with row pointers");
  SPK_CHECK_ARG(nf == 128 && n_knots >= 2 && d_max > 0.f, "spk_cfconv_tab_f32: nf = 128, n_knots >= 2 (experiment)");
  if (g->n_atoms == 0) return SPK_OK;
  SPK_CHECK_ARG(r_ij && h && table && y, "spk_cfconv_tab_f32: null pointer");
  const float inv_step = (float)(n_knots - 1) / d_max;
  SpkProfScope prof("cfconv_tab_fwd", stream);
  hipLaunchKernelGGL(k_cfconv_tab, dim3(spk_grid_for(g->n_atoms, 4, spk_num_cus() * 8)), dim3(256), 0, stream, h, r_ij, g->idx_j, g->rowptr, table, n_knots,
                     inv_step, cutoff, g->n_atoms, y, nullptr);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
