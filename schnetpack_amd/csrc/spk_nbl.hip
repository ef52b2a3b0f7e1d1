// Cell-list neighbour list on the GPU.
//
// Semantics of the reference's TorchNeighborList (transform/neighborlist.py:438-507) for a batch of
// independent systems (molecules / MD replicas): all DIRECTED pairs (i, j, S) with
//     | R_j - R_i + S . cell |  <  cutoff,        (i != j or S != 0),
// S integer cell shifts along the periodic axes, offsets = S . cell (:455-457), idx_i ascending.
// Within a row the order is deterministic: neighbouring bins in (dx, dy, dz) order, atoms of a bin by
// ascending index.  The list is symmetric by construction: the hit test of (j, i, -S) evaluates the
// exact negation of the vector tested for (i, j, S).
//
// Pipeline (all on the device, one D2H of the edge count):
//   k_nbl_desc   one block per system: inverse cell, perpendicular heights, bounding box along the
//                non-periodic axes, bins per axis (bin height >= cutoff where the cell allows it, never
//                more bins than atoms), search reach per axis
//   k_nbl_bin    one thread per atom: wrap into the cell, bin id (sort key)
//   rocprim      stable radix sort (bin id, atom id): atoms of a bin contiguous and ascending
//   k_nbl_bounds bin -> first sorted position (binary search)
//   k_nbl_pairs<false>  one wavefront per atom: count hits      -> exclusive scan -> rowptr (CSR)
//   k_nbl_pairs<true>   same walk, ballot-ordered writes of idx_i / idx_j / shifts / offsets
#include "spk_common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

struct NblSys {
  float cell[9];   // effective cell, row vectors (identity when the system has no periodic axis)
  float inv[9];    // frac = R . inv
  float fmin[3];   // fractional origin of the bin grid along non-periodic axes
  float fext[3];   // fractional extent of the bin grid (1 along periodic axes)
  int nb[3];
  int reach[3];
  int pbc[3];
  int bin0, atom0, natoms, bad;
};

struct NblWs {     // carve-up of the caller's workspace
  NblSys* sys;
  int* atom0;      // [n_sys + 1]
  int* key;        // [N] bin id per atom
  int* key_sorted; // [N]
  int* ids;        // [N] 0..N-1
  int* ids_sorted; // [N] atoms ordered by bin
  int* wrap;       // [N][3] integer cell wraps of the atoms
  int* bin_start;  // [B + 1]
  int* counts;     // [N] neighbours per atom
  int64_t* total;  // [2]: edge count, error flags
  void* sort_tmp;
  size_t sort_tmp_bytes;
  int64_t max_bins;
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t nbl_sort_tmp_bytes(int64_t n) {
  // the size query walks rocPRIM's device / config detection (tens of ms): once per power-of-two bucket
  static size_t cached[64] = {0};
  int b = 0;
  while (((int64_t)1 << b) < n && b < 62) ++b;
  if (cached[b] == 0) {
    size_t bytes = 0;
    int* k = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, (size_t)1 << b, 0, 32, (hipStream_t)0);
    cached[b] = bytes > 0 ? bytes : 1;
  }
  return cached[b];
}

static size_t nbl_carve(NblWs& w, void* base, int64_t N, int64_t n_sys) {
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* r = p ? (void*)(p + off) : nullptr; off += align_up(bytes); return r; };
  const int64_t B = N + n_sys;   // k_nbl_desc never makes more bins than atoms (min 1 per system)
  w.max_bins = B;
  w.sys = (NblSys*)take(sizeof(NblSys) * (size_t)(n_sys > 0 ? n_sys : 1));
  w.atom0 = (int*)take(4 * (size_t)(n_sys + 1));
  w.key = (int*)take(4 * (size_t)(N + 1));
  w.key_sorted = (int*)take(4 * (size_t)(N + 1));
  w.ids = (int*)take(4 * (size_t)(N + 1));
  w.ids_sorted = (int*)take(4 * (size_t)(N + 1));
  w.wrap = (int*)take(12 * (size_t)(N + 1));
  w.bin_start = (int*)take(4 * (size_t)(B + 1));
  w.counts = (int*)take(4 * (size_t)(N + 1));
  w.total = (int64_t*)take(16);
  w.sort_tmp_bytes = nbl_sort_tmp_bytes(N);
  w.sort_tmp = take(w.sort_tmp_bytes);
  return off;
}

extern "C" int64_t spk_nbl_workspace_bytes(int64_t n_atoms, int64_t n_sys) {
  if (n_atoms < 0 || n_sys < 0) return -1;
  NblWs w;
  return (int64_t)nbl_carve(w, nullptr, n_atoms, n_sys);
}

// first atom of every system from the (ascending) molecule index; systems without atoms get the
// start of the next non-empty one
__global__ void k_nbl_atom0(const int64_t* __restrict__ idx_m, int64_t N, int64_t n_sys, int* __restrict__ atom0,
                            int64_t* __restrict__ total) {
  for (int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; a < N; a += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx_m ? idx_m[a] : 0;
    if (m < 0 || m >= n_sys) { atomicOr((unsigned long long*)&total[1], 2ull); continue; }
    if (a > 0) {
      const int64_t mp = idx_m ? idx_m[a - 1] : 0;
      if (mp > m) atomicOr((unsigned long long*)&total[1], 4ull);   // not ascending
      if (mp != m) atom0[m] = (int)a;
    } else {
      atom0[m] = 0;
    }
  }
}
__global__ void k_nbl_atom0_fix(int64_t N, int64_t n_sys, int* __restrict__ atom0) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    atom0[n_sys] = (int)N;
    for (int64_t m = n_sys - 1; m >= 0; --m)
      if (atom0[m] < 0) atom0[m] = atom0[m + 1];
  }
}

__device__ __forceinline__ void nbl_frac(const NblSys& s, float x, float y, float z, float (&f)[3]) {
  f[0] = x * s.inv[0] + y * s.inv[3] + z * s.inv[6];
  f[1] = x * s.inv[1] + y * s.inv[4] + z * s.inv[7];
  f[2] = x * s.inv[2] + y * s.inv[5] + z * s.inv[8];
}

__global__ __launch_bounds__(256) void k_nbl_desc(const float* __restrict__ R, const float* __restrict__ cell,
                                                  const unsigned char* __restrict__ pbc, const int* __restrict__ atom0,
                                                  float cutoff, NblSys* __restrict__ sys, int64_t* __restrict__ total) {
  __shared__ NblSys s;
  __shared__ float red[2][3][256];
  const int m = blockIdx.x;
  const int a0 = atom0[m], a1 = atom0[m + 1];
  if (threadIdx.x == 0) {
    s.atom0 = a0; s.natoms = a1 - a0; s.bad = 0;
    int any = 0;
    for (int k = 0; k < 3; ++k) { s.pbc[k] = (pbc && pbc[3 * m + k]) ? 1 : 0; any |= s.pbc[k]; }
    float c[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (any && cell) for (int k = 0; k < 9; ++k) c[k] = cell[9 * (int64_t)m + k];
    const float det = c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]);
    const float scale = fabsf(c[0]) + fabsf(c[1]) + fabsf(c[2]) + fabsf(c[3]) + fabsf(c[4]) + fabsf(c[5]) + fabsf(c[6]) + fabsf(c[7]) + fabsf(c[8]);
    if (!(fabsf(det) > 1e-9f * scale * scale * scale) || (any && !cell)) {
      s.bad = 1;   // periodic system without a usable cell
      for (int k = 0; k < 9; ++k) c[k] = (k % 4 == 0) ? 1.f : 0.f;
    }
    const float id = 1.0f / (c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]));
    for (int k = 0; k < 9; ++k) s.cell[k] = c[k];
    s.inv[0] = (c[4] * c[8] - c[5] * c[7]) * id; s.inv[1] = (c[2] * c[7] - c[1] * c[8]) * id; s.inv[2] = (c[1] * c[5] - c[2] * c[4]) * id;
    s.inv[3] = (c[5] * c[6] - c[3] * c[8]) * id; s.inv[4] = (c[0] * c[8] - c[2] * c[6]) * id; s.inv[5] = (c[2] * c[3] - c[0] * c[5]) * id;
    s.inv[6] = (c[3] * c[7] - c[4] * c[6]) * id; s.inv[7] = (c[1] * c[6] - c[0] * c[7]) * id; s.inv[8] = (c[0] * c[4] - c[1] * c[3]) * id;
  }
  __syncthreads();
  // bounding box in fractional coordinates (used along the non-periodic axes)
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int a = a0 + threadIdx.x; a < a1; a += blockDim.x) {
    float f[3];
    nbl_frac(s, R[3 * (int64_t)a], R[3 * (int64_t)a + 1], R[3 * (int64_t)a + 2], f);
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], f[k]); hi[k] = fmaxf(hi[k], f[k]); }
  }
  for (int k = 0; k < 3; ++k) { red[0][k][threadIdx.x] = lo[k]; red[1][k][threadIdx.x] = hi[k]; }
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int k = 0; k < 3; ++k) {
        red[0][k][threadIdx.x] = fminf(red[0][k][threadIdx.x], red[0][k][threadIdx.x + st]);
        red[1][k][threadIdx.x] = fmaxf(red[1][k][threadIdx.x], red[1][k][threadIdx.x + st]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float H[3];
    for (int k = 0; k < 3; ++k) {
      // perpendicular height of the cell along axis k = 1 / |column k of inv|
      const float il = sqrtf(s.inv[k] * s.inv[k] + s.inv[3 + k] * s.inv[3 + k] + s.inv[6 + k] * s.inv[6 + k]);
      const float h = 1.0f / il;
      if (s.pbc[k]) { s.fmin[k] = 0.f; s.fext[k] = 1.f; H[k] = h; }
      else {
        const float l = s.natoms > 0 ? red[0][k][0] : 0.f, u = s.natoms > 0 ? red[1][k][0] : 0.f;
        s.fmin[k] = l; s.fext[k] = fmaxf(u - l, 0.f); H[k] = s.fext[k] * h;
      }
      int nb = (int)floorf(H[k] / cutoff);
      s.nb[k] = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    }
    // never more bins than atoms (sparse boxes): shrink all axes by a common factor, then trim
    const long cap = s.natoms > 1 ? s.natoms : 1;
    long prod = (long)s.nb[0] * s.nb[1] * s.nb[2];
    if (prod > cap) {
      const float f = cbrtf((float)cap / (float)prod);
      for (int k = 0; k < 3; ++k) { int nb = (int)floorf(s.nb[k] * f); s.nb[k] = nb < 1 ? 1 : nb; }
      prod = (long)s.nb[0] * s.nb[1] * s.nb[2];
      while (prod > cap) {
        int kmax = s.nb[0] >= s.nb[1] ? (s.nb[0] >= s.nb[2] ? 0 : 2) : (s.nb[1] >= s.nb[2] ? 1 : 2);
        s.nb[kmax] -= 1;
        prod = (long)s.nb[0] * s.nb[1] * s.nb[2];
      }
    }
    for (int k = 0; k < 3; ++k) {
      const float hb = H[k] / (float)s.nb[k];
      int reach = hb > 0.f ? (int)ceilf(cutoff / hb * (1.0f + 1e-5f)) : 0;
      if (!s.pbc[k] && reach > s.nb[k] - 1) reach = s.nb[k] - 1;
      if (reach > 4096) { reach = 4096; s.bad = 1; }   // cutoff thousands of times the cell height
      s.reach[k] = reach;
    }
    if (s.bad) atomicOr((unsigned long long*)&total[1], 1ull);
    s.bin0 = (int)prod;   // number of bins for now; k_nbl_binoffsets turns it into the offset
    sys[m] = s;
  }
}

// exclusive scan of the per-system bin counts (left in bin0 by k_nbl_desc) -> bin offsets; one block
__global__ __launch_bounds__(256) void k_nbl_binoffsets(NblSys* __restrict__ sys, int64_t n_sys) {
  __shared__ int part[256];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n_sys; base += 256) {
    const int64_t m = base + threadIdx.x;
    const int v = m < n_sys ? sys[m].bin0 : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int st = 1; st < 256; st <<= 1) {
      const int add = (int)threadIdx.x >= st ? part[threadIdx.x - st] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (m < n_sys) sys[m].bin0 = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 255) carry += part[255];
    __syncthreads();
  }
}

__global__ void k_nbl_bin(const float* __restrict__ R, const int64_t* __restrict__ idx_m, const NblSys* __restrict__ sys,
                          int64_t N, int* __restrict__ key, int* __restrict__ ids, int* __restrict__ wrap) {
  for (int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; a < N; a += (int64_t)gridDim.x * blockDim.x) {
    const NblSys& s = sys[idx_m ? idx_m[a] : 0];
    float f[3];
    nbl_frac(s, R[3 * a], R[3 * a + 1], R[3 * a + 2], f);
    int b[3];
    for (int k = 0; k < 3; ++k) {
      int w = 0;
      float g;
      if (s.pbc[k]) {
        const float fl = floorf(f[k]);
        w = (int)fl;
        g = f[k] - fl;
        if (g >= 1.0f) { g = 0.f; w += 1; }
      } else {
        g = s.fext[k] > 0.f ? (f[k] - s.fmin[k]) / s.fext[k] : 0.f;
      }
      int bb = (int)(g * (float)s.nb[k]);
      b[k] = bb < 0 ? 0 : (bb > s.nb[k] - 1 ? s.nb[k] - 1 : bb);
      wrap[3 * a + k] = w;
    }
    key[a] = s.bin0 + (b[0] * s.nb[1] + b[1]) * s.nb[2] + b[2];
    ids[a] = (int)a;
  }
}

__global__ void k_nbl_bounds(const int* __restrict__ key_sorted, int64_t N, int64_t B, int* __restrict__ bin_start) {
  for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b <= B; b += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = N;   // first position with key >= b
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (key_sorted[mid] < (int)b) lo = mid + 1; else hi = mid;
    }
    bin_start[b] = (int)lo;
  }
}

__device__ __forceinline__ int nbl_floordiv(int a, int n) { return (a >= 0) ? a / n : -((-a + n - 1) / n); }

struct NblOut {
  int64_t* idx_i; int64_t* idx_j; int32_t* shifts; float* offsets; const int32_t* rowptr;
};

template <bool FILL>
__global__ __launch_bounds__(256) void k_nbl_pairs(const float* __restrict__ R, const int64_t* __restrict__ idx_m,
                                                   const NblSys* __restrict__ sys, const int* __restrict__ key,
                                                   const int* __restrict__ ids_sorted, const int* __restrict__ wrap,
                                                   const int* __restrict__ bin_start, int64_t N, float cutoff,
                                                   int* __restrict__ counts, NblOut out) {
  const int lane = threadIdx.x & 63;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < N; i += nwaves) {
    const NblSys& s = sys[idx_m ? idx_m[i] : 0];
    const int local = key[i] - s.bin0;
    const int bz = local % s.nb[2], by = (local / s.nb[2]) % s.nb[1], bx = local / (s.nb[2] * s.nb[1]);
    const float xi = R[3 * i], yi = R[3 * i + 1], zi = R[3 * i + 2];
    const int wix = wrap[3 * i], wiy = wrap[3 * i + 1], wiz = wrap[3 * i + 2];
    int64_t base = FILL ? (int64_t)out.rowptr[i] : 0;
    int count = 0;
    for (int dx = -s.reach[0]; dx <= s.reach[0]; ++dx) {
      int cx = bx + dx, sx = 0;
      if (s.pbc[0]) { sx = nbl_floordiv(cx, s.nb[0]); cx -= sx * s.nb[0]; } else if (cx < 0 || cx >= s.nb[0]) continue;
      for (int dy = -s.reach[1]; dy <= s.reach[1]; ++dy) {
        int cy = by + dy, sy = 0;
        if (s.pbc[1]) { sy = nbl_floordiv(cy, s.nb[1]); cy -= sy * s.nb[1]; } else if (cy < 0 || cy >= s.nb[1]) continue;
        for (int dz = -s.reach[2]; dz <= s.reach[2]; ++dz) {
          int cz = bz + dz, sz = 0;
          if (s.pbc[2]) { sz = nbl_floordiv(cz, s.nb[2]); cz -= sz * s.nb[2]; } else if (cz < 0 || cz >= s.nb[2]) continue;
          const int bin = s.bin0 + (cx * s.nb[1] + cy) * s.nb[2] + cz;
          const int p0 = bin_start[bin], p1 = bin_start[bin + 1];
          for (int pb = p0; pb < p1; pb += 64) {
            const int p = pb + lane;
            bool hit = false;
            int j = 0, Sx = 0, Sy = 0, Sz = 0;
            float ox = 0.f, oy = 0.f, oz = 0.f;
            if (p < p1) {
              j = ids_sorted[p];
              Sx = sx - wrap[3 * (int64_t)j] + wix; Sy = sy - wrap[3 * (int64_t)j + 1] + wiy; Sz = sz - wrap[3 * (int64_t)j + 2] + wiz;
              // offsets = S . cell (transform/neighborlist.py:457); the reversed pair evaluates the exact negation
              ox = fmaf((float)Sz, s.cell[6], fmaf((float)Sy, s.cell[3], (float)Sx * s.cell[0]));
              oy = fmaf((float)Sz, s.cell[7], fmaf((float)Sy, s.cell[4], (float)Sx * s.cell[1]));
              oz = fmaf((float)Sz, s.cell[8], fmaf((float)Sy, s.cell[5], (float)Sx * s.cell[2]));
              const float x = (R[3 * (int64_t)j] - xi) + ox, y = (R[3 * (int64_t)j + 1] - yi) + oy, z = (R[3 * (int64_t)j + 2] - zi) + oz;
              const float d = sqrtf(x * x + y * y + z * z);
              hit = (d < cutoff) && !((int64_t)j == i && Sx == 0 && Sy == 0 && Sz == 0);
            }
            const uint64_t mask = __ballot(hit);
            if (FILL && hit) {
              const int64_t e = base + count + __popcll(mask & lt);
              out.idx_i[e] = i; out.idx_j[e] = j;
              if (out.shifts) { out.shifts[3 * e] = Sx; out.shifts[3 * e + 1] = Sy; out.shifts[3 * e + 2] = Sz; }
              out.offsets[3 * e] = ox; out.offsets[3 * e + 1] = oy; out.offsets[3 * e + 2] = oz;
            }
            count += __popcll(mask);
          }
        }
      }
    }
    if (!FILL && lane == 0) counts[i] = count;
  }
}

// exclusive scan of counts [n] -> rowptr [n + 1] (one block; n is the number of atoms)
__global__ __launch_bounds__(1024) void k_nbl_scan(const int* __restrict__ counts, int64_t n, int32_t* __restrict__ rowptr,
                                                   int64_t* __restrict__ total) {
  __shared__ long long part[1024];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t a = base + threadIdx.x;
    const long long v = a < n ? counts[a] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int st = 1; st < 1024; st <<= 1) {
      const long long add = (int)threadIdx.x >= st ? part[threadIdx.x - st] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    const long long excl = carry + part[threadIdx.x] - v;
    if (a < n) rowptr[a] = (int32_t)excl;
    if (a < n && excl + v > 2147483647LL) atomicOr((unsigned long long*)&total[1], 8ull);
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) { rowptr[n] = (int32_t)carry; total[0] = carry; }
}

static int nbl_prepare(NblWs& w, const float* R, const int64_t* idx_m, const float* cell, const unsigned char* pbc,
                       int64_t N, int64_t n_sys, float cutoff, hipStream_t stream) {
  SPK_HIP_TRY(hipMemsetAsync(w.total, 0, 16, stream));
  SPK_HIP_TRY(hipMemsetAsync(w.atom0, 0xff, 4 * (size_t)(n_sys + 1), stream));
  hipLaunchKernelGGL(k_nbl_atom0, dim3(spk_grid_for(N, 256, spk_num_cus() * 8)), dim3(256), 0, stream, idx_m, N, n_sys, w.atom0, w.total);
  hipLaunchKernelGGL(k_nbl_atom0_fix, dim3(1), dim3(64), 0, stream, N, n_sys, w.atom0);
  hipLaunchKernelGGL(k_nbl_desc, dim3((unsigned)n_sys), dim3(256), 0, stream, R, cell, pbc, w.atom0, cutoff, w.sys, w.total);
  hipLaunchKernelGGL(k_nbl_binoffsets, dim3(1), dim3(256), 0, stream, w.sys, n_sys);
  hipLaunchKernelGGL(k_nbl_bin, dim3(spk_grid_for(N, 256, spk_num_cus() * 8)), dim3(256), 0, stream, R, idx_m, w.sys, N, w.key, w.ids, w.wrap);
  SPK_LAUNCH_CHECK();
  size_t tmp = w.sort_tmp_bytes;
  // bin ids are < N + n_sys: sort only the bits that can be set
  int bits = 1;
  while (((int64_t)1 << bits) < w.max_bins + 1 && bits < 31) ++bits;
  SPK_HIP_TRY(rocprim::radix_sort_pairs(w.sort_tmp, tmp, w.key, w.key_sorted, w.ids, w.ids_sorted, (size_t)N, 0, bits, stream));
  hipLaunchKernelGGL(k_nbl_bounds, dim3(spk_grid_for(w.max_bins + 1, 256, spk_num_cus() * 8)), dim3(256), 0, stream, w.key_sorted, N, w.max_bins, w.bin_start);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

static int nbl_check_args(const char* who, const float* R, int64_t N, int64_t n_sys, float cutoff, void* ws) {
  SPK_CHECK_ARG(N >= 0 && n_sys >= 1 && N < 2147483647LL, "%s: bad sizes n_atoms=%lld n_sys=%lld", who, (long long)N, (long long)n_sys);
  SPK_CHECK_ARG(cutoff > 0.f, "%s: cutoff must be positive", who);
  SPK_CHECK_ARG(ws != nullptr && (N == 0 || R != nullptr), "%s: null pointer", who);
  return SPK_OK;
}

extern "C" int spk_nbl_count_f32(const float* R, const int64_t* idx_m, const float* cell, const uint8_t* pbc,
                                 int64_t n_atoms, int64_t n_sys, float cutoff, void* workspace, int32_t* rowptr,
                                 int64_t* n_edges_host, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = nbl_check_args("spk_nbl_count_f32", R, n_atoms, n_sys, cutoff, workspace);
  if (rc != SPK_OK) return rc;
  SPK_CHECK_ARG(rowptr && n_edges_host, "spk_nbl_count_f32: null output");
  *n_edges_host = 0;
  if (n_atoms == 0) { SPK_HIP_TRY(hipMemsetAsync(rowptr, 0, 4, stream)); return SPK_OK; }
  SpkProfScope prof("nbl_count", stream);
  NblWs w;
  nbl_carve(w, workspace, n_atoms, n_sys);
  rc = nbl_prepare(w, R, idx_m, cell, pbc, n_atoms, n_sys, cutoff, stream);
  if (rc != SPK_OK) return rc;
  NblOut out = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL((k_nbl_pairs<false>), dim3(spk_grid_for(n_atoms * 64, 256, spk_num_cus() * 32)), dim3(256), 0, stream,
                     R, idx_m, w.sys, w.key, w.ids_sorted, w.wrap, w.bin_start, n_atoms, cutoff, w.counts, out);
  hipLaunchKernelGGL(k_nbl_scan, dim3(1), dim3(1024), 0, stream, w.counts, n_atoms, rowptr, w.total);
  SPK_LAUNCH_CHECK();
  int64_t host[2] = {0, 0};
  SPK_HIP_TRY(hipMemcpyAsync(host, w.total, 16, hipMemcpyDeviceToHost, stream));
  SPK_HIP_TRY(hipStreamSynchronize(stream));
  SPK_CHECK_ARG(!(host[1] & 2), "spk_nbl_count_f32: idx_m entry outside [0, n_sys)");
  SPK_CHECK_ARG(!(host[1] & 4), "spk_nbl_count_f32: idx_m must be ascending (atoms of a system contiguous)");
  SPK_CHECK_ARG(!(host[1] & 1), "spk_nbl_count_f32: periodic system with a singular cell (or a cutoff thousands of cell heights)");
  SPK_CHECK_ARG(!(host[1] & 8), "spk_nbl_count_f32: more than 2^31 - 1 pairs");
  *n_edges_host = host[0];
  return SPK_OK;
}

extern "C" int spk_nbl_fill_f32(const float* R, const int64_t* idx_m, int64_t n_atoms, int64_t n_sys, float cutoff,
                                const void* workspace, const int32_t* rowptr, int64_t n_edges, int64_t* idx_i,
                                int64_t* idx_j, int32_t* shifts, float* offsets, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = nbl_check_args("spk_nbl_fill_f32", R, n_atoms, n_sys, cutoff, (void*)workspace);
  if (rc != SPK_OK) return rc;
  if (n_atoms == 0 || n_edges == 0) return SPK_OK;
  SPK_CHECK_ARG(rowptr && idx_i && idx_j && offsets, "spk_nbl_fill_f32: null output");
  SpkProfScope prof("nbl_fill", stream);
  NblWs w;
  nbl_carve(w, (void*)workspace, n_atoms, n_sys);
  NblOut out = {idx_i, idx_j, shifts, offsets, rowptr};
  hipLaunchKernelGGL((k_nbl_pairs<true>), dim3(spk_grid_for(n_atoms * 64, 256, spk_num_cus() * 32)), dim3(256), 0, stream,
                     R, idx_m, w.sys, w.key, w.ids_sorted, w.wrap, w.bin_start, n_atoms, cutoff, w.counts, out);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
