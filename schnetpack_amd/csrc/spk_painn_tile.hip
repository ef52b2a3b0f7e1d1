// PaiNN message forward (representation/painn.py:31-67) with the filter GEMM on the matrix cores: 32-edge tiles.
//
// The row kernel of spk_painn.hip recomputes the filter slice Phi_e = (phi(d_e) Wf^T + bf) fcut(d_e) of every edge
// with 3 x n_rbf packed VALU FMAs per channel pair.  Here a wavefront owns a tile of 32 consecutive DIRECTED edges of
// the sorted list and evaluates the filter as a GEMM on v_mfma_f32_32x32x2_f32
//   Phi [32 edges x 3F] = A [32 x KP] B [KP x 3F],   A[e][k] = fcut(d_e) phi_k(d_e)  (k < K),  A[e][K] = fcut(d_e)  (bias column),
//   B = packed (Wf | bf) staged once per workgroup in LDS
// so the cutoff and the bias are part of the GEMM and the VALU only does the message algebra.  Operands are chosen
// such that the accumulator has rows = edges, columns = channels: a lane owns one channel, its 16 registers are 16
// edges, neighbour rows c[j], mu[j] are gathered as coalesced 128-byte segments, and the per-centre-atom sums are
// per-lane running sums over registers flushed with one float atomic per run (idx_i is sorted; the outputs are
// pre-initialised with q / mu).  Matrix row m of the tile is edge slot 16 (m>>2 & 1) + (m & 3) + 4 (m >> 3), which
// makes the 16 registers of a half-wave 16 CONSECUTIVE edges: a row of the list is split over as few runs as possible.
//
// Measurements and the dispatch rules that follow from them: HISTORY.md section 4.3, profiles/r01_painn_tile_experiment.json.
#include "spk_painn_msg.h"
#ifndef SPK_RT_HOLLOW
#define SPK_RT_HOLLOW 0     // timing aid of the row-tile backward: 1 = no channel-block loop, 2 = no gathers, 3 = no GEMMs (results are wrong)
#endif
#include "spk_split.h"
#include "spk_filter_split.h"

#define SPK_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)

namespace {

struct __attribute__((aligned(16))) TileRec { int i; int j; float ux; float uy; float uz; float invd; float r0; float r1; };

// packed (Wf | bf): P[((t * KPB + ug) * 64 + lane) * 4 + v] = W'[32 t + (lane & 31)][8 ug + 4 (lane >> 5) + v],
// W'[row][k] = wf[row][k] (k < K), bf[row] (k == K), 0 beyond
template <int NSLOT>
__device__ __forceinline__ void stage_filter(float* dst, const float* __restrict__ wf, const float* __restrict__ bf, int K, int KPB) {
  for (int s = threadIdx.x; s < NSLOT; s += 256) {
    const int lane = s & 63;
    const int ug = (s >> 6) % KPB;
    const int t = (s >> 6) / KPB;
    const int row = 32 * t + (lane & 31);
    const int k0 = 8 * ug + 4 * (lane >> 5);
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = k0 + q;
      v[q] = k < K ? wf[(int64_t)row * K + k] : (k == K ? bf[row] : 0.f);
    }
    *(f32x4*)(dst + (int64_t)s * 4) = v;
  }
}

template <int KPB>
__device__ __forceinline__ f32x16 tile_gemm(const float* __restrict__ sW, int t, const float (&A)[KPB][4], int lane) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const f32x4* wp = (const f32x4*)sW + (int64_t)t * KPB * 64 + lane;
#pragma unroll
  for (int u = 0; u < KPB; ++u) {
    const f32x4 wq = wp[u * 64];
    acc = SPK_MFMA(A[u][0], wq.x, acc);
    acc = SPK_MFMA(A[u][1], wq.y, acc);
    acc = SPK_MFMA(A[u][2], wq.z, acc);
    acc = SPK_MFMA(A[u][3], wq.w, acc);
  }
  return acc;
}

// ---- split-precision form of the filter GEMM (round 6; spk_split.h): the same product on v_mfma_f32_32x32x16_f16 with (high, low) fp16 operand
// pairs -- 6 instructions of 32 cycles per 32 x 32 block instead of 4 KPB of 64.  The contraction index is laid out as in spk_filter_split.h
// (ml_w1_k: k-step 0 = columns 8 hi + e; k-step 1 = 16 + 4 hi + e for e < 4 and, KPB = 4 only, 24 + 4 hi + (e - 4)): every lane
// evaluates 12 (16) radial functions as before.  Images: [NB * 64 slots] h16x8 of k-step 0, then h16x4 (KPB = 3) / h16x8 (KPB = 4) of k-step 1;
// high image then low image, together exactly the bytes of the fp32 image.
template <int KPB, int NB>
struct PtImage {
  static constexpr int STEP1 = NB * 64 * 16;                                   // byte offset of k-step 1 inside an image
  static constexpr int BYTES = STEP1 + NB * 64 * (KPB == 4 ? 16 : 8);          // per image (high or low)
};
template <int KPB, int NB>
__device__ __forceinline__ void stage_filter_split(char* ih, char* il, const float* __restrict__ wf, const float* __restrict__ bf, int K) {
  for (int s = threadIdx.x; s < NB * 64; s += 256) {
    const int lane = s & 63, t = s >> 6, hi = lane >> 5;
    const int row = 32 * t + (lane & 31);
    const float* src = wf + (int64_t)row * K;
    const float bias = bf[row];
    float x[8];
    h16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int k = ml_w1_k(0, hi, e); x[e] = k < K ? src[k] : (k == K ? bias : 0.f); }
    sp_split8(x, h, l);
    ((h16x8*)ih)[s] = h; ((h16x8*)il)[s] = l;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int k = ml_w1_k(1, hi, e); x[e] = (KPB == 4 || e < 4) ? (k < K ? src[k] : (k == K ? bias : 0.f)) : 0.f; }
    sp_split8(x, h, l);
    if (KPB == 4) { ((h16x8*)(ih + PtImage<KPB, NB>::STEP1))[s] = h; ((h16x8*)(il + PtImage<KPB, NB>::STEP1))[s] = l; }
    else {
      ((h16x4*)(ih + PtImage<KPB, NB>::STEP1))[s] = h16x4{h[0], h[1], h[2], h[3]};
      ((h16x4*)(il + PtImage<KPB, NB>::STEP1))[s] = h16x4{l[0], l[1], l[2], l[3]};
    }
  }
}
// the A operands of one tile: value (fc phi_k | fc) and, DERIV, slope (fc phi_k' + fc' phi_k | fc') of this lane's edge slot
template <int KPB, bool DERIV>
__device__ __forceinline__ void tile_operands_split(const RadialDev& rb, int K, int hi, float d, float fc, float dfc, h16x8 (&vh)[2], h16x8 (&vl)[2],
                                                    h16x8 (&dh)[2], h16x8 (&dl)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float v[8], dv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = 0.f; dv[e] = 0.f;
      if (s == 0 || KPB == 4 || e < 4) {
        const int k = ml_w1_k(s, hi, e);
        float p, dp;
        spk_rbf_eval_fast(rb, k, d, p, dp);          // 0 for k >= K
        if (k == K) { p = 1.0f; dp = 0.f; }
        v[e] = fc * p;
        dv[e] = fc * dp + dfc * p;
      }
    }
    sp_split8(v, vh[s], vl[s]);
    if (DERIV) sp_split8(dv, dh[s], dl[s]);
  }
}
template <int KPB, int NB>
__device__ __forceinline__ f32x16 tile_gemm_split(const char* __restrict__ ih, const char* __restrict__ il, int t, const h16x8 (&ah)[2], const h16x8 (&al)[2], int lane) {
  f32x16 acc, cross;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; cross[r] = 0.f; }
  const int slot = t * 64 + lane;
  {
    const h16x8 wh = ((const h16x8*)ih)[slot], wl = ((const h16x8*)il)[slot];
    SP_STEP(ah[0], al[0], wh, wl, acc, cross);
  }
  {
    h16x8 wh, wl;
    if (KPB == 4) { wh = ((const h16x8*)(ih + PtImage<KPB, NB>::STEP1))[slot]; wl = ((const h16x8*)(il + PtImage<KPB, NB>::STEP1))[slot]; }
    else {
      const h16x4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      wh = sp_cat(((const h16x4*)(ih + PtImage<KPB, NB>::STEP1))[slot], z);
      wl = sp_cat(((const h16x4*)(il + PtImage<KPB, NB>::STEP1))[slot], z);
    }
    SP_STEP(ah[1], al[1], wh, wl, acc, cross);
  }
  SP_FOLD(acc, cross);
  return acc;
}

// load from a wave-uniform base and a 32-bit BYTE offset per lane: global_load_dword v, v_off, s[base] -- with element indices the compiler forms a
// 64-bit address per load (v_lshlrev_b64 + v_lshl_add_u64: half of the vector instructions of the row-tile geometry pass were address arithmetic)
__device__ __forceinline__ float ld_off(const float* __restrict__ base, unsigned byte_off) { return *(const float*)((const char*)base + byte_off); }

// sum over the 4 lanes of a quad with DPP (quad_perm [1,0,3,2] then [2,3,0,1]): VALU modifiers, no LDS crossbar round trip
__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  return v;
}

// edge slot handled by lane el (as matrix row el) -- see the file comment
__device__ __forceinline__ int slot_of_row(int el) { return 16 * ((el >> 2) & 1) + (el & 3) + 4 * (el >> 3); }

// MU0: mu == 0 (first interaction): the mu part of the filter and the mu rows are skipped
// MINW: waves per SIMD the register allocation is held to (2: up to 256 VGPRs; 4: up to 128 -- tuning experiment SPK_TILE_WAVES=4).
// xcd_map != 0: workgroup w (XCD w % 8) walks a CONTIGUOUS eighth of the tiles instead of every gridDim-th tile, so the rows its
// neighbours on the same XCD gather are the ones it gathers (default; SPK_XCD_WALK=0 switches it off)
// SPLIT: the filter GEMM on the f16 matrix instructions with split operands (default where KPB is 3 or 4; spk_set_split / SPK_SPLIT=0: fp32)
template <int F, int KPB, bool MU0, int MINW = 2, bool SPLIT = false>
__global__ __launch_bounds__(256, MINW) void k_painn_msg_tile(MsgArgs a, int ntiles, int xcd_map) {
  constexpr int NT = F / 32;          // channel blocks per part
  constexpr int NB = 3 * NT;          // column blocks of the filter GEMM
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                                        // NB * KPB * 256 floats
  TileRec* sE = (TileRec*)(sW + NB * KPB * 256);           // 4 waves x 32 records
  int* sCnt = (int*)(sE + 4 * 32);
  const int K = a.rb.n_rbf;
  constexpr int KPS = (KPB == 3 || KPB == 4) ? KPB : 4;        // (the split form is instantiated for KPB 3 / 4 only)
  const char* sWh = (const char*)smem;
  const char* sWl = sWh + PtImage<KPS, NB>::BYTES;

  if (SPLIT) stage_filter_split<KPS, NB>((char*)smem, (char*)smem + PtImage<KPS, NB>::BYTES, a.wf, a.bf, K);
  else stage_filter<NB * KPB * 64>(sW, a.wf, a.bf, K, KPB);
  if (threadIdx.x == 0) sCnt[0] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  TileRec* myE = sE + wv * 32;
  const int slot = slot_of_row(el);
  constexpr unsigned F3 = 3u * F;

  const int per_xcd = (ntiles + 7) / 8, wg_per_xcd = ((int)gridDim.x + 7) / 8;
  while (true) {
    int nidx = 0;
    if (lane == 0) nidx = atomicAdd(&sCnt[0], 1);
    nidx = __builtin_amdgcn_readfirstlane(nidx);
    int tile = (int)blockIdx.x + nidx * (int)gridDim.x;
    if (xcd_map) {
      const int tl = ((int)blockIdx.x >> 3) + nidx * wg_per_xcd;
      tile = tl < per_xcd ? ((int)blockIdx.x & 7) * per_xcd + tl : ntiles;
    }
    if (tile >= ntiles) break;

    // ---- geometry of this lane's edge slot (lanes 32..63 mirror lanes 0..31)
    const int64_t e_first = (int64_t)tile * 32;
    const int nvalid = (a.E - e_first) < 32 ? (int)(a.E - e_first) : 32;
    const bool valid = slot < nvalid;
    const int64_t e = e_first + (valid ? slot : (nvalid - 1));
    const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
    const int ci = (int)a.idx_i[e], cj = (int)a.idx_j[e];
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    const float invd = 1.0f / d;
    float fc, dfc;
    spk_cutoff_eval_fast(a.rb.cutoff, d, fc, dfc);
    if (!valid) fc = 0.f;
    if (hi == 0) {
      TileRec rec; rec.i = ci; rec.j = cj; rec.ux = rx * invd; rec.uy = ry * invd; rec.uz = rz * invd; rec.invd = invd; rec.r0 = 0.f; rec.r1 = 0.f;
      myE[slot] = rec;
    }
    // A operands: (fc phi_k | fc)
    float Av[KPB][4];
    h16x8 Avh[2], Avl[2], Adh_[2], Adl_[2];
    if (SPLIT) tile_operands_split<KPS, false>(a.rb, K, hi, d, fc, 0.f, Avh, Avl, Adh_, Adl_);
    else {
#pragma unroll
    for (int u = 0; u < KPB; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int k = 8 * u + 4 * hi + v;
        float p, dp;
        spk_rbf_eval_fast(a.rb, k, d, p, dp);          // 0 for k >= K
        if (k == K) p = 1.0f;
        Av[u][v] = fc * p;
      }
    }
    auto gemm = [&](int t) -> f32x16 {
#if SPK_RT_HOLLOW == 3
      { f32x16 o; for (int r = 0; r < 16; ++r) o[r] = fc + (float)t; return o; }
#endif
      if (SPLIT) return tile_gemm_split<KPS, NB>(sWh, sWl, t, Avh, Avl, lane);
      return tile_gemm<KPB>(sW, t, Av, lane);
    };
    spk_wave_lds_sync();   // records visible to the whole wave
    // bit r set <=> the centre atom changes after register r of this half (or the half ends)
    unsigned runmask = 0x8000u;
    {
      int prev = myE[16 * hi].i;
#pragma unroll
      for (int r = 1; r < 16; ++r) {
        const int cur = myE[16 * hi + r].i;
        if (cur != prev) runmask |= 1u << (r - 1);
        prev = cur;
      }
    }

    {
#pragma unroll 1
      for (int cb = 0; cb < NT; ++cb) {
        const unsigned c0 = 32u * cb + el;
#if SPK_RT_HOLLOW == 1
        continue;
#endif
        // ---- scalar part: dq_i = sum Phi_q c_q[j]
        {
          float cq[16];
#pragma unroll
#if SPK_RT_HOLLOW == 2
          for (int r = 0; r < 16; ++r) cq[r] = (float)(myE[16 * hi + r].j & 7);
#else
          for (int r = 0; r < 16; ++r) cq[r] = ld_off(a.c, ((unsigned)myE[16 * hi + r].j * F3 + c0) * 4u);
#endif
          const f32x16 Pq = gemm(cb);
          float acc = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc = fmaf(Pq[r], cq[r], acc);
#if SPK_RT_HOLLOW != 4
            if ((runmask >> r) & 1u) { unsafeAtomicAdd(a.q_out + ((unsigned)myE[16 * hi + r].i * F + c0), acc); acc = 0.f; }
#endif
          }
        }
        // ---- vector part: dmu_i = sum (Phi_R c_R[j]) u + (Phi_mu c_mu[j]) mu[j]
        const f32x16 PR = gemm(NT + cb);
        f32x16 Pm = PR;
        if (!MU0) Pm = gemm(2 * NT + cb);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < 16; g0 += 8) {
          float cR[8], cm[8], m0[8], m1[8], m2[8];
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) {
            const unsigned bj = ((unsigned)myE[16 * hi + g0 + rr].j * F3 + c0) * 4u;       // byte offset (< 2^32: spk_painn_msg_tile_ok)
#if SPK_RT_HOLLOW == 2
            cR[rr] = (float)(bj & 12);
            if (false) {
#else
            cR[rr] = ld_off(a.c, bj + 4 * F);
            if (!MU0) {
#endif
              cm[rr] = ld_off(a.c, bj + 8 * F); m0[rr] = ld_off(a.mu, bj); m1[rr] = ld_off(a.mu, bj + 4 * F); m2[rr] = ld_off(a.mu, bj + 8 * F); }
            else { cm[rr] = 0.f; m0[rr] = 0.f; m1[rr] = 0.f; m2[rr] = 0.f; }
          }
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) {
            const int r = g0 + rr;
            const TileRec er = myE[16 * hi + r];
            const float mR = PR[r] * cR[rr], mm = MU0 ? 0.f : Pm[r] * cm[rr];
            a0 = fmaf(mR, er.ux, MU0 ? a0 : fmaf(mm, m0[rr], a0));
            a1 = fmaf(mR, er.uy, MU0 ? a1 : fmaf(mm, m1[rr], a1));
            a2 = fmaf(mR, er.uz, MU0 ? a2 : fmaf(mm, m2[rr], a2));
#if SPK_RT_HOLLOW != 4
            if ((runmask >> r) & 1u) {
              float* dst = a.mu_out + ((unsigned)er.i * F3 + c0);
              unsafeAtomicAdd(dst, a0); unsafeAtomicAdd(dst + F, a1); unsafeAtomicAdd(dst + 2 * F, a2);
              a0 = 0.f; a1 = 0.f; a2 = 0.f;
            }
#endif
          }
        }
      }
    }
#if SPK_RT_HOLLOW == 4
    asm volatile("" :: "v"(d));
#endif
    spk_wave_lds_sync();   // records may be rewritten by the next tile
  }
}

// ---- backward (first order; sorted + symmetric list as for the row kernel: Phi_e = Phi_rev(e), u_rev(e) = -u_e) ----------
// Two GEMMs per column block: value A = (fc phi_k | fc) and derivative A' = (fc phi_k' + fc' phi_k | fc') -> F = Phi fc and dF/dd.
// The three parts (q, R, mu) are processed one after the other so that only one (F, dF) accumulator pair is live; the
// per-edge sums over channels (dd, t_x, t_y, t_z -> geometry gradient) stay in 64 registers over the channel blocks and
// are reduced over the lanes at the end of the tile (quad sums by DPP, 8 partials per edge through LDS).  gc / gmu of the
// centre atom are flushed per run with float atomics (gc pre-zeroed, gmu pre-set to gmu_out by the launcher).
// GEOM: geometry gradient only (no neighbour gradients gathered, nothing flushed); MU0: mu == 0 everywhere.
// One workgroup per CU (launch bounds 256, 1): the wave may use the whole 512-entry register file (VGPR + AGPR), so the
// gathers of a whole part are in flight behind the part's GEMMs.
template <int F, int KPB, bool GEOM, bool MU0, bool SPLIT = false>
__global__ __launch_bounds__(256, 1) void k_painn_msg_tile_bwd(MsgArgs a, int ntiles, int xcd_map) {
  constexpr int NT = F / 32;
  constexpr int NB = 3 * NT;
  constexpr int RSTR = 9;             // [32 edges][8 quad partials + 1 pad] per quantity
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                                        // NB * KPB * 256 floats
  TileRec* sE = (TileRec*)(sW + NB * KPB * 256);           // 4 waves x 32 records
  float* sR = (float*)(sE + 4 * 32);                       // 4 waves x 4 quantities x 32 x RSTR
  int* sCnt = (int*)(sR + 4 * 4 * 32 * RSTR);
  const int K = a.rb.n_rbf;
  constexpr int KPS = (KPB == 3 || KPB == 4) ? KPB : 4;
  const char* sWh = (const char*)smem;
  const char* sWl = sWh + PtImage<KPS, NB>::BYTES;

  if (SPLIT) stage_filter_split<KPS, NB>((char*)smem, (char*)smem + PtImage<KPS, NB>::BYTES, a.wf, a.bf, K);
  else stage_filter<NB * KPB * 64>(sW, a.wf, a.bf, K, KPB);
  if (threadIdx.x == 0) sCnt[0] = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  TileRec* myE = sE + wv * 32;
  float* myR = sR + wv * (4 * 32 * RSTR);
  const int slot = slot_of_row(el);
  constexpr unsigned F3 = 3u * F;

  while (true) {
    int nidx = 0;
    if (lane == 0) nidx = atomicAdd(&sCnt[0], 1);
    nidx = __builtin_amdgcn_readfirstlane(nidx);
    int tile = (int)blockIdx.x + nidx * (int)gridDim.x;
    if (xcd_map) {      // the workgroups of one XCD walk a contiguous eighth of the tiles (see k_painn_msg_tile)
      const int per_xcd = (ntiles + 7) / 8, tl = ((int)blockIdx.x >> 3) + nidx * (((int)gridDim.x + 7) / 8);
      tile = tl < per_xcd ? ((int)blockIdx.x & 7) * per_xcd + tl : ntiles;
    }
    if (tile >= ntiles) break;

    const int64_t e_first = (int64_t)tile * 32;
    const int nvalid = (a.E - e_first) < 32 ? (int)(a.E - e_first) : 32;
    const bool valid = slot < nvalid;
    const int64_t e = e_first + (valid ? slot : (nvalid - 1));
    const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
    const int ci = (int)a.idx_i[e], cj = (int)a.idx_j[e];
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    const float invd = 1.0f / d;
    float fc, dfc;
    spk_cutoff_eval_fast(a.rb.cutoff, d, fc, dfc);
    if (!valid) { fc = 0.f; dfc = 0.f; }
    if (hi == 0) {
      TileRec rec; rec.i = ci; rec.j = cj; rec.ux = rx * invd; rec.uy = ry * invd; rec.uz = rz * invd; rec.invd = invd; rec.r0 = 0.f; rec.r1 = 0.f;
      myE[slot] = rec;
    }
    float Av[KPB][4], Ad[KPB][4];
    h16x8 Avh[2], Avl[2], Adh[2], Adl[2];
    if (SPLIT) tile_operands_split<KPS, true>(a.rb, K, hi, d, fc, dfc, Avh, Avl, Adh, Adl);
    else {
#pragma unroll
    for (int u = 0; u < KPB; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int k = 8 * u + 4 * hi + v;
        float p, dp;
        spk_rbf_eval_fast(a.rb, k, d, p, dp);
        if (k == K) { p = 1.0f; dp = 0.f; }
        Av[u][v] = fc * p;
        Ad[u][v] = fc * dp + dfc * p;
      }
    }
    auto gemm_v = [&](int t) -> f32x16 {
      if (SPLIT) return tile_gemm_split<KPS, NB>(sWh, sWl, t, Avh, Avl, lane);
      return tile_gemm<KPB>(sW, t, Av, lane);
    };
    auto gemm_d = [&](int t) -> f32x16 {
      if (SPLIT) return tile_gemm_split<KPS, NB>(sWh, sWl, t, Adh, Adl, lane);
      return tile_gemm<KPB>(sW, t, Ad, lane);
    };
    spk_wave_lds_sync();
    unsigned runmask = 0x8000u;
    {
      int prev = myE[16 * hi].i;
#pragma unroll
      for (int r = 1; r < 16; ++r) {
        const int cur = myE[16 * hi + r].i;
        if (cur != prev) runmask |= 1u << (r - 1);
        prev = cur;
      }
    }
    f32x16 sdd, stx, sty, stz;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sdd[r] = 0.f; stx[r] = 0.f; sty[r] = 0.f; stz[r] = 0.f; }

#pragma unroll 1
    for (int cb = 0; cb < NT; ++cb) {
      const unsigned c0 = 32u * cb + el;
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- scalar part
      {
        f32x16 Pq = sdd;
        if (!GEOM) Pq = gemm_v(cb);
        const f32x16 Dq = gemm_d(cb);
        float accq = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < 16; g0 += 4) {
          float cq[4], gqa[4], gqb[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const TileRec er = myE[16 * hi + g0 + rr];
            cq[rr] = a.c[(unsigned)er.j * F3 + c0];
            gqa[rr] = a.gq_out[(unsigned)er.i * F + c0];
            gqb[rr] = GEOM ? 0.f : a.gq_out[(unsigned)er.j * F + c0];
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int r = g0 + rr;
            sdd[r] = fmaf(cq[rr] * gqa[rr], Dq[r], sdd[r]);
            if (!GEOM) {
              accq = fmaf(Pq[r], gqb[rr], accq);
              if ((runmask >> r) & 1u) { unsafeAtomicAdd(a.gc + ((unsigned)myE[16 * hi + r].i * F3 + c0), accq); accq = 0.f; }
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // the gathers of one part at a time
      // ---------------- R part
      {
        const f32x16 PR = gemm_v(NT + cb);
        const f32x16 DR = gemm_d(NT + cb);
        float accR = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < 16; g0 += 4) {
          float cR[4], ga0[4], ga1[4], ga2[4], gb0[4], gb1[4], gb2[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const TileRec er = myE[16 * hi + g0 + rr];
            const unsigned oj = (unsigned)er.j * F3 + c0, oi = (unsigned)er.i * F3 + c0;
            cR[rr] = a.c[oj + F];
            ga0[rr] = a.gmu_out[oi]; ga1[rr] = a.gmu_out[oi + F]; ga2[rr] = a.gmu_out[oi + 2 * F];
            if (!GEOM) { gb0[rr] = a.gmu_out[oj]; gb1[rr] = a.gmu_out[oj + F]; gb2[rr] = a.gmu_out[oj + 2 * F]; }
            else { gb0[rr] = 0.f; gb1[rr] = 0.f; gb2[rr] = 0.f; }
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int r = g0 + rr;
            const TileRec er = myE[16 * hi + r];
            const float gu = ga0[rr] * er.ux + ga1[rr] * er.uy + ga2[rr] * er.uz;
            sdd[r] = fmaf(cR[rr] * gu, DR[r], sdd[r]);
            const float mR = PR[r] * cR[rr];
            stx[r] = fmaf(ga0[rr], mR, stx[r]); sty[r] = fmaf(ga1[rr], mR, sty[r]); stz[r] = fmaf(ga2[rr], mR, stz[r]);
            if (!GEOM) {
              accR = fmaf(-PR[r], gb0[rr] * er.ux + gb1[rr] * er.uy + gb2[rr] * er.uz, accR);
              if ((runmask >> r) & 1u) { unsafeAtomicAdd(a.gc + ((unsigned)er.i * F3 + F + c0), accR); accR = 0.f; }
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- mu part
      if (!(GEOM && MU0)) {
        f32x16 Pm = sdd, Dm = sdd;
        if (!GEOM) Pm = gemm_v(2 * NT + cb);
        if (!MU0) Dm = gemm_d(2 * NT + cb);
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < 16; g0 += 4) {
          float cm[4], mb0[4], mb1[4], mb2[4], ga0[4], ga1[4], ga2[4], gb0[4], gb1[4], gb2[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const TileRec er = myE[16 * hi + g0 + rr];
            const unsigned oj = (unsigned)er.j * F3 + c0, oi = (unsigned)er.i * F3 + c0;
            if (!MU0) {
              cm[rr] = a.c[oj + 2 * F];
              mb0[rr] = a.mu[oj]; mb1[rr] = a.mu[oj + F]; mb2[rr] = a.mu[oj + 2 * F];
              ga0[rr] = a.gmu_out[oi]; ga1[rr] = a.gmu_out[oi + F]; ga2[rr] = a.gmu_out[oi + 2 * F];
            } else { cm[rr] = 0.f; mb0[rr] = 0.f; mb1[rr] = 0.f; mb2[rr] = 0.f; ga0[rr] = 0.f; ga1[rr] = 0.f; ga2[rr] = 0.f; }
            if (!GEOM) { gb0[rr] = a.gmu_out[oj]; gb1[rr] = a.gmu_out[oj + F]; gb2[rr] = a.gmu_out[oj + 2 * F]; }
            else { gb0[rr] = 0.f; gb1[rr] = 0.f; gb2[rr] = 0.f; }
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int r = g0 + rr;
            if (!MU0) {
              const float gm = ga0[rr] * mb0[rr] + ga1[rr] * mb1[rr] + ga2[rr] * mb2[rr];
              sdd[r] = fmaf(cm[rr] * gm, Dm[r], sdd[r]);
            }
            if (!GEOM) {
              v0 = fmaf(Pm[r], gb0[rr], v0); v1 = fmaf(Pm[r], gb1[rr], v1); v2 = fmaf(Pm[r], gb2[rr], v2);
              if ((runmask >> r) & 1u) {
                // gc[a][mu part] += mu_a . S,  gmu[a] += c_mu[a] S
                const unsigned oi = (unsigned)myE[16 * hi + r].i * F3 + c0;
                const float cma = a.c[oi + 2 * F];
                if (!MU0) {
                  const float ma0 = a.mu[oi], ma1 = a.mu[oi + F], ma2 = a.mu[oi + 2 * F];
                  unsafeAtomicAdd(a.gc + oi + 2 * F, ma0 * v0 + ma1 * v1 + ma2 * v2);
                }
                unsafeAtomicAdd(a.gmu + oi, cma * v0);
                unsafeAtomicAdd(a.gmu + oi + F, cma * v1);
                unsafeAtomicAdd(a.gmu + oi + 2 * F, cma * v2);
                v0 = 0.f; v1 = 0.f; v2 = 0.f;
              }
            }
          }
        }
      }
    }
    // ---- per-edge sums over the 32 channel lanes of each half: quad sums by DPP, 8 partials per edge through LDS
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float w0 = quad_sum(sdd[r]), w1 = quad_sum(stx[r]), w2 = quad_sum(sty[r]), w3 = quad_sum(stz[r]);
      if ((el & 3) == 0) {
        float* dst = myR + (16 * hi + r) * RSTR + (el >> 2);
        dst[0] = w0; dst[32 * RSTR] = w1; dst[2 * 32 * RSTR] = w2; dst[3 * 32 * RSTR] = w3;
      }
    }
    spk_wave_lds_sync();
    if (hi == 0 && valid) {
      float dd = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
      const float* src = myR + slot * RSTR;
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) { dd += src[k2]; tx += src[32 * RSTR + k2]; ty += src[2 * 32 * RSTR + k2]; tz += src[3 * 32 * RSTR + k2]; }
      if (d > 0.f) {
        const float ux = rx * invd, uy = ry * invd, uz = rz * invd;
        const float dot = tx * ux + ty * uy + tz * uz;
        a.gr[3 * e] += dd * ux + (tx - dot * ux) * invd;
        a.gr[3 * e + 1] += dd * uy + (ty - dot * uy) * invd;
        a.gr[3 * e + 2] += dd * uz + (tz - dot * uz) * invd;
      }
    }
    spk_wave_lds_sync();   // records / partials may be rewritten by the next tile
  }
}

// ---- row-tile backward (round 6) -------------------------------------------------------------------------------------------------
// The full first-order backward of the message (gc, gmu of the centre atom AND the geometry gradient; painn.py:50-66 transposed) with a
// wavefront per CSR row, as the row kernel of spk_painn.hip owns it -- but with the filter and its slope from the split-precision GEMM
// of 32-edge chunks of the row instead of 2 x 3 x n_rbf packed FMAs per edge and channel pair.  Lanes own a channel of the current
// 32-channel block (both halves the same channel, 16 edges each); the per-atom sums of a block are running sums over the 16 registers,
// parked per lane in LDS between the chunks of a row and added over the two halves at the end of the row: no atomics, no init launch,
// a fixed summation order.  The per-edge sums over channels are reduced as in k_painn_msg_tile_bwd.  Needs a sorted, symmetric list
// (Phi_e = Phi_rev(e), u_rev(e) = -u_e), as the row kernel does.
// GEOM / MU0 as above.  Pairs at or beyond the cutoff contribute zero through f_c = f_c' = 0 in the A operands.
// WANT_G: the geometry gradient (slope GEMMs, per-edge sums); WANT_T: the transposed sums gc / gmu of the row atom.  The full backward runs as a
// G launch and a T launch: together they gather every neighbour row once (c, mu for G; gq_out, gmu_out for T), and each keeps enough registers
// free to have the gathers of eight edges in flight.  GS = edges per gather batch.
constexpr int RT_LIVE_CAP = 256;     // row-tile backward: longest row whose live pairs are compacted (skin lists)
// SKIN: the list holds pairs beyond the cutoff (MD skin lists): the live pairs of a row are compacted first (own instance: the geometry pass sits at
// the 256-register limit, and the few registers of the compaction cost the plain instance 19 % -- 769 -> 915 us on the water box)
template <int F, int KPB, bool WANT_G, bool WANT_T, bool MU0, int GS, bool SKIN>
__global__ __launch_bounds__(256, 2) void k_painn_msg_rowtile_bwd(MsgArgs a) {
  constexpr bool GEOM = !WANT_T;
  constexpr int NT = F / 32;
  constexpr int NB = 3 * NT;
  constexpr int RSTR = 9;
  constexpr int NACC = 5;             // gc_q, gc_R, S_x, S_y, S_z
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* sWh = (char*)smem;
  char* sWl = sWh + PtImage<KPB, NB>::BYTES;
  TileRec* sE = (TileRec*)(sWl + PtImage<KPB, NB>::BYTES);    // 4 waves x 32 records
  float* sR = (float*)(sE + 4 * 32);                           // WANT_G: 4 waves x 4 quantities x 32 x RSTR
  float* sAcc = sR + (WANT_G ? 4 * 4 * 32 * RSTR : 0);         // WANT_T: 4 waves x NT x NACC x 64
  int* sLive = (int*)(sAcc + (WANT_T ? 4 * NT * NACC * 64 : 0));   // SKIN: 4 waves x RT_LIVE_CAP edge numbers
  // (two workgroups per CU must fit the 160 KB of LDS: with both areas allocated a workgroup took 82 KB and every launch ran at half occupancy, 1.7 x slower)
  const int K = a.rb.n_rbf;

  stage_filter_split<KPB, NB>(sWh, sWl, a.wf, a.bf, K);
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  TileRec* myE = sE + wv * 32;
  float* myR = sR + wv * (4 * 32 * RSTR);
  float* myA = sAcc + wv * (NT * NACC * 64);
  int* myL = sLive + wv * RT_LIVE_CAP;
  const int slot = slot_of_row(el);
  constexpr unsigned F3 = 3u * F;

  // the workgroups of one XCD walk a contiguous eighth of the atoms (as k_painn_msg_row)
  const int64_t per_xcd = a.xcd_map ? (a.N + 7) / 8 : a.N;
  const int64_t a_lo = a.xcd_map ? (int64_t)(blockIdx.x & 7) * per_xcd : 0;
  const int64_t a_hi = a.xcd_map ? (a_lo + per_xcd < a.N ? a_lo + per_xcd : a.N) : a.N;
  const int64_t a_first = a.xcd_map ? a_lo + (int64_t)(blockIdx.x >> 3) * 4 + wv : (int64_t)blockIdx.x * 4 + wv;
  const int64_t a_step = a.xcd_map ? (int64_t)((gridDim.x + 7) >> 3) * 4 : (int64_t)gridDim.x * 4;
  for (int64_t atom = a_first; atom < a_hi; atom += a_step) {
    const int32_t e0 = a.rowptr[atom], e1 = a.rowptr[atom + 1];
    const unsigned oa = (unsigned)atom * F3;
    if (!GEOM) {
#pragma unroll
      for (int q = 0; q < NT * NACC; ++q) myA[q * 64 + lane] = 0.f;
    }
    // Skin lists (MD): the pairs of the row inside the cutoff are compacted into a wave-private list first (lanes = pairs, 64 at a time), so
    // that the chunks -- matrix work and gathers -- hold live pairs only.  Rows longer than the list are walked as they are.
    int32_t n_work = e1 - e0;
    const bool compact = SKIN && n_work <= RT_LIVE_CAP;
    if (SKIN && compact) {
      int nl = 0;
      for (int32_t cs = e0; cs < e1; cs += 64) {
        const int32_t em = cs + lane;
        bool live = em < e1;
        if (live) {
          const float x_ = a.rij[3 * (int64_t)em], y_ = a.rij[3 * (int64_t)em + 1], z_ = a.rij[3 * (int64_t)em + 2];
          live = x_ * x_ + y_ * y_ + z_ * z_ < a.rb.cutoff * a.rb.cutoff;
        }
        const unsigned long long mask = __ballot(live);
        const int pos = nl + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (live) myL[pos] = em;
        nl += __popcll(mask);
      }
      n_work = nl;
      spk_wave_lds_sync();
    }
    for (int32_t cs = 0; cs < n_work; cs += 32) {
      // ---- geometry of this lane's edge slot (lanes 32..63 mirror lanes 0..31)
      const bool valid = cs + slot < n_work;
      const int32_t sl = valid ? cs + slot : n_work - 1;
      const int64_t e = (SKIN && compact) ? myL[sl] : e0 + sl;
      const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
      const int cj = (int)a.idx_j[e];
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      const float invd = 1.0f / d;
      float fc, dfc;
      spk_cutoff_eval_fast(a.rb.cutoff, d, fc, dfc);
      if (!valid) { fc = 0.f; dfc = 0.f; }
      if (hi == 0) {
        TileRec rec; rec.i = (int)atom; rec.j = cj; rec.ux = rx * invd; rec.uy = ry * invd; rec.uz = rz * invd; rec.invd = invd; rec.r0 = 0.f; rec.r1 = 0.f;
        myE[slot] = rec;
      }
      h16x8 Avh[2], Avl[2], Adh[2], Adl[2];
      tile_operands_split<KPB, WANT_G>(a.rb, K, hi, d, fc, dfc, Avh, Avl, Adh, Adl);
      spk_wave_lds_sync();
      f32x16 sdd, stx, sty, stz;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sdd[r] = 0.f; stx[r] = 0.f; sty[r] = 0.f; stz[r] = 0.f; }

      if constexpr (!WANT_G) {
        // ---- transposed sums, software-pipelined over the channel blocks: the neighbours' gmu rows of block cb + 1 (48 gathers) are requested
        // before the GEMMs of block cb, so that a round of gather latency is hidden behind a whole block of work instead of three GEMMs.  (The gq
        // row of the block joins in-iteration: 64 prefetched loads would need s_waitcnt vmcnt(64), one more than the counter holds.)
        float gbA[3][16], gbB[3][16];
        auto load_gb = [&](float (&gb)[3][16], int cb_) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned bj = ((unsigned)myE[16 * hi + r].j * F3 + 32u * cb_ + el) * 4u;
            gb[0][r] = ld_off(a.gmu_out, bj); gb[1][r] = ld_off(a.gmu_out, bj + 4 * F); gb[2][r] = ld_off(a.gmu_out, bj + 8 * F);
          }
        };
        load_gb(gbA, 0);
#pragma unroll
        for (int cb = 0; cb < NT; ++cb) {
          const unsigned c0 = 32u * cb + el;
          float (&cur)[3][16] = (cb & 1) ? gbB : gbA;
          float (&nxt)[3][16] = (cb & 1) ? gbA : gbB;
          float gqb[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) gqb[r] = ld_off(a.gq_out, ((unsigned)myE[16 * hi + r].j * F + c0) * 4u);
          if (cb + 1 < NT) load_gb(nxt, cb + 1);
          __builtin_amdgcn_sched_barrier(0);
          const f32x16 Pq = tile_gemm_split<KPB, NB>(sWh, sWl, cb, Avh, Avl, lane);
          const f32x16 PR = tile_gemm_split<KPB, NB>(sWh, sWl, NT + cb, Avh, Avl, lane);
          const f32x16 Pm = tile_gemm_split<KPB, NB>(sWh, sWl, 2 * NT + cb, Avh, Avl, lane);
          float accq = 0.f, accR = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const TileRec er = myE[16 * hi + r];
            accq = fmaf(Pq[r], gqb[r], accq);
            accR = fmaf(-PR[r], cur[0][r] * er.ux + cur[1][r] * er.uy + cur[2][r] * er.uz, accR);
            v0 = fmaf(Pm[r], cur[0][r], v0); v1 = fmaf(Pm[r], cur[1][r], v1); v2 = fmaf(Pm[r], cur[2][r], v2);
          }
          myA[(cb * NACC + 0) * 64 + lane] += accq;
          myA[(cb * NACC + 1) * 64 + lane] += accR;
          myA[(cb * NACC + 2) * 64 + lane] += v0; myA[(cb * NACC + 3) * 64 + lane] += v1; myA[(cb * NACC + 4) * 64 + lane] += v2;
          __builtin_amdgcn_sched_barrier(0);
        }
      } else
      if constexpr (WANT_G && !MU0) {
        // ---- geometry pass of an interaction with vector features, in TWO sweeps over the channel blocks so that neither needs more than the
        // register file: (S) dd = sum_c [c_q gq_i dF_q + c_R (gmu_i . u) dF_R + c_mu (gmu_i . mu_j) dF_mu] -- three slope GEMMs, six gathers, 16
        // running sums; (V) t = sum_c gmu_i (F_R c_R) -- one value GEMM, one gather, 48 running sums.  (In one sweep, with the gathers of a batch
        // held together by the scheduling barrier, the kernel spills 324 B per lane: 1 183 us.)
#pragma unroll 1
        for (int cb = 0; cb < NT; ++cb) {
          const unsigned c0 = 32u * cb + el;
          __builtin_amdgcn_sched_barrier(0);
          const float gqa = ld_off(a.gq_out, ((unsigned)atom * F + c0) * 4u);
          const float ga0 = ld_off(a.gmu_out, (oa + c0) * 4u), ga1 = ld_off(a.gmu_out, (oa + c0) * 4u + 4 * F), ga2 = ld_off(a.gmu_out, (oa + c0) * 4u + 8 * F);
          float cq[16], cR[16], cm[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned bj = ((unsigned)myE[16 * hi + r].j * F3 + c0) * 4u;
            cq[r] = ld_off(a.c, bj); cR[r] = ld_off(a.c, bj + 4 * F); cm[r] = ld_off(a.c, bj + 8 * F);
          }
          __builtin_amdgcn_sched_barrier(0);
          const f32x16 Dq = tile_gemm_split<KPB, NB>(sWh, sWl, cb, Adh, Adl, lane);
          const f32x16 DR = tile_gemm_split<KPB, NB>(sWh, sWl, NT + cb, Adh, Adl, lane);
          const f32x16 Dm = tile_gemm_split<KPB, NB>(sWh, sWl, 2 * NT + cb, Adh, Adl, lane);
#pragma unroll
          for (int g0 = 0; g0 < 16; g0 += GS) {
            float mb0[GS], mb1[GS], mb2[GS];
#pragma unroll
            for (int rr = 0; rr < GS; ++rr) {
              const unsigned bj = ((unsigned)myE[16 * hi + g0 + rr].j * F3 + c0) * 4u;
              mb0[rr] = ld_off(a.mu, bj); mb1[rr] = ld_off(a.mu, bj + 4 * F); mb2[rr] = ld_off(a.mu, bj + 8 * F);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rr = 0; rr < GS; ++rr) {
              const int r = g0 + rr;
              const TileRec er = myE[16 * hi + r];
              const float gu = ga0 * er.ux + ga1 * er.uy + ga2 * er.uz;
              const float gm = ga0 * mb0[rr] + ga1 * mb1[rr] + ga2 * mb2[rr];
              float sv = fmaf(cq[r] * gqa, Dq[r], sdd[r]);
              sv = fmaf(cR[r] * gu, DR[r], sv);
              sdd[r] = fmaf(cm[r] * gm, Dm[r], sv);
            }
          }
        }
#pragma unroll 1
        for (int cb = 0; cb < NT; ++cb) {
          const unsigned c0 = 32u * cb + el;
          __builtin_amdgcn_sched_barrier(0);
          const float ga0 = ld_off(a.gmu_out, (oa + c0) * 4u), ga1 = ld_off(a.gmu_out, (oa + c0) * 4u + 4 * F), ga2 = ld_off(a.gmu_out, (oa + c0) * 4u + 8 * F);
          float cR[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) cR[r] = ld_off(a.c, ((unsigned)myE[16 * hi + r].j * F3 + c0) * 4u + 4 * F);
          __builtin_amdgcn_sched_barrier(0);
          const f32x16 PR = tile_gemm_split<KPB, NB>(sWh, sWl, NT + cb, Avh, Avl, lane);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float mR = PR[r] * cR[r];
            stx[r] = fmaf(ga0, mR, stx[r]); sty[r] = fmaf(ga1, mR, sty[r]); stz[r] = fmaf(ga2, mR, stz[r]);
          }
        }
      } else {
#pragma unroll 1
      for (int cb = 0; cb < NT; ++cb) {
        const unsigned c0 = 32u * cb + el;
        __builtin_amdgcn_sched_barrier(0);
#if SPK_RT_HOLLOW == 1
        continue;
#endif
        if (!WANT_G) {
          // ================ transposed sums: every gather of the block is requested before its three GEMMs (ONE latency round per block:
          // the launch is bound by the rounds of dependent gathers a wave walks through, not by their bytes)
          float gqb[16], gb0[16], gb1[16], gb2[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned j = (unsigned)myE[16 * hi + r].j;
            const unsigned oj = j * F3 + c0;
            const unsigned bq = (j * F + c0) * 4u, bj = oj * 4u;       // byte offsets (< 2^32, checked by the launcher): scalar base + 32-bit lane offset
#if SPK_RT_HOLLOW == 2
            gqb[r] = (float)(j & 7); gb0[r] = (float)(oj & 3); gb1[r] = (float)(oj & 5); gb2[r] = (float)(oj & 9);
#else
            gqb[r] = ld_off(a.gq_out, bq);
            gb0[r] = ld_off(a.gmu_out, bj); gb1[r] = ld_off(a.gmu_out, bj + 4 * F); gb2[r] = ld_off(a.gmu_out, bj + 8 * F);
#endif
          }
          __builtin_amdgcn_sched_barrier(0);     // all 64 gathers are requested before the first MFMA
#if SPK_RT_HOLLOW == 3
          f32x16 Pq, PR, Pm;
          for (int r = 0; r < 16; ++r) { Pq[r] = fc; PR[r] = d; Pm[r] = invd; }
#else
          const f32x16 Pq = tile_gemm_split<KPB, NB>(sWh, sWl, cb, Avh, Avl, lane);
          const f32x16 PR = tile_gemm_split<KPB, NB>(sWh, sWl, NT + cb, Avh, Avl, lane);
          const f32x16 Pm = tile_gemm_split<KPB, NB>(sWh, sWl, 2 * NT + cb, Avh, Avl, lane);
#endif
          float accq = 0.f, accR = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const TileRec er = myE[16 * hi + r];
            accq = fmaf(Pq[r], gqb[r], accq);
            accR = fmaf(-PR[r], gb0[r] * er.ux + gb1[r] * er.uy + gb2[r] * er.uz, accR);
            v0 = fmaf(Pm[r], gb0[r], v0); v1 = fmaf(Pm[r], gb1[r], v1); v2 = fmaf(Pm[r], gb2[r], v2);
          }
          myA[(cb * NACC + 0) * 64 + lane] += accq;
          myA[(cb * NACC + 1) * 64 + lane] += accR;
          myA[(cb * NACC + 2) * 64 + lane] += v0; myA[(cb * NACC + 3) * 64 + lane] += v1; myA[(cb * NACC + 4) * 64 + lane] += v2;
          continue;
        }
        // ================ geometry gradient: the slope GEMMs of the three parts (and the value GEMM of the R part) first, then the neighbours' rows
        // in batches of GS edges with the gathers of ALL parts of a batch in flight together
        // (element indices here: with the byte offsets of the sums pass this sweep spills 204 B per lane and runs 1 036 us instead of 753; split into a dd
        //  sweep and a t sweep it needs 214 registers, no scratch -- and 849 us: profiles/r06_painn_box.md)
        const float gqa = a.gq_out[(unsigned)atom * F + c0];
        const float ga0 = a.gmu_out[oa + c0], ga1 = a.gmu_out[oa + F + c0], ga2 = a.gmu_out[oa + 2 * F + c0];
#if SPK_RT_HOLLOW == 3
        f32x16 Dq, PR, DR, Dm;
        for (int r = 0; r < 16; ++r) { Dq[r] = fc; PR[r] = d; DR[r] = invd; Dm[r] = dfc; }
#else
        const f32x16 Dq = tile_gemm_split<KPB, NB>(sWh, sWl, cb, Adh, Adl, lane);
        const f32x16 PR = tile_gemm_split<KPB, NB>(sWh, sWl, NT + cb, Avh, Avl, lane);
        const f32x16 DR = tile_gemm_split<KPB, NB>(sWh, sWl, NT + cb, Adh, Adl, lane);
        f32x16 Dm = Dq;
        if (!MU0) Dm = tile_gemm_split<KPB, NB>(sWh, sWl, 2 * NT + cb, Adh, Adl, lane);
#endif
#pragma unroll
        for (int g0 = 0; g0 < 16; g0 += GS) {
          float cq[GS], cR[GS], cm[GS], mb0[GS], mb1[GS], mb2[GS];
#pragma unroll
          for (int rr = 0; rr < GS; ++rr) {
            const unsigned oj = (unsigned)myE[16 * hi + g0 + rr].j * F3 + c0;
#if SPK_RT_HOLLOW == 2
            cq[rr] = (float)(oj & 3); cR[rr] = (float)(oj & 5); cm[rr] = (float)(oj & 9); mb0[rr] = (float)(oj & 17); mb1[rr] = (float)(oj & 33); mb2[rr] = (float)(oj & 65);
#else
            cq[rr] = a.c[oj];
            cR[rr] = a.c[oj + F];
            if (!MU0) { cm[rr] = a.c[oj + 2 * F]; mb0[rr] = a.mu[oj]; mb1[rr] = a.mu[oj + F]; mb2[rr] = a.mu[oj + 2 * F]; }
            else { cm[rr] = 0.f; mb0[rr] = 0.f; mb1[rr] = 0.f; mb2[rr] = 0.f; }
#endif
          }
          __builtin_amdgcn_sched_barrier(0);     // the gathers of a batch are requested together
#pragma unroll
          for (int rr = 0; rr < GS; ++rr) {
            const int r = g0 + rr;
            const TileRec er = myE[16 * hi + r];
            const float gu = ga0 * er.ux + ga1 * er.uy + ga2 * er.uz;
            float s = fmaf(cq[rr] * gqa, Dq[r], sdd[r]);
            s = fmaf(cR[rr] * gu, DR[r], s);
            if (!MU0) {
              const float gm = ga0 * mb0[rr] + ga1 * mb1[rr] + ga2 * mb2[rr];
              s = fmaf(cm[rr] * gm, Dm[r], s);
            }
            sdd[r] = s;
            const float mR = PR[r] * cR[rr];
            stx[r] = fmaf(ga0, mR, stx[r]); sty[r] = fmaf(ga1, mR, sty[r]); stz[r] = fmaf(ga2, mR, stz[r]);
          }
        }
      }
      }
      // ---- per-edge sums over the 32 channel lanes of each half: quad sums by DPP, 8 partials per edge through LDS
      if (WANT_G) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float w0 = quad_sum(sdd[r]), w1 = quad_sum(stx[r]), w2 = quad_sum(sty[r]), w3 = quad_sum(stz[r]);
        if ((el & 3) == 0) {
          float* dst = myR + (16 * hi + r) * RSTR + (el >> 2);
          dst[0] = w0; dst[32 * RSTR] = w1; dst[2 * 32 * RSTR] = w2; dst[3 * 32 * RSTR] = w3;
        }
      }
      spk_wave_lds_sync();
      if (hi == 0 && valid) {
        float dd = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
        const float* src = myR + slot * RSTR;
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) { dd += src[k2]; tx += src[32 * RSTR + k2]; ty += src[2 * 32 * RSTR + k2]; tz += src[3 * 32 * RSTR + k2]; }
        if (d > 0.f) {
          const float ux = rx * invd, uy = ry * invd, uz = rz * invd;
          const float dot = tx * ux + ty * uy + tz * uz;
          a.gr[3 * e] += dd * ux + (tx - dot * ux) * invd;
          a.gr[3 * e + 1] += dd * uy + (ty - dot * uy) * invd;
          a.gr[3 * e + 2] += dd * uz + (tz - dot * uz) * invd;
        }
      }
      }
      spk_wave_lds_sync();   // records / partials may be rewritten by the next chunk
    }
    if (!GEOM) {
      // ---- sums of the row: both halves added, results of the centre atom written once (gc = transposed sums, gmu = gmu_out + c_mu S)
      spk_wave_lds_sync();
#pragma unroll
      for (int cb = 0; cb < NT; ++cb) {
        float t[NACC];
#pragma unroll
        for (int q = 0; q < NACC; ++q) t[q] = myA[(cb * NACC + q) * 64 + lane] + myA[(cb * NACC + q) * 64 + (lane ^ 32)];
        if (hi == 0) {
          const unsigned c0 = 32u * cb + el;
          const float cma = a.c[oa + 2 * F + c0];
          const float g0 = a.gmu_out[oa + c0], g1 = a.gmu_out[oa + F + c0], g2 = a.gmu_out[oa + 2 * F + c0];
          float gcm = 0.f;
          if (!MU0) gcm = a.mu[oa + c0] * t[2] + a.mu[oa + F + c0] * t[3] + a.mu[oa + 2 * F + c0] * t[4];
          a.gc[oa + c0] = t[0];
          a.gc[oa + F + c0] = t[1];
          a.gc[oa + 2 * F + c0] = gcm;
          a.gmu[oa + c0] = g0 + cma * t[2];
          a.gmu[oa + F + c0] = g1 + cma * t[3];
          a.gmu[oa + 2 * F + c0] = g2 + cma * t[4];
        }
      }
      spk_wave_lds_sync();   // the slots are zeroed for the next row
    }
  }
}

// ---- row-tile forward (round 6) ----------------------------------------------------------------------------------------------------
// The message sum of painn.py:50-66 with the ownership of the row-tile backward: a wavefront per row, the filter from the split-precision GEMM of
// 32-pair chunks, lanes own a channel of the current block, running sums over the 16 registers of a half parked in LDS between the chunks and
// added over the two halves at the end of the row: q_out = q + dq and mu_out = mu + dmu are written once -- no float atomics (hollowed out, the
// 32-edge tile kernel spends 450 of its 540 us with the gathers switched off and the same with the GEMMs switched off: what is left is the atomic
// flush of every run of a tile, 89 M element updates at the L2), no init launch, a fixed summation order.  Needs a sorted list with row pointers.
template <int F, int KPB, bool MU0, int GS, bool SKIN>
__global__ __launch_bounds__(256, 2) void k_painn_msg_rowtile_fwd(MsgArgs a) {
  constexpr int NT = F / 32;
  constexpr int NB = 3 * NT;
  constexpr int NACC = 4;             // dq, dmu_x, dmu_y, dmu_z
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* sWh = (char*)smem;
  char* sWl = sWh + PtImage<KPB, NB>::BYTES;
  TileRec* sE = (TileRec*)(sWl + PtImage<KPB, NB>::BYTES);    // 4 waves x 32 records
  float* sAcc = (float*)(sE + 4 * 32);                         // 4 waves x NT x NACC x 64
  int* sLive = (int*)(sAcc + 4 * NT * NACC * 64);              // SKIN: 4 waves x RT_LIVE_CAP edge numbers
  const int K = a.rb.n_rbf;

  stage_filter_split<KPB, NB>(sWh, sWl, a.wf, a.bf, K);
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi = lane >> 5, el = lane & 31;
  TileRec* myE = sE + wv * 32;
  float* myA = sAcc + wv * (NT * NACC * 64);
  int* myL = sLive + wv * RT_LIVE_CAP;
  const int slot = slot_of_row(el);
  constexpr unsigned F3 = 3u * F;

  const int64_t per_xcd = a.xcd_map ? (a.N + 7) / 8 : a.N;
  const int64_t a_lo = a.xcd_map ? (int64_t)(blockIdx.x & 7) * per_xcd : 0;
  const int64_t a_hi = a.xcd_map ? (a_lo + per_xcd < a.N ? a_lo + per_xcd : a.N) : a.N;
  const int64_t a_first = a.xcd_map ? a_lo + (int64_t)(blockIdx.x >> 3) * 4 + wv : (int64_t)blockIdx.x * 4 + wv;
  const int64_t a_step = a.xcd_map ? (int64_t)((gridDim.x + 7) >> 3) * 4 : (int64_t)gridDim.x * 4;
  for (int64_t atom = a_first; atom < a_hi; atom += a_step) {
    const int32_t e0 = a.rowptr[atom], e1 = a.rowptr[atom + 1];
    const unsigned oa = (unsigned)atom * F3;
#pragma unroll
    for (int q = 0; q < NT * NACC; ++q) myA[q * 64 + lane] = 0.f;
    int32_t n_work = e1 - e0;
    const bool compact = SKIN && n_work <= RT_LIVE_CAP;
    if (SKIN && compact) {
      int nl = 0;
      for (int32_t cs = e0; cs < e1; cs += 64) {
        const int32_t em = cs + lane;
        bool live = em < e1;
        if (live) {
          const float x_ = a.rij[3 * (int64_t)em], y_ = a.rij[3 * (int64_t)em + 1], z_ = a.rij[3 * (int64_t)em + 2];
          live = x_ * x_ + y_ * y_ + z_ * z_ < a.rb.cutoff * a.rb.cutoff;
        }
        const unsigned long long mask = __ballot(live);
        const int pos = nl + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (live) myL[pos] = em;
        nl += __popcll(mask);
      }
      n_work = nl;
      spk_wave_lds_sync();
    }
    for (int32_t cs = 0; cs < n_work; cs += 32) {
      const bool valid = cs + slot < n_work;
      const int32_t sl = valid ? cs + slot : n_work - 1;
      const int64_t e = (SKIN && compact) ? myL[sl] : e0 + sl;
      const float rx = a.rij[3 * e], ry = a.rij[3 * e + 1], rz = a.rij[3 * e + 2];
      const int cj = (int)a.idx_j[e];
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      const float invd = 1.0f / d;
      float fc, dfc;
      spk_cutoff_eval_fast(a.rb.cutoff, d, fc, dfc);
      if (!valid) fc = 0.f;
      if (hi == 0) {
        TileRec rec; rec.i = (int)atom; rec.j = cj; rec.ux = rx * invd; rec.uy = ry * invd; rec.uz = rz * invd; rec.invd = invd; rec.r0 = 0.f; rec.r1 = 0.f;
        myE[slot] = rec;
      }
      h16x8 Avh[2], Avl[2], Adh_[2], Adl_[2];
      tile_operands_split<KPB, false>(a.rb, K, hi, d, fc, 0.f, Avh, Avl, Adh_, Adl_);
      spk_wave_lds_sync();

      // software-pipelined over the channel blocks: the scalar rows (c_q, c_R) of block cb + 1 are requested before the GEMMs of block cb; the vector
      // rows follow inside the block in batches of GS edges
      float cA[2][16], cB[2][16];
      auto load_c = [&](float (&cc)[2][16], int cb_) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned bj = ((unsigned)myE[16 * hi + r].j * F3 + 32u * cb_ + el) * 4u;
          cc[0][r] = ld_off(a.c, bj); cc[1][r] = ld_off(a.c, bj + 4 * F);
        }
      };
      load_c(cA, 0);
#pragma unroll
      for (int cb = 0; cb < NT; ++cb) {
        const unsigned c0 = 32u * cb + el;
        float (&cq)[16] = (cb & 1) ? cB[0] : cA[0];
        float (&cR)[16] = (cb & 1) ? cB[1] : cA[1];
        __builtin_amdgcn_sched_barrier(0);
        if (cb + 1 < NT) { if (cb & 1) load_c(cA, cb + 1); else load_c(cB, cb + 1); }
        __builtin_amdgcn_sched_barrier(0);     // (without the barriers the scheduler sinks every gather to its use and waits for them one by one: 994 us)
        const f32x16 Pq = tile_gemm_split<KPB, NB>(sWh, sWl, cb, Avh, Avl, lane);
        const f32x16 PR = tile_gemm_split<KPB, NB>(sWh, sWl, NT + cb, Avh, Avl, lane);
        f32x16 Pm = PR;
        if (!MU0) Pm = tile_gemm_split<KPB, NB>(sWh, sWl, 2 * NT + cb, Avh, Avl, lane);
        float accq = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < 16; g0 += GS) {
          float cm[GS], m0[GS], m1[GS], m2[GS];
#pragma unroll
          for (int rr = 0; rr < GS; ++rr) {
            const unsigned bj = ((unsigned)myE[16 * hi + g0 + rr].j * F3 + c0) * 4u;
            if (!MU0) { cm[rr] = ld_off(a.c, bj + 8 * F); m0[rr] = ld_off(a.mu, bj); m1[rr] = ld_off(a.mu, bj + 4 * F); m2[rr] = ld_off(a.mu, bj + 8 * F); }
            else { cm[rr] = 0.f; m0[rr] = 0.f; m1[rr] = 0.f; m2[rr] = 0.f; }
          }
          if (!MU0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rr = 0; rr < GS; ++rr) {
            const int r = g0 + rr;
            const TileRec er = myE[16 * hi + r];
            accq = fmaf(Pq[r], cq[r], accq);
            const float mR = PR[r] * cR[r], mm = MU0 ? 0.f : Pm[r] * cm[rr];
            a0 = fmaf(mR, er.ux, MU0 ? a0 : fmaf(mm, m0[rr], a0));
            a1 = fmaf(mR, er.uy, MU0 ? a1 : fmaf(mm, m1[rr], a1));
            a2 = fmaf(mR, er.uz, MU0 ? a2 : fmaf(mm, m2[rr], a2));
          }
        }
        myA[(cb * NACC + 0) * 64 + lane] += accq;
        myA[(cb * NACC + 1) * 64 + lane] += a0; myA[(cb * NACC + 2) * 64 + lane] += a1; myA[(cb * NACC + 3) * 64 + lane] += a2;
      }
      spk_wave_lds_sync();   // records may be rewritten by the next chunk
    }
    // ---- sums of the row: both halves added, q_out / mu_out of the centre atom written once
    spk_wave_lds_sync();
#pragma unroll
    for (int cb = 0; cb < NT; ++cb) {
      float t[NACC];
#pragma unroll
      for (int q = 0; q < NACC; ++q) t[q] = myA[(cb * NACC + q) * 64 + lane] + myA[(cb * NACC + q) * 64 + (lane ^ 32)];
      if (hi == 0) {
        const unsigned c0 = 32u * cb + el;
        a.q_out[(unsigned)atom * F + c0] = a.q[(unsigned)atom * F + c0] + t[0];
        a.mu_out[oa + c0] = (MU0 ? 0.f : a.mu[oa + c0]) + t[1];
        a.mu_out[oa + F + c0] = (MU0 ? 0.f : a.mu[oa + F + c0]) + t[2];
        a.mu_out[oa + 2 * F + c0] = (MU0 ? 0.f : a.mu[oa + 2 * F + c0]) + t[3];
      }
    }
    spk_wave_lds_sync();
  }
}

// the tile kernel accumulates with atomics: q_out = q, mu_out = mu first
__global__ void k_msg_tile_init(const float* __restrict__ s0, float* __restrict__ d0, int64_t n0, const float* __restrict__ s1,
                                float* __restrict__ d1, int64_t n1) {
  const int64_t n04 = n0 / 4, n14 = n1 / 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n04 + n14; k += (int64_t)gridDim.x * blockDim.x) {
    if (k < n04) ((f32x4*)d0)[k] = s0 ? ((const f32x4*)s0)[k] : z4;
    else ((f32x4*)d1)[k - n04] = s1 ? ((const f32x4*)s1)[k - n04] : z4;
  }
}

template <int F, int KPB, bool MU0>
int launch_tile(const MsgArgs& a, hipStream_t stream) {
  const int64_t nt = (a.E + 31) / 32;
  const size_t lds = (size_t)(3 * (F / 32) * KPB * 256) * sizeof(float) + 4 * 32 * sizeof(TileRec) + 16;
  static SpkPerDevice attr_done;
  int attr_done_dev;
  if (attr_done.pending(&attr_done_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_tile<F, KPB, MU0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done.mark(attr_done_dev);
  }
    // XCD-contiguous walk: 680 -> 625 us on the 32k-atom water box (profiles/r04_tile_experiments.txt)
  const int xcd_map = spk_xcd_walk_default();     // SPK_XCD_WALK=0 switches it off
  static const int wavesN = [] { const char* e = getenv("SPK_TILE_WAVES"); return (e && (e[0] == '3' || e[0] == '4')) ? e[0] - '0' : 0; }();
  if (wavesN) {
    static SpkPerDevice attrN;
    int attrN_dev;
    if (attrN.pending(&attrN_dev)) {
      SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_tile<F, KPB, MU0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_tile<F, KPB, MU0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attrN.mark(attrN_dev);
    }
    const int gridN = (spk_grid_for(nt, 4, spk_num_cus() * wavesN) + 7) / 8 * 8;
    if (wavesN == 3) hipLaunchKernelGGL((k_painn_msg_tile<F, KPB, MU0, 3>), dim3(gridN), dim3(256), lds, stream, a, (int)nt, xcd_map);
    else hipLaunchKernelGGL((k_painn_msg_tile<F, KPB, MU0, 4>), dim3(gridN), dim3(256), lds, stream, a, (int)nt, xcd_map);
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  const int grid = xcd_map ? (spk_grid_for(nt, 4, spk_num_cus() * 2) + 7) / 8 * 8 : spk_grid_for(nt, 4, spk_num_cus() * 2);
  if constexpr (KPB == 3 || KPB == 4) if (spk_get_split()) {
    static SpkPerDevice attr_sp;
    int attr_sp_dev;
    if (attr_sp.pending(&attr_sp_dev)) {
      SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_tile<F, KPB, MU0, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_sp.mark(attr_sp_dev);
    }
    hipLaunchKernelGGL((k_painn_msg_tile<F, KPB, MU0, 2, true>), dim3(grid), dim3(256), lds, stream, a, (int)nt, xcd_map);
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  hipLaunchKernelGGL((k_painn_msg_tile<F, KPB, MU0>), dim3(grid), dim3(256), lds, stream, a, (int)nt, xcd_map);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

template <int F, int KPB, bool GEOM, bool MU0>
int launch_tile_bwd(const MsgArgs& a, hipStream_t stream) {
  const int64_t nt = (a.E + 31) / 32;
  const size_t lds = (size_t)(3 * (F / 32) * KPB * 256) * sizeof(float) + 4 * 32 * sizeof(TileRec) + 4 * 4 * 32 * 9 * sizeof(float) + 16;
  static SpkPerDevice attr_done;
  int attr_done_dev;
  if (attr_done.pending(&attr_done_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_tile_bwd<F, KPB, GEOM, MU0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done.mark(attr_done_dev);
  }
  const int xcd_map = spk_xcd_walk_default();
  const int grid = xcd_map ? (spk_grid_for(nt, 4, spk_num_cus()) + 7) / 8 * 8 : spk_grid_for(nt, 4, spk_num_cus());
  if (spk_get_split()) {
    static SpkPerDevice attr_sp;
    int attr_sp_dev;
    if (attr_sp.pending(&attr_sp_dev)) {
      SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_tile_bwd<F, KPB, GEOM, MU0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_sp.mark(attr_sp_dev);
    }
    hipLaunchKernelGGL((k_painn_msg_tile_bwd<F, KPB, GEOM, MU0, true>), dim3(grid), dim3(256), lds, stream, a, (int)nt, xcd_map);
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  hipLaunchKernelGGL((k_painn_msg_tile_bwd<F, KPB, GEOM, MU0>), dim3(grid), dim3(256), lds, stream, a, (int)nt, xcd_map);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

template <int F, int KPB, bool SKIN>
int launch_rowtile_bwd(const MsgArgs& a_in, hipStream_t stream) {
  constexpr int NB = 3 * (F / 32);
  MsgArgs a = a_in;
  a.xcd_map = (spk_xcd_walk_default() && a.N >= (1 << 14)) ? 1 : 0;
  const size_t lds0 = 2 * (size_t)PtImage<KPB, NB>::BYTES + 4 * 32 * sizeof(TileRec) + (SKIN ? 4 * RT_LIVE_CAP * sizeof(int) : 0);
  const size_t lds_g = lds0 + (size_t)(4 * 4 * 32 * 9) * sizeof(float), lds_t = lds0 + (size_t)(4 * (F / 32) * 5 * 64) * sizeof(float);
  const size_t lds = lds_g > lds_t ? lds_g : lds_t;
  const bool mu0 = a.mu_zero != 0;
  // the geometry pass, then (unless only the geometry gradient is wanted) the transposed sums
  const void* kg = mu0 ? (const void*)k_painn_msg_rowtile_bwd<F, KPB, true, false, true, 8, SKIN> : (const void*)k_painn_msg_rowtile_bwd<F, KPB, true, false, false, 8, SKIN>;
  const void* kt = mu0 ? (const void*)k_painn_msg_rowtile_bwd<F, KPB, false, true, true, 8, SKIN> : (const void*)k_painn_msg_rowtile_bwd<F, KPB, false, true, false, 8, SKIN>;
  static SpkPerDevice attr_done;
  int attr_done_dev;
  if (attr_done.pending(&attr_done_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_rowtile_bwd<F, KPB, true, false, true, 8, SKIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_rowtile_bwd<F, KPB, true, false, false, 8, SKIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_rowtile_bwd<F, KPB, false, true, true, 8, SKIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_rowtile_bwd<F, KPB, false, true, false, 8, SKIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done.mark(attr_done_dev);
  }
  // (measured on the water box, profiles/r06_painn_box.md: 1 / 3 workgroups per CU lose 40-70 % / 25 %; starting the four rows of a workgroup in
  //  lockstep behind a barrier changes nothing -- the L1 is not where the neighbour rows are shared)
  const int grid = a.xcd_map ? (spk_grid_for(a.N, 4, spk_num_cus() * 2) + 7) / 8 * 8 : spk_grid_for(a.N, 4, spk_num_cus() * 2);
  void* args[] = {(void*)&a};
  if (a.geom_only != 2) {      // (geom_only == 2: the transposed sums alone -- the by-neighbour pass of an asymmetric list)
    SpkProfScope prof(a.geom_only ? "painn_msg_bwd_rowtile_geom" : "painn_msg_bwd_rowtile_g", stream);
    SPK_HIP_TRY(hipLaunchKernel(kg, dim3(grid), dim3(256), args, lds_g, stream));
  }
  if (a.geom_only != 1) {
    SpkProfScope prof("painn_msg_bwd_rowtile_t", stream);
    SPK_HIP_TRY(hipLaunchKernel(kt, dim3(grid), dim3(256), args, lds_t, stream));
  }
  return SPK_OK;
}

template <int F, int KPB, bool SKIN>
int launch_rowtile_fwd(const MsgArgs& a_in, hipStream_t stream) {
  constexpr int NB = 3 * (F / 32);
  MsgArgs a = a_in;
  a.xcd_map = (spk_xcd_walk_default() && a.N >= (1 << 14)) ? 1 : 0;
  const size_t lds = 2 * (size_t)PtImage<KPB, NB>::BYTES + 4 * 32 * sizeof(TileRec) + (size_t)(4 * (F / 32) * 4 * 64) * sizeof(float) + (SKIN ? 4 * RT_LIVE_CAP * sizeof(int) : 0);
  const void* kern = a.mu_zero ? (const void*)k_painn_msg_rowtile_fwd<F, KPB, true, 8, SKIN> : (const void*)k_painn_msg_rowtile_fwd<F, KPB, false, 8, SKIN>;
  static SpkPerDevice attr_done;
  int attr_done_dev;
  if (attr_done.pending(&attr_done_dev)) {
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_rowtile_fwd<F, KPB, true, 8, SKIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SPK_HIP_TRY(hipFuncSetAttribute((const void*)k_painn_msg_rowtile_fwd<F, KPB, false, 8, SKIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done.mark(attr_done_dev);
  }
  const int grid = a.xcd_map ? (spk_grid_for(a.N, 4, spk_num_cus() * 2) + 7) / 8 * 8 : spk_grid_for(a.N, 4, spk_num_cus() * 2);
  void* args[] = {(void*)&a};
  SpkProfScope prof(a.mu_zero ? "painn_msg_fwd_rowtile_mu0" : "painn_msg_fwd_rowtile", stream);
  SPK_HIP_TRY(hipLaunchKernel(kern, dim3(grid), dim3(256), args, lds, stream));
  return SPK_OK;
}

int g_tile_mode = 0;   // 0 auto (large lists), 1 always when the shape allows, -1 never
int g_rowtile_mode = 0;   // 0 auto (large sorted symmetric lists, split path on), 1 always when the shape allows, -1 never

}  // namespace

extern "C" void spk_painn_set_tile(int32_t mode) { g_tile_mode = mode > 0 ? 1 : (mode < 0 ? -1 : 0); }

bool spk_painn_msg_tile_ok(const MsgArgs& a) {
  const int kpb = a.rb.n_rbf / 8 + 1;    // room for the bias column
  if (g_tile_mode < 0 || !(a.F == 128 || a.F == 64) || kpb < 3 || kpb > 5 || a.E < 32 || a.N * 3 * (int64_t)a.F >= (1LL << 30)) return false;
  return g_tile_mode > 0 || (a.E >= (1 << 19) && !a.skin_list);
}

int spk_painn_msg_tile_fwd(const MsgArgs& a, hipStream_t stream) {
  const int64_t nf = a.N * (int64_t)a.F;
  hipLaunchKernelGGL(k_msg_tile_init, dim3(spk_grid_for(nf, 256, spk_num_cus() * 8)), dim3(256), 0, stream, a.q, a.q_out, nf, a.mu, a.mu_out, 3 * nf);
  SPK_LAUNCH_CHECK();
  const int kpb = a.rb.n_rbf / 8 + 1;
#define SPK_TILE_CASE(Fv, Kv) if (a.F == Fv && kpb == Kv) return a.mu_zero ? launch_tile<Fv, Kv, true>(a, stream) : launch_tile<Fv, Kv, false>(a, stream);
  SPK_TILE_CASE(128, 3) SPK_TILE_CASE(128, 4) SPK_TILE_CASE(128, 5)
  SPK_TILE_CASE(64, 3) SPK_TILE_CASE(64, 4) SPK_TILE_CASE(64, 5)
#undef SPK_TILE_CASE
  spk_set_error("painn message tile kernel: internal dispatch error (F=%d n_rbf=%d)", a.F, a.rb.n_rbf);
  return SPK_ERR_ARG;
}

// backward: gc = 0 and gmu = gmu_out first (unless only the geometry gradient is formed), then the tile kernel
int spk_painn_msg_tile_bwd(const MsgArgs& a, hipStream_t stream) {
  const int64_t nf = a.N * (int64_t)a.F;
  if (!a.geom_only) {
    hipLaunchKernelGGL(k_msg_tile_init, dim3(spk_grid_for(nf, 256, spk_num_cus() * 8)), dim3(256), 0, stream, (const float*)nullptr, a.gc, 3 * nf,
                       a.gmu_out, a.gmu, 3 * nf);
    SPK_LAUNCH_CHECK();
  }
  const int kpb = a.rb.n_rbf / 8 + 1;
  const bool geom = a.geom_only != 0, mu0 = a.mu_zero != 0;
#define SPK_TILE_CASE(Fv, Kv)                                                            \
  if (a.F == Fv && kpb == Kv) {                                                          \
    if (geom && mu0) return launch_tile_bwd<Fv, Kv, true, true>(a, stream);              \
    if (geom) return launch_tile_bwd<Fv, Kv, true, false>(a, stream);                    \
    if (mu0) return launch_tile_bwd<Fv, Kv, false, true>(a, stream);                     \
    return launch_tile_bwd<Fv, Kv, false, false>(a, stream);                             \
  }
  SPK_TILE_CASE(128, 3) SPK_TILE_CASE(128, 4) SPK_TILE_CASE(64, 3) SPK_TILE_CASE(64, 4)
#undef SPK_TILE_CASE
  spk_set_error("painn message tile kernel (backward): no instance for F=%d n_rbf=%d", a.F, a.rb.n_rbf);
  return SPK_ERR_ARG;
}

bool spk_painn_msg_tile_bwd_ok(const MsgArgs& a) {
  const int kpb = a.rb.n_rbf / 8 + 1;
  if (g_tile_mode < 0 || !(a.F == 128 || a.F == 64) || kpb < 3 || kpb > 4 || a.E < 32 || a.N * 3 * (int64_t)a.F >= (1LL << 31)) return false;
  // measured (profiles/r01_painn_tile_experiment.json): the geometry-only variant (2 waves / SIMD, no spills) beats the row
  // kernel by 25 %; the full variant needs the whole register file (1 wave / SIMD) and is 1.6 x slower than the row kernel
  return g_tile_mode > 0 || (a.geom_only && !a.skin_list);
}

// row-tile backward (sorted + symmetric list with row pointers; n_rbf + 1 <= 32; split path on)
extern "C" void spk_painn_set_rowtile(int32_t mode) { g_rowtile_mode = mode > 0 ? 1 : (mode < 0 ? -1 : 0); }

bool spk_painn_msg_rowtile_bwd_ok(const MsgArgs& a) {
  static const int env = [] { const char* e = getenv("SPK_PAINN_ROWTILE"); return e ? (e[0] == '1' ? 1 : -1) : 0; }();
  const int mode = g_rowtile_mode ? g_rowtile_mode : env;
  const int kpb = a.rb.n_rbf / 8 + 1;
  if (mode < 0 || !spk_get_split() || a.F != 128 || kpb < 3 || kpb > 4 || !a.rowptr || a.N * 3 * (int64_t)a.F >= (1LL << 30)) return false;
  return mode > 0 || a.E >= (1 << 19);
}

int spk_painn_msg_rowtile_bwd(const MsgArgs& a, hipStream_t stream) {
  const int kpb = a.rb.n_rbf / 8 + 1;
  if (a.skin_list) return kpb == 3 ? launch_rowtile_bwd<128, 3, true>(a, stream) : launch_rowtile_bwd<128, 4, true>(a, stream);
  return kpb == 3 ? launch_rowtile_bwd<128, 3, false>(a, stream) : launch_rowtile_bwd<128, 4, false>(a, stream);
}

// row-tile forward: sorted list with row pointers (no symmetry needed)
bool spk_painn_msg_rowtile_fwd_ok(const MsgArgs& a) {
  static const int env = [] { const char* e = getenv("SPK_PAINN_ROWTILE_FWD"); return e ? (e[0] == '1' ? 1 : -1) : 0; }();
  const int mode = g_rowtile_mode ? g_rowtile_mode : env;
  const int kpb = a.rb.n_rbf / 8 + 1;
  if (mode < 0 || !spk_get_split() || a.F != 128 || kpb < 3 || kpb > 4 || !a.rowptr || a.N * 3 * (int64_t)a.F >= (1LL << 30)) return false;
  return mode > 0 || a.E >= (1 << 19);
}

int spk_painn_msg_rowtile_fwd(const MsgArgs& a, hipStream_t stream) {
  const int kpb = a.rb.n_rbf / 8 + 1;
  if (a.skin_list) return kpb == 3 ? launch_rowtile_fwd<128, 3, true>(a, stream) : launch_rowtile_fwd<128, 4, true>(a, stream);
  return kpb == 3 ? launch_rowtile_fwd<128, 3, false>(a, stream) : launch_rowtile_fwd<128, 4, false>(a, stream);
}
