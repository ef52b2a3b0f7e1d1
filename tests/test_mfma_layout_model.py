"""CPU model of the lane <-> element maps that the gfx950 MFMA kernels rely on.

There is no GPU in the build container, so the index algebra of
``schnetpack_amd/csrc/spk_dense.hip`` / ``spk_cfconv.hip`` ("T-GEMM" convention, packed LDS
weight image, chained GEMMs that re-use the accumulator as the next B operand, the per-wave
transposition buffer and the segmented flush) is replayed here lane by lane with numpy, using
the documented operand layout of ``v_mfma_f32_32x32x2_f32``
(/opt/skills/guides/cdna_hip_programming.md section 3):

    A: lane l holds A[i = l & 31][k = l >> 5]      B: lane l holds B[k = l >> 5][j = l & 31]
    C/D: acc[r] of lane l  <->  D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]

The model result must equal the plain-math cfconv of the oracle.
"""
import numpy as np
import torch

from oracle import spk_oracle as O

LANES = np.arange(64)
HI = LANES >> 5
EL = LANES & 31


def mfma_32x32x2(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64,16] per-lane accumulators -> new acc."""
    A = np.zeros((32, 2), np.float64)
    B = np.zeros((2, 32), np.float64)
    A[EL, HI] = a
    B[HI, EL] = b
    D = A @ B  # [32, 32]
    out = acc.copy()
    for r in range(16):
        rows = (r & 3) + 8 * (r >> 2) + 4 * HI
        out[:, r] += D[rows, EL]
    return out


def stage_packed(w, nout, K, KB):
    """LDS image P[((t*KB+ug)*64+lane)*4+v] = W[32t+(lane&31)][8ug+4(lane>>5)+v] (0 beyond K)."""
    P = np.zeros(((nout // 32) * KB * 64, 4), np.float64)
    for s in range(P.shape[0]):
        lane = s & 63
        ug = (s >> 6) % KB
        t = (s >> 6) // KB
        row = 32 * t + (lane & 31)
        k0 = 8 * ug + 4 * (lane >> 5)
        for v in range(4):
            if k0 + v < K:
                P[s, v] = w[row, k0 + v]
    return P


def acc_rows(t):
    """feature index held by acc[r] of every lane for feature tile t: [64,16]"""
    r = np.arange(16)
    return 32 * t + (r[None, :] & 3) + 8 * (r[None, :] >> 2) + 4 * HI[:, None]


def cfconv_tile_model(h, phi_fn, fc, idx_i_tile, idx_j_tile, w1, b1, w2, b2, NF, n_rbf, y):
    """Replays one 32-edge tile of k_cfconv_mfma<NF, KPB, ..., BWD=false>.  phi_fn(k) -> [32]."""
    KPB = (n_rbf + 7) // 8
    NT, KB2 = NF // 32, NF // 8
    sW2 = stage_packed(w2, NF, NF, KB2)
    sW1 = stage_packed(w1, NF, n_rbf, KPB)
    nvalid = len(idx_i_tile)
    valid = EL < nvalid
    ec = np.where(valid, EL, nvalid - 1)
    j = idx_j_tile[ec]
    i = idx_i_tile[ec]
    fcl = np.where(valid, fc[ec], 0.0)
    myI = np.full(36, -1)
    myI[:32] = np.where(EL[:32] < nvalid, idx_i_tile[np.minimum(EL[:32], nvalid - 1)], -1)
    # radial basis per lane slot
    phi = np.zeros((KPB, 4, 64))
    for u in range(KPB):
        for v in range(4):
            k = 8 * u + 4 * HI + v
            for lane in range(64):
                if k[lane] < n_rbf:
                    phi[u, v, lane] = phi_fn(k[lane])[ec[lane]]
    # GEMM1 + ssp
    z = []
    for c in range(NT):
        acc = b1[acc_rows(c)].astype(np.float64)
        for u in range(KPB):
            wq = sW1[(c * KPB + u) * 64 + LANES]
            for v in range(4):
                acc = mfma_32x32x2(wq[:, v], phi[u, v], acc)
        z.append(np.log1p(np.exp(acc)) - np.log(2.0))
    # flush mask per half
    flush = np.zeros((2, 16), bool)
    for half in range(2):
        for k in range(16):
            flush[half, k] = (k == 15) or (myI[16 * half + k] != myI[16 * half + k + 1])
    # GEMM2 per output tile
    for t in range(NT):
        g = b2[acc_rows(t)].astype(np.float64)
        for c in range(NT):
            for q in range(4):
                wq = sW2[(t * KB2 + 4 * c + q) * 64 + LANES]
                for v in range(4):
                    g = mfma_32x32x2(wq[:, v], z[c][:, 4 * q + v], g)
        myT = np.zeros((32, 36))
        for q in range(4):
            col = 32 * t + 8 * q + 4 * HI
            for v in range(4):
                p = g[:, 4 * q + v] * fcl * h[j, col + v]
                myT[EL, 8 * q + 4 * HI + v] = p
        for lane in range(64):
            fl, half = lane & 31, lane >> 5
            acc = 0.0
            for k in range(16):
                acc += myT[16 * half + k, fl]
                if flush[half, k]:
                    ci = myI[16 * half + k]
                    if ci >= 0:
                        y[ci, 32 * t + fl] += acc
                    acc = 0.0


def test_mfma_model_matches_plain_matmul():
    rng = np.random.RandomState(0)
    A = rng.randn(32, 2)
    B = rng.randn(2, 32)
    acc = np.zeros((64, 16))
    out = mfma_32x32x2(A[EL, HI], B[HI, EL], acc)
    D = A @ B
    for r in range(16):
        np.testing.assert_allclose(out[:, r], D[(r & 3) + 8 * (r >> 2) + 4 * HI, EL])


def test_dense_tgemm_index_algebra():
    """k_dense_mfma: out[m][i] = sum_k w[i][k] x[m][k] (+ the transposed-weight variant)."""
    rng = np.random.RandomState(1)
    M, K, NO = 40, 24, 64
    x = rng.randn(M, K)
    w = rng.randn(NO, K)
    for trans in (False, True):
        wt = w if not trans else rng.randn(K, NO)  # trans: w is [KC, NW] and A[i][kk] = w[kk][i]
        out = np.zeros((M, NO))
        for mt in range((M + 31) // 32):
            for t in range(NO // 32):
                m = mt * 32 + EL
                mc = np.minimum(m, M - 1)
                acc = np.zeros((64, 16))
                for ug in range(K // 8):
                    kk0 = 8 * ug + 4 * HI
                    for v in range(4):
                        bv = x[mc, kk0 + v]
                        av = wt[32 * t + EL, kk0 + v] if not trans else wt[kk0 + v, 32 * t + EL]
                        acc = mfma_32x32x2(av, bv, acc)
                for lane in range(64):
                    if m[lane] < M:
                        for q in range(4):
                            for v in range(4):
                                out[m[lane], 32 * t + 8 * q + 4 * HI[lane] + v] = acc[lane, 4 * q + v]
        expect = x @ (wt.T if not trans else wt)
        np.testing.assert_allclose(out, expect, rtol=1e-12, atol=1e-12)


def test_cfconv_tile_algebra_matches_oracle():
    torch.manual_seed(0)
    NF, n_rbf, N = 64, 20, 11
    rng = np.random.RandomState(2)
    # sorted idx_i with ragged degrees, 70 edges -> 3 tiles (last one partial)
    deg = [9, 0, 13, 5, 7, 1, 12, 6, 8, 4, 5]
    idx_i = np.repeat(np.arange(N), deg)
    E = len(idx_i)
    idx_j = rng.randint(0, N, size=E)
    d = rng.uniform(0.7, 5.5, size=E)
    h = rng.randn(N, NF)
    w1 = rng.randn(NF, n_rbf) * 0.3
    b1 = rng.randn(NF) * 0.1
    w2 = rng.randn(NF, NF) * 0.1
    b2 = rng.randn(NF) * 0.1
    off, wid = O.gaussian_rbf_params(n_rbf, 5.0)
    phi = O.gaussian_rbf(torch.from_numpy(d), off.double(), wid.double()).numpy()  # [E, n_rbf]
    fc = O.cosine_cutoff(torch.from_numpy(d), 5.0).numpy()
    y = np.zeros((N, NF))
    for tile in range((E + 31) // 32):
        sl = slice(32 * tile, min(E, 32 * tile + 32))
        cfconv_tile_model(h, lambda k, sl=sl: phi[sl, k], fc[sl], idx_i[sl], idx_j[sl], w1, b1, w2,
                          b2, NF, n_rbf, y)
    # plain math
    td = torch.from_numpy
    W = O.dense(td(phi), td(w1), td(b1), O.shifted_softplus)
    W = O.dense(W, td(w2), td(b2)) * td(fc)[:, None]
    expect = O.scatter_add(td(h)[td(idx_j)] * W, td(idx_i), N).numpy()
    np.testing.assert_allclose(y, expect, rtol=1e-10, atol=1e-10)
