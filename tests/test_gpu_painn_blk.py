"""Block kernels of the PaiNN message for large lists (``csrc/spk_painn_blk.hip``; representation/painn.py:31-67): the device-built
block plan against a numpy restatement of its definition, the forward / backward launches through the C ABI against the float64
autograd oracle of the message, and the whole force call of a periodic water box with the block path switched on against the CPU
oracle (the reference's algorithm).  Tolerance 1e-5 relative (north_star), relative = max|a-b| / max|b|."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S
from test_gpu_ops import _msg_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    return torch.device("cuda:0")


def _water(n_side):
    b = S.water_box(n_side)
    r = O.pairwise_vectors(b["R"], b["idx_i"], b["idx_j"], b["offsets"])
    return b, r


def _graph(name):
    if name == "water6":
        b, r = _water(6)
        return r, b["idx_i"], b["idx_j"], b["Z"].shape[0]
    if name == "aspirin":
        b = S.molecule_batch("aspirin", 5, seed=6)      # 105 atoms: the last group of 16 is partial
        return O.pairwise_vectors(b["R"], b["idx_i"], b["idx_j"], b["offsets"]), b["idx_i"], b["idx_j"], b["Z"].shape[0]
    rb_ = S.random_graph_batch(90, 70, seed=8, sort=True)     # dense rows: 70 neighbours, several tiles per atom, not symmetric
    return rb_["r_ij"], rb_["idx_i"], rb_["idx_j"], rb_["Z"].shape[0]


@pytest.mark.parametrize("graph,cap", [("water6", 0), ("water6", 64), ("aspirin", 0), ("random", 0), ("random", 80), ("random", 8)])
def test_block_plan_matches_its_definition(dev, graph, cap):
    """Sub-blocks, ascending unique neighbour lists, local indices and atom-aligned 16-edge tiles of ``spk_blocks_build``."""
    from schnetpack_amd import ops
    r, idx_i, idx_j, N = _graph(graph)
    plan = ops.EdgePlan(idx_i.to(dev), idx_j.to(dev), N, r.to(dev))
    ok, max_u, n_tiles = plan.build_blocks(20, 128, cap)
    rowptr = plan.rowptr.cpu().numpy().astype(np.int64)
    jj = idx_j.numpy()
    deg = np.diff(rowptr)
    if graph == "random" and cap == 8:
        assert not ok          # single atoms with 70 neighbours do not fit 8 rows: the plan refuses, the row kernels stay
        return
    assert ok
    bufs = {k: v.cpu().numpy() for k, v in plan._block_bufs.items()}
    jl = bufs["jl"].view(np.uint16)
    from schnetpack_amd import _lib
    BA = int(_lib.lib().spk_blocks_group_atoms())
    ng = (N + BA - 1) // BA
    eff_cap = cap if cap > 0 else plan.blocks.cap
    seen_max = 0
    for g in range(ng):
        nsub = int(bufs["sub_n"][g])
        assert nsub in (1, 2, 4, 8, 16) and nsub <= BA
        w = BA // nsub
        for s in range(nsub):
            lo = min(BA * g + s * w, N); hi = min(lo + w, N)
            e0, e1 = rowptr[lo], rowptr[hi]
            want = np.unique(jj[e0:e1])
            U = int(bufs["sub_u"][BA * g + s])
            assert U == len(want) and U <= eff_cap
            seen_max = max(seen_max, U)
            assert np.array_equal(bufs["uniq"][e0:e0 + U], want)
            assert np.array_equal(want[jl[e0:e1]], jj[e0:e1])
        if nsub > 1:   # a coarser split would not have fitted
            w2 = 2 * w
            too_big = False
            for s in range(BA // w2):
                lo = min(BA * g + s * w2, N); hi = min(lo + w2, N)
                too_big |= len(np.unique(jj[rowptr[lo]:rowptr[hi]])) > eff_cap or (rowptr[hi] - rowptr[lo]) > 2048
            assert too_big
    assert max_u == seen_max
    # block descriptors: the sub-blocks of all groups in atom order
    nb = int(plan.blocks.n_blocks)
    desc = bufs["blk_desc"][:4 * nb].reshape(-1, 4)
    want_desc = []
    for g in range(ng):
        nsub = int(bufs["sub_n"][g]); w = BA // nsub
        for s in range(nsub):
            lo = min(BA * g + s * w, N); hi = min(lo + w, N)
            want_desc.append((lo, hi - lo, rowptr[lo], int(bufs["sub_u"][BA * g + s])))
    assert nb == len(want_desc) and np.array_equal(desc, np.array(want_desc, dtype=desc.dtype))
    tiles = (deg + 15) // 16
    assert n_tiles == int(tiles.sum())
    t0 = np.concatenate([[0], np.cumsum(tiles)])
    assert np.array_equal(bufs["atom_tile0"][:N + 1], t0)
    info = bufs["tile_info"][:2 * n_tiles].reshape(-1, 2)
    for a in range(N):
        for k in range(int(tiles[a])):
            assert info[t0[a] + k, 0] == rowptr[a] + 16 * k and info[t0[a] + k, 1] == min(16, deg[a] - 16 * k)


@pytest.mark.parametrize("graph,cap", [("water6", 0), ("water6", 64), ("aspirin", 0), ("random", 0), ("random", 80)])
@pytest.mark.parametrize("F,n_rbf", [(128, 20), (64, 16), (128, 25)])
def test_block_message_forward_backward(dev, graph, cap, F, n_rbf):
    """spk_painn_message_{fwd,bwd}_f32 with a block plan on the graph (spk_painn_set_block(1)) against the float64 oracle; the
    backward (pass T needs the reversed edge) on the symmetric lists only -- an asymmetric list keeps the simple kernel there."""
    from schnetpack_amd import _lib, ops
    r, idx_i, idx_j, N = _graph(graph)
    g = torch.Generator().manual_seed(21)
    c = torch.randn(N, 3 * F, generator=g); q = torch.randn(N, F, generator=g); mu = torch.randn(N, 3, F, generator=g)
    wf = torch.randn(3 * F, n_rbf, generator=g) * 0.3; bf = torch.randn(3 * F, generator=g) * 0.1
    gq = torch.randn(N, F, generator=g); gmu = torch.randn(N, 3, F, generator=g)
    qo, muo, gco, gmuo, gro = _msg_oracle(c, q, mu, r, idx_i, idx_j, wf, bf, N, F, gq, gmu)
    plan = ops.EdgePlan(idx_i.to(dev), idx_j.to(dev), N, r.to(dev))
    ok, max_u, n_tiles = plan.build_blocks(n_rbf, F, cap)
    assert ok and plan._graph.blocks
    off, w = O.gaussian_rbf_params(n_rbf, 5.0)
    offd, wd = off.to(dev), w.to(dev)
    rb = ops.radial_struct(_lib.SPK_RBF_GAUSSIAN, n_rbf, offd, wd, 5.0)
    D = lambda t: t.to(dev).contiguous()
    cd, qd, mud, rd, wfd, bfd, gqd, gmud = map(D, (c, q, mu, r, wf, bf, gq, gmu))
    L = _lib.lib()
    L.spk_profile_enable(1); L.spk_profile_report()
    try:
        L.spk_painn_set_block(1)
        q_out = torch.full((N, F), float("nan"), device=dev); mu_out = torch.full((N, 3, F), float("nan"), device=dev)
        _lib.check(L.spk_painn_message_fwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(qd), _lib.fptr(mud), _lib.fptr(rd),
                                               _lib.fptr(wfd), _lib.fptr(bfd), F, _lib.fptr(q_out), _lib.fptr(mu_out), _lib.stream()))
        assert rel_err(q_out.cpu(), qo) < TOL and rel_err(mu_out.cpu(), muo) < TOL
        q2 = torch.empty_like(q_out); mu2 = torch.empty_like(mu_out)
        _lib.check(L.spk_painn_message_fwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(qd), _lib.fptr(mud), _lib.fptr(rd),
                                               _lib.fptr(wfd), _lib.fptr(bfd), F, _lib.fptr(q2), _lib.fptr(mu2), _lib.stream()))
        assert torch.equal(q2, q_out) and torch.equal(mu2, mu_out)          # no atomics: bit-reproducible
        if plan.symmetric:
            gc = torch.full((N, 3 * F), float("nan"), device=dev); gmu_in = torch.full((N, 3, F), float("nan"), device=dev)
            gr = torch.zeros(r.shape[0], 3, device=dev)
            _lib.check(L.spk_painn_message_bwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(mud), _lib.fptr(gqd), _lib.fptr(gmud),
                                                   _lib.fptr(rd), _lib.fptr(wfd), _lib.fptr(bfd), F, _lib.fptr(gc), _lib.fptr(gmu_in), _lib.fptr(gr),
                                                   _lib.stream()))
            assert rel_err(gc.cpu(), gco) < TOL and rel_err(gmu_in.cpu(), gmuo) < TOL and rel_err(gr.cpu(), gro) < TOL
        rep = L.spk_profile_report().decode()
        assert "painn_msg_fwd_blk" in rep and ("painn_msg_bwd_blk_G" in rep) == plan.symmetric, rep      # the block kernels ran, nothing else
        assert "painn_msg_fwd_row" not in rep and "painn_msg_fwd_tile" not in rep
    finally:
        L.spk_painn_set_block(0)
        L.spk_profile_enable(0)


@pytest.mark.parametrize("radial", ["gaussian", "bessel"])
def test_water_box_force_call_through_the_block_kernels(dev, radial):
    """The whole PaiNN force call (forward with the mu = 0 first interaction, backward with the geometry-only first interaction)
    of a small periodic water box with the block path forced on, against the CPU oracle in float64."""
    from schnetpack_amd import _lib, model as M
    L = _lib.lib()
    b = S.water_box(7)       # 1 029 atoms, ~55 k pairs
    rep_p = O.init_painn_params(radial=radial)
    head_p = O.init_atomwise_params(128, seed=1)
    ref = O.energy_and_forces("painn", rep_p, head_p, b, 3, dtype=torch.float64)
    m = M.build_model("painn", radial=radial)
    M.load_reference_params(m, rep_p, head_p)
    m = m.to(dev).eval()
    os.environ["SPK_BLOCKS"] = "1"
    L.spk_profile_enable(1); L.spk_profile_report()
    try:
        L.spk_painn_set_block(1)
        out = m(M.batch_to_inputs(b, dev))
        rep = L.spk_profile_report().decode()
    finally:
        L.spk_painn_set_block(0)
        L.spk_profile_enable(0)
        os.environ.pop("SPK_BLOCKS", None)
    assert "painn_msg_fwd_blk_mu0" in rep and "painn_msg_bwd_blk_T" in rep and "painn_msg_bwd_blk_G_mu0" in rep and "painn_blk_prep" in rep, rep
    assert rel_err(out["energy"].cpu(), ref["energy"]) < TOL
    assert rel_err(out["forces"].cpu(), ref["forces"]) < TOL
