"""GPU parity tests of the individual C-ABI entry points against the CPU oracle.

Tolerance: max|hip - oracle| / max|oracle| <= 1e-5 (fp32; BASELINE.json north_star) unless the
op is exact (index work, gather), where the comparison is bit-exact.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(params=["mfma", "simple", "directed", "mol"])
def variant(request):
    """mfma = default dispatch (pair kernels on symmetric lists), mol = group-local LDS-accumulating pair
    kernel (experiment, block-diagonal lists), directed = MFMA kernel with one filter per directed edge,
    simple = straightforward cross-check kernels."""
    from schnetpack_amd import _lib
    _lib.set_variant({"simple": _lib.VARIANT_SIMPLE, "directed": _lib.VARIANT_MFMA_DIRECTED,
                      "mol": _lib.VARIANT_MFMA_MOL, "mfma": _lib.VARIANT_AUTO}[request.param])
    yield request.param
    _lib.set_variant(_lib.VARIANT_AUTO)


def test_library_loads_on_gfx950(dev):
    from schnetpack_amd import _lib
    assert _lib.lib().spk_version() >= 100
    info = _lib.device_info()
    assert info["wavefront"] == 64
    assert info["compute_units"] >= 64
    assert info["gfx"] == 950


def test_cpu_tensors_fail_loudly(dev):
    from schnetpack_amd import ops
    from schnetpack_amd._lib import SpkHipError
    with pytest.raises(SpkHipError):
        ops.scatter_add(torch.ones(3, 2), torch.tensor([0, 0, 1]), 2)


# ----------------------------------------------------------------------------- edge plan
def test_edge_plan_flags_and_rowptr(dev):
    from schnetpack_amd import ops
    from schnetpack_amd._lib import SpkHipError
    b = S.molecule_batch("aspirin", 3, seed=1)
    r = O.pairwise_vectors(b["R"], b["idx_i"], b["idx_j"], b["offsets"])
    N = b["Z"].shape[0]
    plan = ops.EdgePlan(b["idx_i"].to(dev), b["idx_j"].to(dev), N, r.to(dev), want_groups=True)
    assert plan.sorted and plan.symmetric
    counts = torch.bincount(b["idx_i"], minlength=N)
    expect = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]).int()
    assert torch.equal(plan.rowptr.cpu(), expect)
    # reverse-edge map is an involution that swaps (i, j) and negates r; half list = canonical edges
    rev = plan.rev[: plan.n_edges].long().cpu()
    assert torch.equal(rev[rev], torch.arange(plan.n_edges))
    assert torch.equal(b["idx_i"][rev], b["idx_j"]) and torch.equal(b["idx_j"][rev], b["idx_i"])
    assert torch.equal(r[rev], -r)
    half = plan.half.long().cpu()
    assert half.shape[0] * 2 == plan.n_edges and bool((rev[half] > half).all())
    assert bool((half[1:] > half[:-1]).all())
    # block-diagonal groups: 3 molecules of 21 atoms, group-aligned tiles
    atom0, pair0, tile0, max_atoms, n_tiles, max_pairs = plan.groups
    assert atom0.cpu().tolist() == [0, 21, 42, 63] and max_atoms == 21
    assert max_pairs == max(int(pair0[k + 1] - pair0[k]) for k in range(3))
    # position of the pair of every directed edge (molecule-resident kernels)
    ep = plan.edge_pair.long().cpu()
    assert torch.equal(ep[half], torch.arange(half.shape[0])) and torch.equal(ep[rev[half]], torch.arange(half.shape[0]))
    p0 = pair0.cpu().tolist()
    assert p0[0] == 0 and p0[-1] == half.shape[0]
    hi_atoms = b["idx_i"][half]
    for gidx in range(3):
        seg = hi_atoms[p0[gidx]:p0[gidx + 1]]
        assert bool(((seg >= 21 * gidx) & (seg < 21 * (gidx + 1))).all())
    assert tile0.cpu().tolist() == np.cumsum([0] + [-(-(p0[k + 1] - p0[k]) // 32) for k in range(3)]).tolist()
    assert n_tiles == int(tile0[-1])
    # drop one edge -> asymmetric; shuffle -> unsorted
    plan2 = ops.EdgePlan(b["idx_i"][1:].to(dev), b["idx_j"][1:].to(dev), N, r[1:].to(dev))
    assert plan2.sorted and not plan2.symmetric
    perm = torch.randperm(b["idx_i"].shape[0], generator=torch.Generator().manual_seed(0))
    plan3 = ops.EdgePlan(b["idx_i"][perm].to(dev), b["idx_j"][perm].to(dev), N, r[perm].to(dev))
    assert not plan3.sorted and not plan3.symmetric
    bad = b["idx_j"].clone()
    bad[5] = N
    with pytest.raises(SpkHipError):
        ops.EdgePlan(b["idx_i"].to(dev), bad.to(dev), N, None)
    # empty list
    e = torch.zeros(0, dtype=torch.long, device=dev)
    plan4 = ops.EdgePlan(e, e, 4, None)
    assert plan4.sorted and plan4.n_edges == 0
    assert torch.equal(plan4.rowptr.cpu(), torch.zeros(5, dtype=torch.int32))


# ----------------------------------------------------------------------------- scatter_add
def test_scatter_add_golden_known_answers(dev):
    from schnetpack_amd.nn import scatter_add
    ka = np.load(GOLDEN + "/nn_known_answers.npz")
    x = torch.from_numpy(ka["scat_x"]).to(dev)
    idx = torch.from_numpy(ka["scat_idx"]).to(dev)
    y0 = scatter_add(x, idx, dim_size=7)
    assert rel_err(y0.cpu(), torch.from_numpy(ka["scat_y0"])) < TOL
    xt = x.permute(1, 0, 2).contiguous()
    y1 = scatter_add(xt, idx, dim_size=7, dim=1)
    assert rel_err(y1.cpu(), torch.from_numpy(ka["scat_y1"])) < TOL


@pytest.mark.parametrize("shape,dim,n,sorted_idx", [
    ((300, 128), 0, 40, True), ((300, 128), 0, 40, False), ((257, 3, 64), 0, 33, True),
    ((5, 200, 7), 1, 19, False), ((5, 200, 8), 1, 19, True), ((1000, 1), 0, 77, True),
    ((64,), 0, 5, True), ((0, 16), 0, 4, True)])
def test_scatter_add_matches_oracle(dev, shape, dim, n, sorted_idx):
    from schnetpack_amd.nn import scatter_add
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g)
    idx = torch.randint(0, n, (shape[dim],), generator=g)
    if sorted_idx:
        idx = idx.sort().values
    y = scatter_add(x.to(dev), idx.to(dev), n, dim)
    ref = O.scatter_add(x, idx, n, dim)
    assert y.shape == ref.shape and y.dtype == torch.float32
    if ref.numel():
        assert rel_err(y.cpu(), ref) < TOL


def test_scatter_add_first_and_second_order_autograd(dev):
    """backward = gather, double backward = scatter again (Forces with create_graph=True)."""
    from schnetpack_amd.nn import scatter_add
    g = torch.Generator().manual_seed(4)
    x = torch.randn(50, 6, generator=g)
    w = torch.randn(9, 6, generator=g)
    idx = torch.randint(0, 9, (50,), generator=g).sort().values

    def run(x, w, idx, fn):
        x = x.clone().requires_grad_(True)
        w = w.clone().requires_grad_(True)
        y = fn(x * x, idx, 9)
        (gx,) = torch.autograd.grad((y * w).sum(), [x], create_graph=True)
        (gw,) = torch.autograd.grad((gx ** 2).sum(), [w])
        return gx.detach().cpu(), gw.cpu()

    gx_h, gw_h = run(x.to(dev), w.to(dev), idx.to(dev), scatter_add)
    gx_o, gw_o = run(x, w, idx, O.scatter_add)
    assert rel_err(gx_h, gx_o) < TOL and rel_err(gw_h, gw_o) < TOL


def test_scatter_add_large_bitwise_stable_when_sorted(dev):
    """At bench scale: segmented path is deterministic (two runs bit-identical) and agrees with
    the atomic path to round-off."""
    from schnetpack_amd import ops
    E, C, N = 77928, 128, 5376
    g = torch.Generator().manual_seed(0)
    idx = torch.randint(0, N, (E,), generator=g).sort().values.to(dev)
    x = torch.randn(E, C, generator=g).to(dev)
    rp = ops.segment_rowptr(idx, N)
    assert rp is not None
    y1 = ops._scatter_raw(x, idx, N, 0, rp)
    y2 = ops._scatter_raw(x, idx, N, 0, rp)
    y3 = ops._scatter_raw(x, idx, N, 0, None)
    assert torch.equal(y1, y2)
    assert rel_err(y3, y1) < TOL
    # linearity: scatter(a x) == a scatter(x)
    assert rel_err(ops._scatter_raw(2.0 * x, idx, N, 0, rp), 2.0 * y1) < 1e-6


# ----------------------------------------------------------------------------- radial / cutoff
def test_radial_cutoff_known_answers(dev):
    from schnetpack_amd.nn import BesselRBF, CosineCutoff, GaussianRBF
    ka = np.load(GOLDEN + "/nn_known_answers.npz")
    d = torch.from_numpy(ka["d"]).to(dev)
    d2 = torch.from_numpy(ka["d2"]).to(dev)
    g = GaussianRBF(20, 5.0).to(dev).eval()
    assert rel_err(g(d).cpu(), torch.from_numpy(ka["gauss20_5"])) < TOL
    g2 = GaussianRBF(5, 1.5, start=0.5).to(dev).eval()
    y = g2(d2)
    assert y.shape == (3, 2, 5)
    assert rel_err(y.cpu(), torch.from_numpy(ka["gauss5_1p5_start0p5"])) < TOL
    bz = BesselRBF(20, 5.0).to(dev).eval()
    assert rel_err(bz(d).cpu(), torch.from_numpy(ka["bessel20_5"])) < TOL
    assert rel_err(BesselRBF(7, 3.0).to(dev).eval()(d2).cpu(), torch.from_numpy(ka["bessel7_3"])) < TOL
    c = CosineCutoff(5.0).to(dev).eval()
    assert rel_err(c(d).cpu(), torch.from_numpy(ka["cos5"])) < TOL
    c2 = CosineCutoff(1.8).to(dev).eval()
    out = c2(d2).cpu()
    assert rel_err(out, torch.from_numpy(ka["cos1p8"])) < TOL
    assert float(out[2, 0]) == 0.0 and float(out[2, 1]) == 0.0  # zero beyond the cutoff
    assert float(c2.cutoff) == pytest.approx(1.8)


@pytest.mark.parametrize("kind", ["gaussian", "bessel"])
def test_radial_cutoff_backward(dev, kind):
    from schnetpack_amd.nn import BesselRBF, CosineCutoff, GaussianRBF
    g = torch.Generator().manual_seed(1)
    d = (torch.rand(200, generator=g) * 5.5 + 0.3)
    gphi = torch.randn(200, 20, generator=g)
    gfc = torch.randn(200, generator=g)
    rb = (GaussianRBF(20, 5.0) if kind == "gaussian" else BesselRBF(20, 5.0)).to(dev).eval()
    cf = CosineCutoff(5.0).to(dev).eval()
    dd = d.to(dev).requires_grad_(True)
    loss = (rb(dd) * gphi.to(dev)).sum() + (cf(dd) * gfc.to(dev)).sum()
    (gd,) = torch.autograd.grad(loss, [dd])
    dc = d.clone().double().requires_grad_(True)
    if kind == "gaussian":
        off, w = O.gaussian_rbf_params(20, 5.0)
        phi = O.gaussian_rbf(dc, off.double(), w.double())
    else:
        phi = O.bessel_rbf(dc, O.bessel_rbf_params(20, 5.0).double())
    lo = (phi * gphi.double()).sum() + (O.cosine_cutoff(dc, 5.0) * gfc.double()).sum()
    (gd_o,) = torch.autograd.grad(lo, [dc])
    assert rel_err(gd.cpu(), gd_o) < TOL


# ----------------------------------------------------------------------------- dense
@pytest.mark.parametrize("m,k,n,act", [(5376, 128, 128, "ssp"), (100, 128, 128, None), (77, 256, 128, "silu"),
                                       (33, 128, 384, None), (640, 128, 64, "silu"), (50, 20, 128, "ssp"),
                                       (31, 64, 1, None), (1, 128, 128, "ssp")])
def test_dense_forward_backward(dev, variant, m, k, n, act):
    from schnetpack_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g) * 0.1
    gy = torch.randn(m, n, generator=g)
    actf = {None: None, "ssp": O.shifted_softplus, "silu": O.silu}[act]
    xo = x.clone().double().requires_grad_(True)
    yo = O.dense(xo, w.double(), b.double(), actf)
    (gxo,) = torch.autograd.grad((yo * gy.double()).sum(), [xo])
    xd = x.to(dev).requires_grad_(True)
    wd, bd, gyd = w.to(dev), b.to(dev), gy.to(dev)  # keep the device buffers alive across raw calls
    y = ops.dense(xd, wd, bd, act)
    assert rel_err(y.detach().cpu(), yo.detach()) < TOL
    # first-order input gradient through the HIP backward kernel
    from schnetpack_amd import _lib
    a = ops._ACT_IDS[act]
    _, pre = ops.dense_raw(xd.detach(), wd, bd, a, want_pre=True)
    dx = torch.empty(m, k, device=dev)
    _lib.check(_lib.lib().spk_dense_bwd_input_f32(_lib.fptr(gyd), _lib.fptr(pre), _lib.fptr(wd), None,
                                                   _lib.fptr(dx), m, k, n, a, _lib.stream()))
    torch.cuda.synchronize()
    assert rel_err(dx.cpu(), gxo) < TOL
    # autograd (differentiable composite backward)
    (gx,) = torch.autograd.grad((y * gyd).sum(), [xd])
    assert rel_err(gx.cpu(), gxo) < TOL


@pytest.mark.parametrize("m,k,n,act", [(168, 128, 128, "ssp"), (2436, 20, 128, "ssp"), (2436, 128, 128, None), (168, 256, 128, "silu"),
                                       (45, 128, 384, "silu"), (640, 20, 1152, None), (168, 128, 64, "ssp"), (1, 64, 32, "silu")])
def test_dense_on_value_tangent_pairs(dev, m, k, n, act):
    """spk_dense_dual_f32 (one launch per Dense layer of the force-matching engine's (value, tangent) pairs) in its three modes against
    float64 formulas: forward pair (activation or cutoff row scale), tangent alone, reverse of the pair -- nn/base.py:52-55 under the
    double differentiation of atomistic/response.py:59-68."""
    from schnetpack_amd import _lib, ops
    L = _lib.lib()
    assert L.spk_dense_dual_supported(m, k, n)
    g = torch.Generator().manual_seed(11)
    xv = torch.randn(m, k, generator=g); xt = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g) * 0.1
    fc = torch.rand(m, generator=g); fc1 = torch.randn(m, generator=g)
    a = ops._ACT_IDS[act]
    D = lambda t: t.to(dev).contiguous()
    f0 = {None: lambda z: z, "ssp": O.shifted_softplus, "silu": O.silu}[act]

    def derivs(z):      # act', act'' in float64 by autograd
        z = z.clone().requires_grad_(True)
        (d1,) = torch.autograd.grad(f0(z).sum(), [z], create_graph=True)
        if act is None:
            return torch.ones_like(z), torch.zeros_like(z)
        (d2,) = torch.autograd.grad(d1.sum(), [z])
        return d1.detach(), d2

    xvd, xtd, wd, bd, fcd, fc1d = map(D, (xv, xt, w, b, fc, fc1))
    pv_o = xv.double() @ w.double().T + b.double(); pt_o = xt.double() @ w.double().T
    d1, _ = derivs(pv_o)

    def run(**kw):
        d = _lib.DenseDualT()
        keep = []
        for key, val in kw.items():
            if isinstance(val, torch.Tensor):
                keep.append(val)
                setattr(d, key, _lib.fptr(val))
            else:
                setattr(d, key, val)
        _lib.check(L.spk_dense_dual_f32(ctypes.byref(d), _lib.stream()))
        torch.cuda.synchronize()

    new = lambda *shape: torch.full(shape, float("nan"), device=dev)
    # forward pair with the activation
    yv, yt, pv, pt = new(m, n), new(m, n), new(m, n), new(m, n)
    run(x_v=xvd, x_t=xtd, w=wd, b=bd, y_v=yv, y_t=yt, pre_v=pv, pre_t=pt, m=m, k_in=k, n_out=n, act=a, mode=0, trans=0)
    assert rel_err(pv.cpu(), pv_o) < TOL and rel_err(pt.cpu(), pt_o) < TOL
    assert rel_err(yv.cpu(), f0(pv_o)) < TOL and rel_err(yt.cpu(), d1 * pt_o) < TOL
    if act is None:      # cutoff row scale with its derivative (linear layers)
        yv, yt = new(m, n), new(m, n)
        run(x_v=xvd, x_t=xtd, w=wd, b=bd, fc=fcd, fc1=fc1d, y_v=yv, y_t=yt, m=m, k_in=k, n_out=n, act=a, mode=0, trans=0)
        assert rel_err(yv.cpu(), pv_o * fc.double()[:, None]) < TOL
        assert rel_err(yt.cpu(), pt_o * fc.double()[:, None] + pv_o * fc1.double()[:, None]) < TOL
    # tangent alone, at saved pre-activations
    sv = torch.randn(m, n, generator=g); svd = D(sv)
    s1, s2 = derivs(sv.double())
    yt, pt = new(m, n), new(m, n)
    run(x_t=xtd, w=wd, pre_v_in=svd, y_t=yt, pre_t=pt, m=m, k_in=k, n_out=n, act=a, mode=1, trans=0)
    assert rel_err(pt.cpu(), pt_o) < TOL and rel_err(yt.cpu(), s1 * pt_o) < TOL
    # the transposed product (input-gradient form: x [m, n] times w [n, k]) in tangent mode and as the reverse of the pair
    gz = torch.randn(m, n, generator=g); hz = torch.randn(m, n, generator=g)
    av = torch.randn(m, k, generator=g); at = torch.randn(m, k, generator=g)
    gzd, hzd, avd, atd = map(D, (gz, hz, av, at))
    a1, a2 = derivs(av.double())
    Gz = gz.double() @ w.double(); Hz = hz.double() @ w.double()
    yt = new(m, k)
    run(x_t=gzd, w=wd, pre_v_in=avd, y_t=yt, m=m, k_in=n, n_out=k, act=a, mode=1, trans=1)
    assert rel_err(yt.cpu(), a1 * Gz) < TOL
    yv, yt = new(m, k), new(m, k)
    run(x_v=gzd, x_t=hzd, w=wd, pre_v_in=avd, pre_t_in=atd, y_v=yv, y_t=yt, m=m, k_in=n, n_out=k, act=a, mode=2, trans=1)
    assert rel_err(yv.cpu(), Gz * a1 + Hz * a2 * at.double()) < TOL and rel_err(yt.cpu(), Hz * a1) < TOL
    # the forward pair beyond the one-tile-per-workgroup budget: grid-stride kernel (pair rows of larger batches), a partial last row tile
    mb = (40000 if n <= 384 else 8000) + 7
    assert not L.spk_dense_dual_supported(mb, k, n) and L.spk_dense_dual_fwd_supported(mb, k, n)
    xb = torch.randn(2, mb, k, generator=g); fcb = torch.rand(mb, generator=g); fc1b = torch.randn(mb, generator=g)
    xbd, fcbd, fc1bd = D(xb), D(fcb), D(fc1b)
    pvb = xb[0].double() @ w.double().T + b.double(); ptb = xb[1].double() @ w.double().T
    yv, yt, pv, pt = new(mb, n), new(mb, n), new(mb, n), new(mb, n)
    run(x_v=xbd[0], x_t=xbd[1], w=wd, b=bd, y_v=yv, y_t=yt, pre_v=pv, pre_t=pt, m=mb, k_in=k, n_out=n, act=a, mode=0, trans=0)
    d1b, _ = derivs(pvb)
    assert rel_err(pv.cpu(), pvb) < TOL and rel_err(pt.cpu(), ptb) < TOL and rel_err(yv.cpu(), f0(pvb)) < TOL and rel_err(yt.cpu(), d1b * ptb) < TOL
    if act is None:
        yv, yt = new(mb, n), new(mb, n)
        run(x_v=xbd[0], x_t=xbd[1], w=wd, b=bd, fc=fcbd, fc1=fc1bd, y_v=yv, y_t=yt, m=mb, k_in=k, n_out=n, act=a, mode=0, trans=0)
        assert rel_err(yv.cpu(), pvb * fcb.double()[:, None]) < TOL
        assert rel_err(yt.cpu(), ptb * fcb.double()[:, None] + pvb * fc1b.double()[:, None]) < TOL
    # refused: shapes outside the MFMA tiles, a missing saved pre-activation
    bad = _lib.DenseDualT()
    bad.x_t = _lib.fptr(xtd); bad.w = _lib.fptr(wd); bad.y_t = _lib.fptr(yt); bad.m = m; bad.k_in = k; bad.n_out = n; bad.mode = 1
    assert L.spk_dense_dual_f32(ctypes.byref(bad), _lib.stream()) != 0
    assert not L.spk_dense_dual_supported(m, 18, n)


def test_dense_residual_and_preactivation(dev, variant):
    from schnetpack_amd import ops, _lib
    g = torch.Generator().manual_seed(8)
    x = torch.randn(70, 128, generator=g)
    w = torch.randn(128, 128, generator=g) / 11.0
    b = torch.randn(128, generator=g)
    r = torch.randn(70, 128, generator=g)
    xd, wd, bd, rd = x.to(dev), w.to(dev), b.to(dev), r.to(dev)
    y, pre = ops.dense_raw(xd, wd, bd, _lib.SPK_ACT_SSP, res=rd, want_pre=True)
    torch.cuda.synchronize()
    pre_o = O.dense(x.double(), w.double(), b.double())
    assert rel_err(pre.cpu(), pre_o) < TOL
    assert rel_err(y.cpu(), O.shifted_softplus(pre_o) + r.double()) < TOL


# ----------------------------------------------------------------------------- cfconv
def _cfconv_oracle(h, r, idx_i, idx_j, p, n_atoms, kind, gy):
    """fp64 oracle: y, and (gh, gr) for upstream gradient gy"""
    h = h.double().requires_grad_(True)
    r = r.double().requires_grad_(True)
    d = torch.sqrt((r * r).sum(1))
    if kind == "gaussian":
        off, w = O.gaussian_rbf_params(p["n_rbf"], 5.0)
        phi = O.gaussian_rbf(d, off.double(), w.double())
    else:
        phi = O.bessel_rbf(d, O.bessel_rbf_params(p["n_rbf"], 5.0).double())
    fc = O.cosine_cutoff(d, 5.0)
    W = O.dense(phi, p["w1"].double(), p["b1"].double(), O.shifted_softplus)
    W = O.dense(W, p["w2"].double(), p["b2"].double()) * fc[:, None]
    y = O.scatter_add(h[idx_j] * W, idx_i, n_atoms)
    gh, gr = torch.autograd.grad((y * gy.double()).sum(), [h, r])
    return y.detach(), gh, gr


def _cfconv_hip(dev, h, r, idx_i, idx_j, p, n_atoms, kind, gy, transposed=False):
    from schnetpack_amd import _lib, ops
    plan = ops.EdgePlan(idx_i.to(dev), idx_j.to(dev), n_atoms, r.to(dev))
    if transposed:
        plan.build_transposed()
    if kind == "gaussian":
        off, w = O.gaussian_rbf_params(p["n_rbf"], 5.0)
        keep = (off.to(dev), w.to(dev))  # device buffers must outlive the raw C calls
        rb = ops.radial_struct(_lib.SPK_RBF_GAUSSIAN, p["n_rbf"], keep[0], keep[1], 5.0)
    else:
        fr = O.bessel_rbf_params(p["n_rbf"], 5.0).float().to(dev)
        rb = ops.radial_struct(_lib.SPK_RBF_BESSEL, p["n_rbf"], fr, None, 5.0)
        keep = (fr,)
    nf = h.shape[1]
    t = {k: v.to(dev).contiguous() for k, v in p.items() if torch.is_tensor(v)}
    hd, rd, gyd = h.to(dev).contiguous(), r.to(dev).contiguous(), gy.to(dev).contiguous()
    y = torch.empty(n_atoms, nf, device=dev)
    L = _lib.lib()
    _lib.check(L.spk_schnet_cfconv_fwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(hd), _lib.fptr(rd), _lib.fptr(t["w1"]),
                                           _lib.fptr(t["b1"]), _lib.fptr(t["w2"]), _lib.fptr(t["b2"]), nf, _lib.fptr(y), _lib.stream()))
    gh = torch.empty(n_atoms, nf, device=dev)
    gr = torch.zeros(r.shape[0], 3, device=dev)
    _lib.check(L.spk_schnet_cfconv_bwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(hd), _lib.fptr(gyd), _lib.fptr(rd),
                                           _lib.fptr(t["w1"]), _lib.fptr(t["b1"]), _lib.fptr(t["w2"]), _lib.fptr(t["b2"]), nf,
                                           _lib.fptr(gh), _lib.fptr(gr), _lib.stream()))
    torch.cuda.synchronize()
    del keep
    return y.cpu(), gh.cpu(), gr.cpu(), plan


def _filter_params(nf, n_rbf, seed):
    g = torch.Generator().manual_seed(seed)
    return {"n_rbf": n_rbf, "w1": torch.randn(nf, n_rbf, generator=g) * 0.4, "b1": torch.randn(nf, generator=g) * 0.2,
            "w2": torch.randn(nf, nf, generator=g) / nf ** 0.5, "b2": torch.randn(nf, generator=g) * 0.2}


@pytest.mark.parametrize("graph", ["aspirin_sym", "random_sorted", "random_unsorted", "ragged_tail"])
@pytest.mark.parametrize("nf,n_rbf,kind", [(128, 20, "gaussian"), (64, 20, "gaussian"), (128, 16, "bessel"), (96, 11, "gaussian")])
def test_cfconv_forward_backward(dev, variant, graph, nf, n_rbf, kind):
    g = torch.Generator().manual_seed(11)
    if graph == "aspirin_sym":
        b = S.molecule_batch("aspirin", 5, seed=2)
        r = O.pairwise_vectors(b["R"], b["idx_i"], b["idx_j"], b["offsets"])
        idx_i, idx_j, n_atoms = b["idx_i"], b["idx_j"], b["Z"].shape[0]
    else:
        rb = S.random_graph_batch(150 if graph != "ragged_tail" else 7, 9 if graph != "ragged_tail" else 5,
                                  seed=5, sort=(graph != "random_unsorted"))
        r, idx_i, idx_j, n_atoms = rb["r_ij"], rb["idx_i"], rb["idx_j"], rb["Z"].shape[0]
        if graph == "ragged_tail":  # 35 edges: one full tile + 3 edges; atom 6 has no edges
            keep = idx_i < 6
            r, idx_i, idx_j = r[keep], idx_i[keep], idx_j[keep]
    p = _filter_params(nf, n_rbf, 3)
    h = torch.randn(n_atoms, nf, generator=g)
    gy = torch.randn(n_atoms, nf, generator=g)
    yo, gho, gro = _cfconv_oracle(h, r, idx_i, idx_j, p, n_atoms, kind, gy)
    y, gh, gr, plan = _cfconv_hip(dev, h, r, idx_i, idx_j, p, n_atoms, kind, gy)
    assert plan.symmetric == (graph == "aspirin_sym")
    assert rel_err(y, yo) < TOL
    assert rel_err(gh, gho) < TOL
    assert rel_err(gr, gro) < TOL


@pytest.mark.parametrize("nf,n_rbf,kind", [(128, 20, "gaussian"), (64, 16, "bessel")])
def test_cfconv_backward_through_the_by_neighbour_list(dev, nf, n_rbf, kind):
    """Sorted but ASYMMETRIC lists (one-sided / half lists of external back-ends): with the list's by-neighbour copy on the graph
    (spk_transposed_build) the backward's scatter over idx_j runs as the forward kernel over the transposed list + a directed backward
    without atomics on gh -- same results as the oracle and as the atomic path; the transposed arrays against their definition."""
    from schnetpack_amd import _lib
    g = torch.Generator().manual_seed(13)
    rb = S.random_graph_batch(300, 24, seed=7, sort=True)
    r, idx_i, idx_j, n_atoms = rb["r_ij"], rb["idx_i"], rb["idx_j"], rb["Z"].shape[0]
    p = _filter_params(nf, n_rbf, 3)
    h = torch.randn(n_atoms, nf, generator=g); gy = torch.randn(n_atoms, nf, generator=g)
    yo, gho, gro = _cfconv_oracle(h, r, idx_i, idx_j, p, n_atoms, kind, gy)
    _lib.profile_enable(True); _lib.profile_report()
    y, gh, gr, plan = _cfconv_hip(dev, h, r, idx_i, idx_j, p, n_atoms, kind, gy, transposed=True)
    tags = _lib.profile_report(); _lib.profile_enable(False)
    assert not plan.symmetric and plan.sorted
    # forward, transposed forward, directed backward (large lists -- or SPK_CF_ROWTILE=1 -- take the row-tile forward for the first two)
    n_fwd = tags.get("cfconv_fwd_mfma", [0])[0] + tags.get("cfconv_fwd_rowtile", [0])[0]
    assert n_fwd == 2 and tags["cfconv_bwd_mfma_atomic"][0] == 1, tags
    assert rel_err(y, yo) < TOL and rel_err(gh, gho) < TOL and rel_err(gr, gro) < TOL
    y2, gh2, gr2, _ = _cfconv_hip(dev, h, r, idx_i, idx_j, p, n_atoms, kind, gy, transposed=False)
    assert rel_err(gh, gh2) < 2e-6 and rel_err(gr, gr2) < 2e-6
    # definition of the transposed arrays: a stable sort of the pairs by neighbour
    order = torch.sort(idx_j, stable=True).indices
    tb = plan._transposed_bufs
    assert torch.equal(tb["perm"].cpu().long(), order)
    assert torch.equal(tb["idx_i"].cpu(), idx_j[order]) and torch.equal(tb["idx_j"].cpu(), idx_i[order])
    want_rp = torch.searchsorted(idx_j[order].contiguous(), torch.arange(n_atoms + 2))
    assert torch.equal(tb["rowptr"].cpu().long(), want_rp)


def test_cfconv_empty_and_isolated(dev, variant):
    p = _filter_params(128, 20, 1)
    h = torch.randn(4, 128)
    e = torch.zeros(0, dtype=torch.long)
    y, gh, gr, _ = _cfconv_hip(dev, h, torch.zeros(0, 3), e, e, p, 4, "gaussian", torch.randn(4, 128))
    assert float(y.abs().max()) == 0.0 and float(gh.abs().max()) == 0.0 and gr.shape == (0, 3)


def test_cfconv_mfma_equals_simple_at_bench_scale(dev):
    """cfg 2 shape (N=5376, E~77.9k, F=128): both kernel variants and both transposed-sum paths
    agree; linear in h."""
    from schnetpack_amd import _lib
    b = S.molecule_batch("aspirin", 256, seed=0)
    r = O.pairwise_vectors(b["R"], b["idx_i"], b["idx_j"], b["offsets"])
    N = b["Z"].shape[0]
    p = _filter_params(128, 20, 9)
    g = torch.Generator().manual_seed(1)
    h = torch.randn(N, 128, generator=g)
    gy = torch.randn(N, 128, generator=g)
    _lib.set_variant(_lib.VARIANT_MFMA)
    y1, gh1, gr1, plan = _cfconv_hip(dev, h, r, b["idx_i"], b["idx_j"], p, N, "gaussian", gy)
    assert plan.symmetric
    y1b, _, _, _ = _cfconv_hip(dev, 3.0 * h, r, b["idx_i"], b["idx_j"], p, N, "gaussian", gy)
    _lib.set_variant(_lib.VARIANT_MFMA_DIRECTED)
    y3, gh3, gr3, _ = _cfconv_hip(dev, h, r, b["idx_i"], b["idx_j"], p, N, "gaussian", gy)
    _lib.set_variant(_lib.VARIANT_MFMA_MOL)
    y4, gh4, gr4, _ = _cfconv_hip(dev, h, r, b["idx_i"], b["idx_j"], p, N, "gaussian", gy)
    _lib.set_variant(_lib.VARIANT_SIMPLE)
    y2, gh2, gr2, _ = _cfconv_hip(dev, h, r, b["idx_i"], b["idx_j"], p, N, "gaussian", gy)
    _lib.set_variant(_lib.VARIANT_AUTO)
    assert rel_err(y1, y2) < TOL and rel_err(gh1, gh2) < TOL and rel_err(gr1, gr2) < TOL
    assert rel_err(y3, y2) < TOL and rel_err(gh3, gh2) < TOL and rel_err(gr3, gr2) < TOL
    assert rel_err(y4, y2) < TOL and rel_err(gh4, gh2) < TOL and rel_err(gr4, gr2) < TOL
    assert rel_err(y1b, 3.0 * y1) < 2e-6
    # reversed edges carry opposite geometry gradients on a symmetric list: sum_e gr_e r_e parity
    assert torch.isfinite(gr1).all()


# ----------------------------------------------------------------------------- PaiNN message
def _msg_oracle(c, q, mu, r, idx_i, idx_j, wf, bf, n_atoms, F, gq, gmu):
    c = c.double().requires_grad_(True)
    mu = mu.double().requires_grad_(True)
    r = r.double().requires_grad_(True)
    d = torch.sqrt((r * r).sum(1, keepdim=True))
    u = r / d
    off, w = O.gaussian_rbf_params(wf.shape[1], 5.0)
    phi = O.gaussian_rbf(d, off.double(), w.double())
    Wij = O.dense(phi, wf.double(), bf.double()) * O.cosine_cutoff(d, 5.0)[..., None]
    m = Wij * c.unsqueeze(1)[idx_j]
    dq = O.scatter_add(m[..., :F], idx_i, n_atoms)
    dmu = O.scatter_add(m[..., F:2 * F] * u[..., None] + m[..., 2 * F:] * mu[idx_j], idx_i, n_atoms)
    qo = q.double().unsqueeze(1) + dq
    muo = mu + dmu
    gc, gmu_in, gr = torch.autograd.grad((qo.squeeze(1) * gq.double()).sum() + (muo * gmu.double()).sum(), [c, mu, r])
    return qo.squeeze(1).detach(), muo.detach(), gc, gmu_in, gr


@pytest.mark.parametrize("graph", ["aspirin_sym", "random_sorted", "random_unsorted"])
@pytest.mark.parametrize("F,n_rbf", [(128, 20), (64, 20), (128, 25), (128, 32), (64, 16), (96, 20)])
def test_painn_message_forward_backward(dev, variant, graph, F, n_rbf):
    from schnetpack_amd import _lib, ops
    g = torch.Generator().manual_seed(21)
    if graph == "aspirin_sym":
        b = S.molecule_batch("aspirin", 4, seed=6)
        r = O.pairwise_vectors(b["R"], b["idx_i"], b["idx_j"], b["offsets"])
        idx_i, idx_j, N = b["idx_i"], b["idx_j"], b["Z"].shape[0]
    else:
        rb_ = S.random_graph_batch(90, 70 if graph == "random_sorted" else 8, seed=8, sort=(graph != "random_unsorted"))
        r, idx_i, idx_j, N = rb_["r_ij"], rb_["idx_i"], rb_["idx_j"], rb_["Z"].shape[0]
    c = torch.randn(N, 3 * F, generator=g)
    q = torch.randn(N, F, generator=g)
    mu = torch.randn(N, 3, F, generator=g)
    wf = torch.randn(3 * F, n_rbf, generator=g) * 0.3
    bf = torch.randn(3 * F, generator=g) * 0.1
    gq = torch.randn(N, F, generator=g)
    gmu = torch.randn(N, 3, F, generator=g)
    qo, muo, gco, gmuo, gro = _msg_oracle(c, q, mu, r, idx_i, idx_j, wf, bf, N, F, gq, gmu)
    plan = ops.EdgePlan(idx_i.to(dev), idx_j.to(dev), N, r.to(dev))
    off, w = O.gaussian_rbf_params(n_rbf, 5.0)
    offd, wd = off.to(dev), w.to(dev)
    rb = ops.radial_struct(_lib.SPK_RBF_GAUSSIAN, n_rbf, offd, wd, 5.0)
    D = lambda t: t.to(dev).contiguous()
    cd, qd, mud, rd, wfd, bfd, gqd, gmud = map(D, (c, q, mu, r, wf, bf, gq, gmu))
    q_out = torch.empty(N, F, device=dev)
    mu_out = torch.empty(N, 3, F, device=dev)
    L = _lib.lib()
    # forward through both kernel families: row kernel (-1) and the MFMA tile kernel (1: whenever the shape has one;
    # shapes / lists without one fall through to the row or simple kernel)
    try:
        for mode in (-1, -2, 1):       # row kernel, row kernel with the LDS radial table (where the shape has it), MFMA tile kernel
            L.spk_painn_set_tile(-1 if mode == -2 else mode)
            L.spk_painn_set_row_table(1 if mode == -2 else 0)
            q_out.fill_(float("nan")); mu_out.fill_(float("nan"))
            _lib.check(L.spk_painn_message_fwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(qd), _lib.fptr(mud), _lib.fptr(rd),
                                                   _lib.fptr(wfd), _lib.fptr(bfd), F, _lib.fptr(q_out), _lib.fptr(mu_out), _lib.stream()))
            assert rel_err(q_out.cpu(), qo) < TOL, mode
            assert rel_err(mu_out.cpu(), muo) < TOL, mode
    finally:
        L.spk_painn_set_tile(0)
        L.spk_painn_set_row_table(-1)
    gc = torch.empty(N, 3 * F, device=dev)
    gmu_in = torch.empty(N, 3, F, device=dev)
    try:
        for mode in (-1, -2, 1):     # row kernel, row kernel + LDS radial table, MFMA tile kernel (where the shape / list has one)
            L.spk_painn_set_tile(-1 if mode == -2 else mode)
            L.spk_painn_set_row_table(1 if mode == -2 else 0)
            gc.fill_(float("nan")); gmu_in.fill_(float("nan"))
            gr = torch.zeros(r.shape[0], 3, device=dev)
            _lib.check(L.spk_painn_message_bwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(mud), _lib.fptr(gqd), _lib.fptr(gmud),
                                                   _lib.fptr(rd), _lib.fptr(wfd), _lib.fptr(bfd), F, _lib.fptr(gc), _lib.fptr(gmu_in), _lib.fptr(gr),
                                                   _lib.stream()))
            assert rel_err(gc.cpu(), gco) < TOL, mode
            assert rel_err(gmu_in.cpu(), gmuo) < TOL, mode
            assert rel_err(gr.cpu(), gro) < TOL, mode
    finally:
        L.spk_painn_set_tile(0)
        L.spk_painn_set_row_table(-1)


@pytest.mark.parametrize("F,n_rbf,mu_zero", [(128, 20, False), (64, 20, False), (128, 32, False), (128, 20, True)])
def test_painn_message_backward_on_asymmetric_lists_without_atomics(dev, F, n_rbf, mu_zero):
    """Sorted but ASYMMETRIC list (what the LAMMPS interface and vesin hand over, interfaces/lammps/pair_schnetpack.cpp:240-267,
    transform/neighborlist.py:446-456) with its by-neighbour copy attached (spk_transposed_build): the backward runs as two row passes --
    transposed sums over the by-neighbour list (TS), geometry gradient over the list (GEOM) -- equal to the oracle, equal to the
    edge-parallel atomic kernel it replaces, and BIT-reproducible (the atomic kernel is not)."""
    import os
    from schnetpack_amd import _lib, ops
    g = torch.Generator().manual_seed(77)
    rb_ = S.random_graph_batch(700, 24, seed=3, sort=True)
    r, idx_i, idx_j, N = rb_["r_ij"], rb_["idx_i"], rb_["idx_j"], rb_["Z"].shape[0]
    c = torch.randn(N, 3 * F, generator=g)
    q = torch.randn(N, F, generator=g)
    mu = torch.zeros(N, 3, F) if mu_zero else torch.randn(N, 3, F, generator=g)
    wf = torch.randn(3 * F, n_rbf, generator=g) * 0.3
    bf = torch.randn(3 * F, generator=g) * 0.1
    gq = torch.randn(N, F, generator=g)
    gmu = torch.randn(N, 3, F, generator=g)
    _, _, gco, gmuo, gro = _msg_oracle(c, q, mu, r, idx_i, idx_j, wf, bf, N, F, gq, gmu)
    plan = ops.EdgePlan(idx_i.to(dev), idx_j.to(dev), N, r.to(dev))
    assert plan.sorted and not plan.symmetric
    off, w = O.gaussian_rbf_params(n_rbf, 5.0)
    offd, wd = off.to(dev), w.to(dev)
    rb = ops.radial_struct(_lib.SPK_RBF_GAUSSIAN, n_rbf, offd, wd, 5.0)
    D = lambda t: t.to(dev).contiguous()
    cd, mud, rd, wfd, bfd, gqd, gmud = map(D, (c, mu, r, wf, bf, gq, gmu))
    L = _lib.lib()

    def run():
        gc = torch.full((N, 3 * F), float("nan"), device=dev)
        gmu_in = torch.full((N, 3, F), float("nan"), device=dev)
        gr = torch.zeros(r.shape[0], 3, device=dev)
        _lib.profile_enable(True); _lib.profile_report()
        _lib.check(L.spk_painn_message_bwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(mud), _lib.fptr(gqd), _lib.fptr(gmud),
                                               _lib.fptr(rd), _lib.fptr(wfd), _lib.fptr(bfd), F, _lib.fptr(gc), _lib.fptr(gmu_in), _lib.fptr(gr), _lib.stream()))
        tags = set(_lib.profile_report()); _lib.profile_enable(False)
        return gc, gmu_in, gr, tags

    a0 = run()                                  # no by-neighbour copy yet: the edge-parallel kernel with float atomics
    assert "painn_msg_bwd_simple" in a0[3]
    plan.build_transposed()
    a1, a2 = run(), run()
    assert a1[3] == {"painn_msg_bwd_row_tsum", "painn_msg_bwd_row_geom"}, a1[3]
    for got in (a0, a1):
        assert rel_err(got[0].cpu(), gco) < TOL and rel_err(got[1].cpu(), gmuo) < TOL and rel_err(got[2].cpu(), gro) < TOL
    for x, y in zip(a1[:3], a2[:3]):
        assert torch.equal(x, y)                # fixed summation order
    if F == 128 and n_rbf == 20:
        # round 6: on large lists the two passes run in row-tile form (forced here): sums over the by-neighbour list, geometry over the list
        try:
            L.spk_painn_set_rowtile(1)
            b1, b2 = run(), run()
        finally:
            L.spk_painn_set_rowtile(0)
        assert b1[3] == {"painn_msg_bwd_rowtile_t", "painn_msg_bwd_rowtile_geom"}, b1[3]
        assert rel_err(b1[0].cpu(), gco) < TOL and rel_err(b1[1].cpu(), gmuo) < TOL and rel_err(b1[2].cpu(), gro) < TOL
        for x, y in zip(b1[:3], b2[:3]):
            assert torch.equal(x, y)


@pytest.mark.parametrize("n_rbf,mu_zero,skin", [(20, False, False), (20, True, False), (28, False, False), (20, False, True)])
def test_painn_message_row_tile(dev, n_rbf, mu_zero, skin):
    """Row-tile backward (round 6, spk_painn_tile.hip): a wavefront per row, filter and slope from the split-precision GEMM of 32-pair chunks,
    geometry launch + transposed-sums launch.  Symmetric ring lists whose rows need one, two and three chunks (degree 24 / 56 / 70), pairs
    beyond the cutoff in every row (f_c = 0; with `skin` the rows are compacted first): equal to the float64 oracle, equal to the row kernel it
    replaces, and bit-reproducible.  (The geometry-only launch of an eval-mode backward is exercised by the box force calls of test_gpu_scale.py.)"""
    from schnetpack_amd import _lib, ops
    F = 128
    g = torch.Generator().manual_seed(5)
    L = _lib.lib()
    for degree in (24, 56, 70):
        rb_ = S.ring_graph_batch(300, degree, seed=degree, dmin=0.9, dmax=5.6 if skin else 5.05)
        r, idx_i, idx_j, N = rb_["r_ij"], rb_["idx_i"], rb_["idx_j"], rb_["Z"].shape[0]
        c = torch.randn(N, 3 * F, generator=g)
        q = torch.randn(N, F, generator=g)
        mu = torch.zeros(N, 3, F) if mu_zero else torch.randn(N, 3, F, generator=g)
        wf = torch.randn(3 * F, n_rbf, generator=g) * 0.3
        bf = torch.randn(3 * F, generator=g) * 0.1
        gq = torch.randn(N, F, generator=g)
        gmu = torch.randn(N, 3, F, generator=g)
        qo, muo, gco, gmuo, gro = _msg_oracle(c, q, mu, r, idx_i, idx_j, wf, bf, N, F, gq, gmu)
        plan = ops.EdgePlan(idx_i.to(dev), idx_j.to(dev), N, r.to(dev))
        assert plan.sorted and plan.symmetric
        if skin:
            plan.set_filter(True)
        off, w = O.gaussian_rbf_params(n_rbf, 5.0)
        offd, wd = off.to(dev), w.to(dev)
        rb = ops.radial_struct(_lib.SPK_RBF_GAUSSIAN, n_rbf, offd, wd, 5.0)
        D = lambda t: t.to(dev).contiguous()
        cd, mud, rd, wfd, bfd, gqd, gmud = map(D, (c, mu, r, wf, bf, gq, gmu))

        def run(want_sums=True):
            gc = torch.full((N, 3 * F), float("nan"), device=dev)
            gmu_in = torch.full((N, 3, F), float("nan"), device=dev)
            gr = torch.zeros(r.shape[0], 3, device=dev)
            _lib.profile_enable(True); _lib.profile_report()
            _lib.check(L.spk_painn_message_bwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(mud), _lib.fptr(gqd), _lib.fptr(gmud),
                                                   _lib.fptr(rd), _lib.fptr(wfd), _lib.fptr(bfd), F, _lib.fptr(gc) if want_sums else None,
                                                   _lib.fptr(gmu_in) if want_sums else None, _lib.fptr(gr), _lib.stream()))
            tags = set(_lib.profile_report()); _lib.profile_enable(False)
            return gc, gmu_in, gr, tags

        def run_fwd():
            qd = q.to(dev).contiguous()
            q_out = torch.full((N, F), float("nan"), device=dev)
            mu_out = torch.full((N, 3, F), float("nan"), device=dev)
            _lib.profile_enable(True); _lib.profile_report()
            _lib.check(L.spk_painn_message_fwd_f32(plan.graph(), ctypes.byref(rb), _lib.fptr(cd), _lib.fptr(qd), _lib.fptr(mud), _lib.fptr(rd),
                                                   _lib.fptr(wfd), _lib.fptr(bfd), F, _lib.fptr(q_out), _lib.fptr(mu_out), _lib.stream()))
            tags = set(_lib.profile_report()); _lib.profile_enable(False)
            return q_out, mu_out, tags

        try:
            L.spk_painn_set_rowtile(1)
            f1, f2 = run_fwd(), run_fwd()        # the forward in the same form (a wavefront per row, no atomics)
            assert f1[2] == {"painn_msg_fwd_rowtile"}, f1[2]      # (the mu == 0 instance is chosen by the representation driver, not by this entry point)
            assert rel_err(f1[0].cpu(), qo) < TOL and rel_err(f1[1].cpu(), muo) < TOL, degree
            assert torch.equal(f1[0], f2[0]) and torch.equal(f1[1], f2[1])
            a1, a2 = run(), run()
            assert a1[3] == {"painn_msg_bwd_rowtile_g", "painn_msg_bwd_rowtile_t"}, a1[3]
            assert rel_err(a1[0].cpu(), gco) < TOL and rel_err(a1[1].cpu(), gmuo) < TOL and rel_err(a1[2].cpu(), gro) < TOL, degree
            for x, y in zip(a1[:3], a2[:3]):
                assert torch.equal(x, y)                # no atomics: fixed summation order
            L.spk_painn_set_rowtile(-1)
            b1 = run()
            assert not any("rowtile" in t for t in b1[3])
            assert rel_err(a1[2].cpu(), b1[2].cpu()) < TOL
        finally:
            L.spk_painn_set_rowtile(0)


def test_painn_mixing_elementwise(dev):
    from schnetpack_amd import _lib
    g = torch.Generator().manual_seed(31)
    N, F, eps = 57, 128, 1e-8
    q = torch.randn(N, F, generator=g)
    mu = torch.randn(N, 3, F, generator=g)
    mix = torch.randn(N, 3, 2 * F, generator=g)
    a = torch.randn(N, 3 * F, generator=g)
    gq = torch.randn(N, F, generator=g)
    gmu = torch.randn(N, 3, F, generator=g)
    gctx = torch.randn(N, 2 * F, generator=g)
    # oracle
    mixo = mix.double().requires_grad_(True)
    ao = a.double().requires_grad_(True)
    V, W = mixo[..., :F], mixo[..., F:]
    Vn = torch.sqrt((V * V).sum(1) + eps)
    ctx_o = torch.cat([q.double(), Vn], -1)
    qo = q.double() + ao[:, :F] + ao[:, 2 * F:] * (V * W).sum(1)
    muo = mu.double() + ao[:, None, F:2 * F] * W
    ga_o, gmix_o = torch.autograd.grad((qo * gq.double()).sum() + (muo * gmu.double()).sum() + (ctx_o * gctx.double()).sum(), [ao, mixo])
    D = lambda t: t.to(dev).contiguous()
    qd, mud, mixd, ad, gqd, gmud, gctxd = map(D, (q, mu, mix, a, gq, gmu, gctx))
    L = _lib.lib()
    ctx = torch.empty(N, 2 * F, device=dev)
    _lib.check(L.spk_painn_mix_ctx_f32(_lib.fptr(qd), _lib.fptr(mixd), N, F, eps, _lib.fptr(ctx), _lib.stream()))
    assert rel_err(ctx.cpu(), ctx_o.detach()) < TOL
    q_out, mu_out = torch.empty(N, F, device=dev), torch.empty(N, 3, F, device=dev)
    _lib.check(L.spk_painn_mix_update_f32(_lib.fptr(qd), _lib.fptr(mud), _lib.fptr(mixd), _lib.fptr(ad), N, F, _lib.fptr(q_out), _lib.fptr(mu_out), _lib.stream()))
    assert rel_err(q_out.cpu(), qo.detach()) < TOL and rel_err(mu_out.cpu(), muo.detach()) < TOL
    ga, gmix = torch.empty(N, 3 * F, device=dev), torch.empty(N, 3, 2 * F, device=dev)
    _lib.check(L.spk_painn_mix_update_bwd_f32(None, _lib.fptr(mixd), _lib.fptr(ad), _lib.fptr(gqd), _lib.fptr(gmud), N, F, _lib.fptr(ga), _lib.fptr(gmix), _lib.stream()))
    gq1 = torch.empty(N, F, device=dev)
    _lib.check(L.spk_painn_mix_ctx_bwd_f32(_lib.fptr(mixd), _lib.fptr(gctxd), _lib.fptr(gqd), N, F, eps, _lib.fptr(gmix), _lib.fptr(gq1), _lib.stream()))
    assert rel_err(ga.cpu(), ga_o) < TOL
    assert rel_err(gmix.cpu(), gmix_o) < TOL
    assert rel_err(gq1.cpu(), gq.double() + gctx.double()[:, :F]) < TOL


# ----------------------------------------------------------------------------- pairwise vectors
def test_pairwise_vectors_forward_backward_and_second_order(dev):
    """r_ij = R[j] - R[i] + offsets (atomistic/distances.py:14-26): bit-exact forward (and the
    reversed edge is the exact negation), atomic scatter backward, closed under differentiation."""
    from schnetpack_amd import ops
    b = S.molecule_batch("aspirin", 3, seed=5)
    g = torch.Generator().manual_seed(2)
    off = torch.randn(b["idx_i"].shape[0], 3, generator=g)
    R = b["R"].clone()
    ref = O.pairwise_vectors(R, b["idx_i"], b["idx_j"], off)
    Rd = R.to(dev).requires_grad_(True)
    ii, jj = b["idx_i"].to(dev), b["idx_j"].to(dev)
    r = ops.pairwise_vectors(Rd, ii, jj, off.to(dev))
    assert torch.equal(r.detach().cpu(), ref)
    r0 = ops.pairwise_vectors(Rd, ii, jj, None)
    assert torch.equal(r0.detach().cpu(), R[b["idx_j"]] - R[b["idx_i"]])
    w = torch.randn(r.shape, generator=g)

    def run(R, ii, jj, off, w, fn):
        R = R.clone().requires_grad_(True)
        rr = fn(R, ii, jj, off)
        (gR,) = torch.autograd.grad(((rr ** 2) * w).sum(), [R], create_graph=True)
        (g2,) = torch.autograd.grad((gR ** 2).sum(), [R])
        return gR.detach().cpu(), g2.cpu()

    gh, g2h = run(R.to(dev), ii, jj, off.to(dev), w.to(dev), ops.pairwise_vectors)
    go, g2o = run(R.double(), b["idx_i"], b["idx_j"], off.double(), w.double(), O.pairwise_vectors)
    assert rel_err(gh, go) < TOL and rel_err(g2h, g2o) < TOL
    # no offsets: the list is symmetric => segmented row-sum backward (no atomics)
    gh, g2h = run(R.to(dev), ii, jj, None, w.to(dev), ops.pairwise_vectors)
    go, g2o = run(R.double(), b["idx_i"], b["idx_j"], torch.zeros_like(off).double(), w.double(), O.pairwise_vectors)
    assert rel_err(gh, go) < TOL and rel_err(g2h, g2o) < TOL


@pytest.mark.parametrize("system", ["aspirin", "water", "ragged"])
def test_pairwise_backward_row_sum_equals_atomic_scatter(dev, system):
    """spk_pairwise_bwd_graph_f32 (segmented row sum over rev[e] on symmetric sorted lists) against the
    fp64 oracle and against the atomic kernel; rows longer than 16 edges (water: ~52) and empty rows."""
    from schnetpack_amd import _lib, ops
    if system == "aspirin":
        b = S.molecule_batch("aspirin", 5, seed=2)
    elif system == "water":
        b = S.water_box(n_side=4, seed=1)
    else:   # isolated atoms (empty rows) between molecules
        b = S.molecule_batch("ethanol", 3, seed=2)
        b["R"] = torch.cat([b["R"], torch.full((4, 3), 50.0) + 20 * torch.arange(4.0)[:, None]])
        b["Z"] = torch.cat([b["Z"], torch.ones(4, dtype=torch.long)])
    N = b["Z"].shape[0]
    r = O.pairwise_vectors(b["R"], b["idx_i"], b["idx_j"], b["offsets"])
    ii, jj = b["idx_i"].to(dev), b["idx_j"].to(dev)
    plan = ops.EdgePlan(ii, jj, N, r.to(dev))
    assert plan.sorted and plan.symmetric
    gr = torch.randn(r.shape, generator=torch.Generator().manual_seed(4))
    expect = torch.zeros(N, 3, dtype=torch.float64)
    expect.index_add_(0, b["idx_j"], gr.double())
    expect.index_add_(0, b["idx_i"], -gr.double())
    grd = gr.to(dev)
    g_row = torch.full((N, 3), float("nan"), device=dev)
    g_at = torch.full((N, 3), float("nan"), device=dev)
    _lib.check(_lib.lib().spk_pairwise_bwd_graph_f32(_lib.fptr(grd), plan.graph(), _lib.fptr(g_row), _lib.stream()))
    _lib.check(_lib.lib().spk_pairwise_bwd_f32(_lib.fptr(grd), _lib.iptr(ii), _lib.iptr(jj), ii.shape[0], N, _lib.fptr(g_at), _lib.stream()))
    torch.cuda.synchronize()
    assert rel_err(g_row.cpu(), expect) < TOL and rel_err(g_at.cpu(), expect) < TOL
    g_row2 = torch.empty_like(g_row)
    _lib.check(_lib.lib().spk_pairwise_bwd_graph_f32(_lib.fptr(grd), plan.graph(), _lib.fptr(g_row2), _lib.stream()))
    assert torch.equal(g_row, g_row2)          # deterministic


# ----------------------------------------------------------------------------- fused dense chain
def _chain_struct(_lib, m, inp, layers, in_pre=None, in_act=0, zero=None, tmp=None):
    c = _lib.ChainT()
    c.n_layers = len(layers)
    c.in_act = in_act
    c.m = m
    c.inp = _lib.fptr(inp)
    c.in_pre = _lib.fptr(in_pre)
    c.zero_ptr = _lib.fptr(zero)
    c.zero_count = zero.numel() if zero is not None else 0
    if tmp is not None:
        c.tmp[0] = tmp[0].data_ptr()
        c.tmp[1] = tmp[1].data_ptr()
    for l, L in enumerate(layers):
        for k in ("w", "b", "res", "out", "pre_out", "post_pre"):
            setattr(c.layers[l], k, _lib.fptr(L.get(k)))
        for k in ("k", "n_out", "act", "trans", "post_act"):
            setattr(c.layers[l], k, int(L.get(k, 0)))
    return c


@pytest.fixture(params=[0, 16, 32])
def chain_rows(request):
    """Row-tile height of the fused chain kernel: by size, 16 (v_mfma_f32_16x16x4_f32), 32 (32x32x2)."""
    from schnetpack_amd import _lib
    _lib.lib().spk_chain_set_rows(request.param)
    yield request.param
    _lib.lib().spk_chain_set_rows(0)


def _pack(w, transposed):
    """spk_pack_weight_f32 image of a Linear weight [n_out, k_in] (device tensor)."""
    from schnetpack_amd import _lib
    P = torch.empty(w.numel(), device=w.device)
    _lib.check(_lib.lib().spk_pack_weight_f32(_lib.fptr(w), w.shape[0], w.shape[1], transposed, _lib.fptr(P), _lib.stream()))
    return P


def test_pack_weight_layout(dev):
    """P[((t KB + ug) 64 + lane) 4 + v] = A[32 t + (lane & 31)][8 ug + 4 (lane >> 5) + v], A = W or W^T."""
    g = torch.Generator().manual_seed(3)
    W = torch.randn(96, 40, generator=g)             # [n_out, k_in]
    for transposed, A in ((0, W), (1, W.t())):        # A [width, contraction]
        if transposed:
            Wd = torch.randn(40, 96, generator=g)    # contraction 40 = n_out, width 96 = k_in
            A = Wd.t()
        else:
            Wd = W
        P = _pack(Wd.to(dev).contiguous(), transposed).cpu()
        NW, KC = A.shape
        KB = KC // 8
        s = torch.arange(NW * KC)
        v, lane, blk = s & 3, (s >> 2) & 63, s >> 8
        ug, t = blk % KB, blk // KB
        assert torch.equal(P, A[32 * t + (lane & 31), 8 * ug + 4 * (lane >> 5) + v])


@pytest.mark.parametrize("layout", ["rowmajor", "kmajor", "packed"])
@pytest.mark.parametrize("m", [5376, 37])
def test_dense_chain_forward_style(dev, variant, m, chain_rows, layout):
    """f2out.0 (ssp, pre saved) -> f2out.1 (+ residual, stored) -> in2f (stored), buffer cleared.
    packed: the fused kernels (16- and 32-row tiles); row-major [n_out, k] or k-major [k, n_out] (transposed
    copy) weights: layer by layer -- the k-major form with an activation is what model widths without a packed
    image use."""
    from schnetpack_amd import _lib
    packed = layout == "packed"
    if packed and variant == "simple":
        pytest.skip("packed weights are an MFMA-kernel format")
    g = torch.Generator().manual_seed(41)
    F = 128
    y = torch.randn(m, F, generator=g)
    x = torch.randn(m, F, generator=g)
    w3, b3 = torch.randn(F, F, generator=g) / 11, torch.randn(F, generator=g) * 0.1
    w4, b4 = torch.randn(F, F, generator=g) / 11, torch.randn(F, generator=g) * 0.1
    win = torch.randn(F, F, generator=g) / 11
    pre_o = O.dense(y.double(), w3.double(), b3.double())
    x_o = x.double() + O.dense(O.shifted_softplus(pre_o), w4.double(), b4.double())
    h_o = O.dense(x_o, win.double())
    D = lambda t: t.to(dev).contiguous()
    yd, xd, w3d, b3d, w4d, b4d, wind = map(D, (y, x, w3, b3, w4, b4, win))
    pre = torch.empty(m, F, device=dev)
    h = torch.empty(m, F, device=dev)
    junk = torch.ones(1000, device=dev)
    tmp = (torch.empty(m, F, device=dev), torch.empty(m, F, device=dev))
    tr = 0
    if packed:
        w3d, w4d, wind = _pack(w3d, 0), _pack(w4d, 0), _pack(wind, 0)
        tr = 2
    elif layout == "kmajor":
        w3d, w4d, wind = w3d.t().contiguous(), w4d.t().contiguous(), wind.t().contiguous()
        tr = 1
    c = _chain_struct(_lib, m, yd, [
        dict(w=w3d, b=b3d, pre_out=pre, k=F, n_out=F, act=_lib.SPK_ACT_SSP, trans=tr),
        dict(w=w4d, b=b4d, res=xd, out=xd, k=F, n_out=F, trans=tr),
        dict(w=wind, out=h, k=F, n_out=F, trans=tr)], zero=junk, tmp=tmp)
    _lib.check(_lib.lib().spk_dense_chain_f32(ctypes.byref(c), _lib.stream()))
    torch.cuda.synchronize()
    assert rel_err(pre.cpu(), pre_o) < TOL
    assert rel_err(xd.cpu(), x_o) < TOL      # in-place residual update
    assert rel_err(h.cpu(), h_o) < TOL
    assert float(junk.abs().max()) == 0.0


@pytest.mark.parametrize("NF,packed", [(64, False), (128, False), (128, True), (384, True)])
def test_dense_chain_backward_style(dev, variant, NF, packed, chain_rows):
    """(gh W_in + gx) -> (. W4) * ssp'(pre3) -> (. W3): the transposed chain of the backward."""
    from schnetpack_amd import _lib
    if packed and variant == "simple":
        pytest.skip("packed weights are an MFMA-kernel format")
    g = torch.Generator().manual_seed(43)
    m, F = 777, 128
    gh = torch.randn(m, NF, generator=g)
    gx = torch.randn(m, F, generator=g)
    pre3 = torch.randn(m, F, generator=g)
    win = torch.randn(NF, F, generator=g) / 8     # in2f  [nf, F]
    w4 = torch.randn(F, F, generator=g) / 11      # f2out.1 [F, F]
    w3 = torch.randn(F, NF, generator=g) / 8      # f2out.0 [F, nf]
    gx_o = gx.double() + gh.double() @ win.double()
    gt_o = (gx_o @ w4.double()) * torch.sigmoid(pre3.double())
    gy_o = gt_o @ w3.double()
    D = lambda t: t.to(dev).contiguous()
    ghd, gxd, pred, wind, w4d, w3d = map(D, (gh, gx, pre3, win, w4, w3))
    gx_new = torch.empty(m, F, device=dev)
    gy = torch.empty(m, NF, device=dev)
    tmp = (torch.empty(m, F, device=dev), torch.empty(m, F, device=dev))
    tr = 1
    if packed:
        wind, w4d, w3d = _pack(wind, 1), _pack(w4d, 1), _pack(w3d, 1)
        tr = 2
    c = _chain_struct(_lib, m, ghd, [
        dict(w=wind, res=gxd, out=gx_new, k=NF, n_out=F, trans=tr),
        dict(w=w4d, post_pre=pred, post_act=_lib.SPK_ACT_SSP, k=F, n_out=F, trans=tr),
        dict(w=w3d, out=gy, k=F, n_out=NF, trans=tr)], tmp=tmp)
    _lib.check(_lib.lib().spk_dense_chain_f32(ctypes.byref(c), _lib.stream()))
    torch.cuda.synchronize()
    assert rel_err(gx_new.cpu(), gx_o) < TOL
    assert rel_err(gy.cpu(), gy_o) < TOL


def test_dense_chain_odd_shapes_use_layerwise_path(dev):
    """n_rbf -> F -> 1 style shapes (k % 8 != 0, n_out % 32 != 0) run layer by layer."""
    from schnetpack_amd import _lib
    g = torch.Generator().manual_seed(44)
    m = 91
    x = torch.randn(m, 20, generator=g)
    w1, b1 = torch.randn(64, 20, generator=g) / 4, torch.randn(64, generator=g) * 0.1
    w2, b2 = torch.randn(1, 64, generator=g) / 8, torch.randn(1, generator=g)
    o = O.dense(O.silu(O.dense(x.double(), w1.double(), b1.double())), w2.double(), b2.double())
    D = lambda t: t.to(dev).contiguous()
    xd, w1d, b1d, w2d, b2d = map(D, (x, w1, b1, w2, b2))
    out = torch.empty(m, 1, device=dev)
    tmp = (torch.empty(m, 64, device=dev), torch.empty(m, 64, device=dev))
    c = _chain_struct(_lib, m, xd, [dict(w=w1d, b=b1d, k=20, n_out=64, act=_lib.SPK_ACT_SILU),
                                    dict(w=w2d, b=b2d, out=out, k=64, n_out=1)], tmp=tmp)
    _lib.check(_lib.lib().spk_dense_chain_f32(ctypes.byref(c), _lib.stream()))
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), o) < TOL


@pytest.mark.parametrize("inner,n_rows", [(1, 1), (1, 3), (3, 2), (8, 4)])
def test_scatter_add_few_long_rows(dev, inner, n_rows):
    """Per-atom values of a few large systems summed per system (Atomwise on one 32 k-atom box): the
    wave-per-row kernel; rows of very different lengths, an empty row in the middle."""
    from schnetpack_amd import ops
    g = torch.Generator().manual_seed(inner + n_rows)
    lens = [20000, 0, 7777, 301][:n_rows] if n_rows > 1 else [31944]
    idx = torch.repeat_interleave(torch.arange(len(lens)), torch.tensor(lens))
    x = torch.randn(idx.shape[0], inner, generator=g)
    want = torch.zeros(len(lens), inner, dtype=torch.float64).index_add_(0, idx, x.double())
    got = ops.scatter_add(x.to(dev), idx.to(dev), len(lens))
    assert rel_err(got.cpu(), want) < 1e-5
