"""The kernels of the training regime (csrc/spk_train.hip and the masked Dense tiles) against the float64 formulas of
tests/cpu_reference_kernels.py (evaluated on the host), through ``torch.ops.spk_hip`` -- values, then first and second
derivatives through the operators' own autograd, on the shapes a SchNet / PaiNN training step produces (168 atoms, 2.4 k pairs,
widths 20 / 64 / 128 / 1) and on awkward ones."""
import pytest
import torch

import cpu_reference_kernels as crk
from conftest import rel_err

pytestmark = pytest.mark.gpu
ops = torch.ops.spk_hip
TOL = 2e-6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def rnd(*s, seed=0):
    g = torch.Generator().manual_seed(seed + sum(s))
    return torch.randn(*s, generator=g)


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_act_mul(dev, act, order):
    z, a, c = 3 * rnd(777, 33), rnd(777, 33, seed=1), rnd(777, 33, seed=2)
    ref = crk.act_mul(a.double(), z.double(), act, order, c.double())
    got = ops.act_mul(a.to(dev), z.to(dev), act, order, c.to(dev))
    assert rel_err(got.cpu(), ref) < TOL
    assert rel_err(ops.act_mul(None, z.to(dev), act, order).cpu(), crk.act_mul(None, z.double(), act, order)) < TOL or (act == 0 and order >= 2)


@pytest.mark.parametrize("n,k,o", [(168, 128, 128), (2432, 20, 128), (2432, 128, 20), (168, 64, 1), (168, 1, 64), (5, 12, 36), (1000, 132, 100), (70, 128, 1152),
                                   (2560, 20, 1152), (100, 640, 64), (33, 1000, 8)])
def test_linear_and_matmul_nn_any_width(dev, n, k, o):
    """x W^T + b and u W on the MFMA tiles for widths that are multiples of 4 (masked partial tiles) and on the simple kernel
    otherwise."""
    x, w, b, u = rnd(n, k), rnd(o, k, seed=1) / k ** 0.5, rnd(o, seed=2), rnd(n, o, seed=3)
    assert rel_err(ops.linear(x.to(dev), w.to(dev), b.to(dev)).cpu(), x.double() @ w.double().t() + b.double()) < TOL
    assert rel_err(ops.linear(x.to(dev), w.to(dev), None).cpu(), x.double() @ w.double().t()) < TOL
    assert rel_err(ops.matmul_nn(u.to(dev), w.to(dev)).cpu(), u.double() @ w.double()) < TOL
    for act in (1, 2):
        y = ops.dense(x.to(dev), w.to(dev), b.to(dev), act)
        assert rel_err(y.cpu(), crk.act_order(x.double() @ w.double().t() + b.double(), act, 0)) < TOL


@pytest.mark.parametrize("n,o,k", [(168, 128, 128), (2432, 128, 20), (2432, 20, 128), (168, 1, 64), (168, 64, 1), (1, 5, 7), (7, 33, 65), (20000, 128, 128), (3, 1152, 20)])
def test_matmul_tn(dev, n, o, k):
    u, x = rnd(n, o), rnd(n, k, seed=1)
    G, cs = ops.matmul_tn(u.to(dev), x.to(dev))
    assert rel_err(G.cpu(), u.double().t() @ x.double()) < TOL
    assert rel_err(cs.cpu(), u.double().sum(0)) < TOL
    G2, cs2 = ops.matmul_tn(u.to(dev), x.to(dev))          # tickets were left clean; slices are added in a fixed order
    assert torch.equal(G, G2) and torch.equal(cs, cs2)


@pytest.mark.parametrize("m,k,o,trans", [(2560, 128, 128, True), (2560, 128, 128, False), (168, 64, 128, True), (2432, 20, 128, True), (504, 128, 256, False),
                                          (70, 6, 10, True)])
def test_gemm_pair_is_both_products_of_a_dense_backward(dev, m, k, o, trans):
    """out = a w (trans) | a w^T and (u^T x, column sums of u) from ONE launch equal the two separate operators (the weight-gradient half bit for bit)
    (widths that are not multiples of 4 take the two launches inside the operator)."""
    w, u, x = rnd(o, k, seed=1) / k ** 0.5, rnd(m, o, seed=2), rnd(m, k, seed=3)
    a = u if trans else x
    out, G, cs = ops.gemm_pair(a.to(dev), w.to(dev), trans, u.to(dev), x.to(dev))
    ref_out = ops.matmul_nn(a.to(dev), w.to(dev)) if trans else ops.linear(a.to(dev), w.to(dev), None)
    G2, cs2 = ops.matmul_tn(u.to(dev), x.to(dev))
    # (the stand-alone Dense launch of a small problem splits the contraction over the waves of a workgroup -- k_dense_mfma_sk -- so its sums
    #  associate differently from the one-wave-per-tile walk inside the pair kernel: equal to fp32 rounding, not bit for bit)
    assert rel_err(out.cpu(), ref_out.cpu()) < 1e-6 and torch.equal(G, G2) and torch.equal(cs, cs2)
    assert rel_err(out.cpu(), (a.double() @ w.double()) if trans else (a.double() @ w.double().t())) < TOL
    assert rel_err(G.cpu(), u.double().t() @ x.double()) < TOL


def _lists(n_atoms, E, seed):
    g = torch.Generator().manual_seed(seed)
    ii = torch.randint(0, n_atoms, (E,), generator=g).sort().values
    jj = torch.randint(0, n_atoms, (E,), generator=g)
    return ii, jj


@pytest.mark.parametrize("N,E,F", [(168, 2432, 128), (50, 300, 20), (9, 0, 8), (33, 100, 7)])
def test_cfconv_and_edge_mul(dev, N, E, F):
    ii, jj = _lists(N, E, 3)
    x, W, a = rnd(N, F), rnd(E, F, seed=1), rnd(N, F, seed=2)
    ref = crk.cfconv(x.double(), W.double(), ii, jj, N)
    torch.ops.spk_hip.clear_caches()
    got = ops.cfconv(x.to(dev), W.to(dev), ii.to(dev), jj.to(dev), N)                 # ascending output index: segmented sum
    assert rel_err(got.cpu(), ref) < TOL or E == 0
    assert E > 0 or float(got.abs().max()) == 0.0
    got_t = ops.cfconv(x.to(dev), W.to(dev), jj.to(dev), ii.to(dev), N)               # unsorted output index: atomics
    assert rel_err(got_t.cpu(), crk.cfconv(x.double(), W.double(), jj, ii, N)) < TOL or E == 0
    if E:
        assert rel_err(ops.edge_mul(a.to(dev), x.to(dev), ii.to(dev), jj.to(dev)).cpu(), a.double()[ii] * x.double()[jj]) < TOL


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_radial_functions_and_their_derivatives(dev, kind, order):
    R = 20
    d = torch.rand(3000, generator=torch.Generator().manual_seed(5)) * 5.5 + 0.4        # some pairs beyond the cutoff
    a, G = rnd(3000), rnd(3000, R, seed=1)
    p0 = torch.linspace(0.0, 5.0, R) if kind == 0 else torch.arange(1, R + 1) * torch.pi / 5.0
    p1 = torch.full((R,), 5.0 / (R - 1)) if kind == 0 else None
    ref = crk.radial_d(d.double(), a.double(), kind, p0.double(), None if p1 is None else p1.double(), 5.0, order)
    got = ops.radial_d(d.to(dev), a.to(dev), kind, p0.to(dev), None if p1 is None else p1.to(dev), 5.0, order)
    assert got.shape == ref.shape and rel_err(got.cpu(), ref) < 5e-6
    if kind != 2:
        ref = crk.radial_c(G.double(), d.double(), a.double(), kind, p0.double(), None if p1 is None else p1.double(), 5.0, order)
        got = ops.radial_c(G.to(dev), d.to(dev), a.to(dev), kind, p0.to(dev), None if p1 is None else p1.to(dev), 5.0, order)
        assert rel_err(got.cpu(), ref) < 5e-6


def test_identity_indices(dev):
    N, E, F = 40, 300, 24
    ii, jj = _lists(N, E, 9)
    W, x, xe = rnd(E, F), rnd(N, F, seed=1), rnd(E, F, seed=2)
    assert rel_err(ops.edge_mul(W.to(dev), x.to(dev), None, jj.to(dev)).cpu(), W.double() * x.double()[jj]) < TOL
    assert rel_err(ops.cfconv(xe.to(dev), W.to(dev), ii.to(dev), None, N).cpu(), crk.cfconv(xe.double(), W.double(), ii, None, N)) < TOL
    assert rel_err(ops.cfconv(x.to(dev), W.to(dev), None, jj.to(dev), E).cpu(), x.double()[jj] * W.double()) < TOL


@pytest.mark.parametrize("op", [0, 1, 2, 3, 4])
def test_three_vector_products_read_split_halves_in_place(dev, op):
    """vec3 on dense operands and on halves of split tensors (row-strided views: no copy), against the formulas."""
    M, F = 333, 48
    V2, s3, u = rnd(M, 3, 2 * F), rnd(M, 1, 3 * F, seed=1), rnd(M, 3, seed=2)
    Vd, sd = V2.to(dev), s3.to(dev)
    for half in (0, 1):
        V, Vr = Vd[..., half * F:(half + 1) * F], V2.double()[..., half * F:(half + 1) * F]
        W, Wr = Vd[..., (1 - half) * F:(2 - half) * F], V2.double()[..., (1 - half) * F:(2 - half) * F]
        sv, sr = sd[..., half * F:(half + 1) * F], s3.double()[..., half * F:(half + 1) * F]
        if op == 0:
            got, ref = ops.vec3(0, V, sv), crk.vec3(0, Vr, sr)
        elif op == 1:
            got, ref = ops.vec3(1, V, W), crk.vec3(1, Vr, Wr)
        elif op == 2:
            got, ref = ops.vec3(2, sv, u.to(dev)), crk.vec3(2, sr, u.double())
        elif op == 3:
            got, ref = ops.vec3(3, V, u.to(dev)), crk.vec3(3, Vr, u.double())
        else:
            got, ref = ops.vec3(4, V, sv), crk.vec3(4, Vr, sr)
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert rel_err(got.cpu(), ref) < TOL
    # transposed (non row-uniform) operand: falls back to a dense copy, same values
    Vt = rnd(3, M, F).to(dev).permute(1, 0, 2)
    assert rel_err(ops.vec3(1, Vt, Vt).cpu(), crk.vec3(1, Vt.cpu().double(), Vt.cpu().double())) < TOL


def test_rowscale_rowdot_edge_norm(dev):
    W, s, b = rnd(2432, 128), rnd(2432, seed=1), rnd(2432, 128, seed=2)
    assert rel_err(ops.rowscale(W.to(dev), s.to(dev)).cpu(), W.double() * s.double()[:, None]) < TOL
    assert rel_err(ops.rowdot(W.to(dev), b.to(dev)).cpu(), (W.double() * b.double()).sum(1)) < TOL
    W3 = rnd(100, 1, 36)
    assert rel_err(ops.rowscale(W3.to(dev), s[:100, None].to(dev)).cpu(), W3.double() * s[:100, None, None].double()) < TOL
    r = rnd(999, 3)
    assert rel_err(ops.edge_norm(r.to(dev)).cpu(), r.double().norm(dim=1)) < TOL


def test_second_derivatives_on_the_device(dev):
    """One interaction's worth of the family chained on the device: d -> (phi, f_c) -> filter Dense -> rowscale -> cfconv ->
    Dense; the gradient w.r.t. d is taken with create_graph and a function of it is differentiated w.r.t. the weights --
    against the same chain on float64 torch formulas."""
    N, E, F, R = 60, 700, 64, 20
    ii, jj = _lists(N, E, 11)
    d0 = torch.rand(E, generator=torch.Generator().manual_seed(1)) * 4.5 + 0.5
    x0, w1, b1, w2 = rnd(N, F), rnd(F, R) / R ** 0.5, rnd(F, seed=1) * 0.1, rnd(F, F, seed=2) / F ** 0.5
    p0, p1 = torch.linspace(0.0, 5.0, R), torch.full((R,), 5.0 / (R - 1))

    def chain(o, d, x, w1, b1, w2, p0, p1, ii, jj):
        phi = o["radial_d"](d, None, 0, p0, p1, 5.0, 0)
        fc = o["radial_d"](d, None, 2, p0, None, 5.0, 0)
        W = o["rowscale"](o["dense"](phi, w1, b1, 1), fc)
        y = o["cfconv"](x, W, ii, jj, x.shape[0])
        y = o["dense"](y, w2, None, 1)
        (gd,) = torch.autograd.grad(y.sum(), [d], create_graph=True)
        return (gd ** 2).sum() + (y ** 2).sum()

    hip = {k: getattr(ops, k) for k in ("radial_d", "rowscale", "dense", "cfconv")}
    ref = dict(crk.KERNELS, dense=lambda x, w, b, act: crk.dense_forward(x, w, b, act)[0])
    leaves_h = [t.to(dev).requires_grad_(True) for t in (d0, x0, w1, b1, w2)]
    leaves_r = [t.double().requires_grad_(True) for t in (d0, x0, w1, b1, w2)]
    lh = chain(hip, *leaves_h, p0.to(dev), p1.to(dev), ii.to(dev), jj.to(dev))
    lr = chain(ref, *leaves_r, p0.double(), p1.double(), ii, jj)
    assert abs(float(lh) - float(lr)) < 1e-5 * abs(float(lr))
    gh = torch.autograd.grad(lh, leaves_h)
    gr = torch.autograd.grad(lr, leaves_r)
    for a, b, name in zip(gh, gr, ("d", "x", "w1", "b1", "w2")):
        assert rel_err(a.cpu(), b) < 2e-5, name


def test_training_operators_refuse_malformed_input(dev):
    """Error behaviour of the operator layer: shape / order mismatches raise RuntimeError (TORCH_CHECK or the C ABI's SPK_ERR_ARG),
    nothing is computed on a silent fallback."""
    x, W = rnd(10, 8).to(dev), rnd(30, 8).to(dev)
    ii = torch.randint(0, 10, (30,), device=dev).sort().values
    with pytest.raises(RuntimeError, match="index tensors must have"):
        ops.cfconv(x, W, ii[:-1], ii, 10)
    with pytest.raises(RuntimeError, match="must be"):
        ops.cfconv(x, rnd(30, 6).to(dev), ii, ii, 10)
    with pytest.raises(RuntimeError, match="at least one index"):
        ops.edge_mul(W, W, None, None)
    with pytest.raises(RuntimeError, match="order"):
        ops.act_mul(None, x, 2, 4)                              # silu: derivatives up to the third
    with pytest.raises(RuntimeError, match="order"):
        ops.radial_d(x[:, 0].contiguous(), None, 0, torch.linspace(0, 5, 4, device=dev), torch.ones(4, device=dev), 5.0, 4)
    with pytest.raises(RuntimeError, match="no basis dimension"):
        ops.radial_c(x, x[:, 0].contiguous(), None, 2, torch.ones(1, device=dev), None, 5.0, 0)
    with pytest.raises(RuntimeError, match=r"expected a \[\.\.\., 3, F\] operand"):
        ops.vec3(1, x, x)
    with pytest.raises(RuntimeError, match="do not match"):
        ops.gemm_pair(x, rnd(5, 7).to(dev), True, x, x)
    with pytest.raises(RuntimeError, match="float32"):
        ops.rowdot(x.double(), x.double())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.vec3(0, torch.zeros(2, 3, 4), torch.zeros(2, 4))


def test_force_matching_loss_value_and_gradients(dev):
    """torch.ops.spk_hip.fm_loss = w_E MSE(E) + w_F MSE(F) (one launch) and its first-order gradients (one launch) against the
    framework arithmetic it replaces in GraphedTrainStep."""
    torch.manual_seed(3)
    E = torch.randn(8, device=dev, requires_grad=True)
    F = torch.randn(168, 3, device=dev, requires_grad=True)
    Et, Ft = torch.randn(8, device=dev), torch.randn(168, 3, device=dev)
    wE, wF = 0.01, 0.99
    ref = wE * ((E - Et) ** 2).mean() + wF * ((F - Ft) ** 2).mean()
    gE_ref, gF_ref = torch.autograd.grad(ref * 1.7, (E, F))
    got = torch.ops.spk_hip.fm_loss(E, Et, F, Ft, wE, wF)
    gE, gF = torch.autograd.grad(got * 1.7, (E, F))
    assert got.shape == () and abs(float(got) - float(ref)) < 1e-6 * abs(float(ref))
    assert float((gE - gE_ref).abs().max()) < 1e-6 * float(gE_ref.abs().max())
    assert float((gF - gF_ref).abs().max()) < 1e-6 * float(gF_ref.abs().max())
