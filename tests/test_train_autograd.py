"""The autograd layer of the training-regime operators (csrc/spk_torch_train.h, DenseFn in spk_torch.cpp) checked on the
build box: inside ``reference_kernels()`` (tests/cpu_reference_kernels.py -- test-only float64 formulas on the CPU key, which
the product refuses) ``gradcheck`` / ``gradgradcheck`` run through the real C++ autograd Functions, and a whole force-matching
step (energy + forces with create_graph, loss, backward) is compared with the float64 oracle.  What this pins: which operator
each backward calls, with which roles, to second order -- host logic that is identical on the device."""
import collections

import pytest
import torch
from torch.autograd import gradcheck, gradgradcheck

import cpu_reference_kernels as crk
from oracle import spk_oracle as O
from schnetpack_amd import model as M, synthetic as S

D = torch.float64
ops = torch.ops.spk_hip


def T(*s):
    return torch.randn(*s, dtype=D, requires_grad=True)


def _cases():
    torch.manual_seed(0)
    E, N, F, R = 14, 5, 8, 6
    ii = torch.randint(0, N, (E,)).sort().values
    jj = torch.randint(0, N, (E,))
    p0 = torch.linspace(0.5, 4.0, R, dtype=D)
    p1 = torch.full((R,), 0.7, dtype=D)
    fr = torch.arange(1, R + 1, dtype=D) * 3.14159 / 5.0
    d = (torch.rand(E, dtype=D) * 4 + 0.5).requires_grad_()
    return {
        "act_mul ssp": (lambda a, z, c: ops.act_mul(a, z, 1, 0, c), (T(4, 3), T(4, 3), T(4, 3))),
        "act_mul ssp' without a": (lambda z: ops.act_mul(None, z, 1, 1), (T(4, 3),)),
        "act_mul silu'": (lambda a, z: ops.act_mul(a, z, 2, 1), (T(4, 3), T(4, 3))),
        "linear": (lambda x, w, b: ops.linear(x, w, b), (T(7, 4), T(3, 4), T(3))),
        "linear 3d, no bias": (lambda x, w: ops.linear(x, w, None), (T(2, 7, 4), T(3, 4))),
        "matmul_nn": (lambda u, w: ops.matmul_nn(u, w), (T(7, 3), T(3, 4))),
        "matmul_tn": (lambda u, x: ops.matmul_tn(u, x), (T(7, 3), T(7, 4))),
        "cfconv": (lambda x, W: ops.cfconv(x, W, ii, jj, N), (T(N, F), T(E, F))),
        "cfconv, unsorted output index": (lambda x, W: ops.cfconv(x, W, jj, ii, N), (T(N, F), T(E, F))),
        "edge_mul": (lambda a, b: ops.edge_mul(a, b, ii, jj), (T(N, F), T(N, F))),
        "radial_d gaussian": (lambda d, a: ops.radial_d(d, a, 0, p0, p1, 5.0, 0), (d, T(E))),
        "radial_d gaussian' without a": (lambda d: ops.radial_d(d, None, 0, p0, p1, 5.0, 1), (d,)),
        "radial_d bessel": (lambda d, a: ops.radial_d(d, a, 1, fr, None, 5.0, 0), (d, T(E))),
        # GaussianRBF(trainable=True), nn/radial.py:40-45: offsets and widths are operands with gradients, to second order
        "radial_d gaussian, trainable offsets / widths": (lambda d, a, mu, w: ops.radial_d(d, a, 0, mu, w, 5.0, 0),
                                                          (d, T(E), p0.clone().requires_grad_(), p1.clone().requires_grad_())),
        "radial_d gaussian', trainable, [E, 1] distances": (lambda d1, mu, w: ops.radial_d(d1, None, 0, mu, w, 5.0, 1),
                                                            (d.detach().unsqueeze(1).requires_grad_(), p0.clone().requires_grad_(), p1.clone().requires_grad_())),
        "radial_c gaussian, trainable": (lambda G, d, a, mu, w: ops.radial_c(G, d, a, 0, mu, w, 5.0, 0),
                                         (T(E, R), d, T(E), p0.clone().requires_grad_(), p1.clone().requires_grad_())),
        "radial_c gaussian', trainable": (lambda G, d, mu, w: ops.radial_c(G, d, None, 0, mu, w, 5.0, 1),
                                          (T(E, R), d, p0.clone().requires_grad_(), p1.clone().requires_grad_())),
        "radial_d cutoff": (lambda d, a: ops.radial_d(d, a, 2, p0, None, 5.0, 0), (d, T(E))),
        "radial_d cutoff without a": (lambda d: ops.radial_d(d, None, 2, p0, None, 5.0, 0), (d,)),
        "radial_c gaussian": (lambda G, d, a: ops.radial_c(G, d, a, 0, p0, p1, 5.0, 0), (T(E, R), d, T(E))),
        "radial_c bessel'": (lambda G, d: ops.radial_c(G, d, None, 1, fr, None, 5.0, 1), (T(E, R), d)),
        "edge_mul, identity index on a": (lambda a, b: ops.edge_mul(a, b, None, jj), (T(E, F), T(N, F))),
        "cfconv, identity source": (lambda x, W: ops.cfconv(x, W, ii, None, N), (T(E, F), T(E, F))),
        "cfconv, identity output": (lambda x, W: ops.cfconv(x, W, None, jj, E), (T(N, F), T(E, F))),
        "vscale": (lambda V, s: ops.vec3(0, V, s), (T(N, 3, F), T(N, 1, F))),
        "vscale on halves of split tensors": (lambda V, s: ops.vec3(0, V[..., F:], s[..., :F]), (T(N, 3, 2 * F), T(N, 1, 3 * F))),
        "vdot": (lambda A, B: ops.vec3(1, A, B), (T(N, 3, F), T(N, 3, F))),
        "vdot on the two halves of one tensor": (lambda A: ops.vec3(1, A[..., :F], A[..., F:]), (T(N, 3, 2 * F),)),
        "vouter": (lambda s, u: ops.vec3(2, s, u), (T(E, 1, F), T(E, 3))),
        "vcontract": (lambda G, u: ops.vec3(3, G, u), (T(E, 3, F), T(E, 3))),
        "vrowdot": (lambda G, s: ops.vec3(4, G, s), (T(E, 3, F), T(E, 1, F))),
        "rowscale": (lambda W, s: ops.rowscale(W, s), (T(E, F), T(E))),
        "rowdot": (lambda a, b: ops.rowdot(a, b), (T(E, F), T(E, F))),
        "edge_norm": (lambda r: ops.edge_norm(r), (T(E, 3),)),
        "dense ssp": (lambda x, w, b: ops.dense(x, w, b, 1), (T(7, 4), T(3, 4), T(3))),
        "dense silu, no bias": (lambda x, w: ops.dense(x, w, None, 2), (T(7, 4), T(3, 4))),
        "dense linear": (lambda x, w, b: ops.dense(x, w, b, 0), (T(7, 4), T(3, 4), T(3))),
        "scatter_add": (lambda x: ops.scatter_add(x, ii, N, 0), (T(E, F),)),
        "gather": (lambda x: ops.gather(x, jj, 0), (T(N, F),)),
        "pairwise": (lambda Rr, off: ops.pairwise(Rr, ii, jj, off), (T(N, 3), T(E, 3))),
    }


CASES = _cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_operator_is_differentiable_to_second_order(name):
    fn, inp = CASES[name]
    with crk.reference_kernels():
        assert gradcheck(fn, inp, eps=1e-6, atol=1e-6, rtol=1e-5)
        assert gradgradcheck(fn, inp, eps=1e-6, atol=1e-6, rtol=1e-5)


def test_refusal_is_back_after_the_context():
    with crk.reference_kernels():
        ops.rowdot(torch.ones(2, 2), torch.ones(2, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rowdot(torch.ones(2, 2), torch.ones(2, 2))


def test_only_the_gaussian_basis_has_trainable_parameters():
    """Bessel frequencies are buffers in the reference (nn/radial.py:99-103): a frequency tensor that requires grad is refused."""
    fr = (torch.arange(1, 7, dtype=D) * 0.6).requires_grad_()
    with crk.reference_kernels(), pytest.raises(RuntimeError, match="only the Gaussian basis"):
        ops.radial_d(torch.rand(4, dtype=D), None, 1, fr, None, 5.0, 0)
    p0 = torch.linspace(0.5, 4.0, 6, dtype=D, requires_grad=True)
    with crk.reference_kernels():
        out = ops.radial_d(torch.rand(4, dtype=D), None, 0, p0, torch.ones(6, dtype=D), 5.0, 0)
        out.sum().backward()
    assert p0.grad is not None and torch.isfinite(p0.grad).all()


@pytest.mark.parametrize("kind,radial", [("schnet", "gaussian"), ("schnet", "bessel"), ("painn", "gaussian"), ("painn", "bessel"),
                                         ("schnet", "gaussian_trainable"), ("painn", "gaussian_trainable")])
def test_force_matching_step_matches_the_oracle(kind, radial):
    """energy -> forces (create_graph) -> loss -> weight gradients, whole model in training mode, float64.  `gaussian_trainable`:
    GaussianRBF(trainable=True) -- the gradients w.r.t. offsets and widths come from the same closed operators."""
    F, n_rbf = 32, 8
    trainable = radial.endswith("_trainable")
    radial = radial.split("_")[0]
    b = S.molecule_batch("aspirin", 3, seed=12)
    rep_p = O.init_schnet_params(F, 3, n_rbf, 5.0, radial=radial) if kind == "schnet" else O.init_painn_params(F, 3, n_rbf, 5.0, radial=radial)
    head_p = O.init_atomwise_params(F, seed=1)
    g = torch.Generator().manual_seed(0)
    Et = torch.randn(3, generator=g).double()
    Ft = torch.randn(b["Z"].shape[0], 3, generator=g).double()
    trained = ("weight", "bias", "radial_basis.offsets", "radial_basis.widths") if trainable else ("weight", "bias")
    rp = {k: (v.clone().double().requires_grad_(True) if v.is_floating_point() and k.endswith(trained) else
              (v.double() if v.is_floating_point() else v)) for k, v in rep_p.items()}
    hp = {k: v.clone().double().requires_grad_(True) for k, v in head_p.items()}
    R = b["R"].clone().double().requires_grad_(True)
    r_ij = O.pairwise_vectors(R, b["idx_i"], b["idx_j"], b["offsets"].double())
    if kind == "schnet":
        x = O.schnet_representation(b["Z"], r_ij, b["idx_i"], b["idx_j"], rp, 3)
    else:
        x, _ = O.painn_representation(b["Z"], r_ij, b["idx_i"], b["idx_j"], rp, 3)
    E = O.atomwise_energy(x, b["idx_m"], 3, hp)
    (dEdR,) = torch.autograd.grad([E.sum()], [R], create_graph=True)
    loss_o = 0.01 * ((E - Et) ** 2).mean() + 0.99 * ((-dEdR - Ft) ** 2).mean()
    names = [k for k, v in rp.items() if torch.is_tensor(v) and v.requires_grad]
    go = dict(zip(names, torch.autograd.grad(loss_o, [rp[k] for k in names], allow_unused=True)))
    if trainable:
        assert go["radial_basis.offsets"] is not None and float(go["radial_basis.offsets"].abs().max()) > 0 and float(go["radial_basis.widths"].abs().max()) > 0

    model = M.build_model(kind, F, 3, n_rbf, 5.0, radial, trainable_rbf=trainable)
    assert model.fm_engine == (not trainable) and model.representation._fused      # trainable bases: closed operators in training, fused kernels in eval
    M.load_reference_params(model, rep_p, head_p)
    model = model.double().train()
    inp = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in M.batch_to_inputs(b, torch.device("cpu")).items()}
    calls = collections.Counter()
    saved = dict(crk.KERNELS)

    def counted(name, fn):
        def w(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return w
    crk.KERNELS = {n: counted(n, f) for n, f in saved.items()}
    try:
        with crk.reference_kernels():
            out = model(inp)
            loss = 0.01 * ((out["energy"] - Et) ** 2).mean() + 0.99 * ((out["forces"] - Ft) ** 2).mean()
            loss.backward()
    finally:
        crk.KERNELS = saved
    assert abs(float(loss.detach()) - float(loss_o.detach())) <= 1e-12 * abs(float(loss_o.detach()))
    got = dict(model.representation.named_parameters())
    for k in names:
        if go[k] is not None:
            assert float((got[k].grad - go[k]).abs().max()) <= 1e-10 * float(go[k].abs().max()), k
    # the whole step is a few hundred operator calls of the HIP family (SchNet: Dense 5 x 3 + head, each forward / backward /
    # twice-backward = linear, matmul_nn, matmul_tn, act_mul; no torch matmul in between)
    if kind == "schnet" and not trainable:
        assert sum(calls.values()) <= 190, dict(calls)
        # weight gradients ride with the input gradient of the same layer in one launch wherever a pass needs both
        assert calls["gemm_pair"] >= 25 and calls["matmul_tn"] + calls["gemm_pair"] >= 30 and calls["cfconv"] >= 12, dict(calls)
