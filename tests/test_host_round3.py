"""Host logic added in round 3 (no GPU): routing of the standard potential, the wire format's host-resident plan meta, the
bookkeeping of the bench line."""
import importlib.util
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_classify_potential_routes_both_models_and_leaves_other_compositions():
    from schnetpack_amd import model as M
    from schnetpack_amd.atomistic import Atomwise, Forces, PairwiseDistances
    for kind in ("schnet", "painn"):
        m = M.build_model(kind)
        assert M.classify_potential(m) == 2 and m._potential_forces          # energies and forces: the two-launch operator
        m.output_modules[1].calc_stress = True
        assert M.classify_potential(m) == 0                                  # stress wanted: the reference composition runs
    # energy head only (no Forces module): SchNet has the differentiable one-operator form, PaiNN keeps module by module
    rep = M.build_model("schnet").representation
    only_e = M.NeuralNetworkPotential(rep, input_modules=[PairwiseDistances()], output_modules=[Atomwise(n_in=128, output_key="energy")])
    assert M.classify_potential(only_e) == 1
    rep_p = M.build_model("painn").representation
    only_e = M.NeuralNetworkPotential(rep_p, input_modules=[PairwiseDistances()], output_modules=[Atomwise(n_in=128, output_key="energy")])
    assert M.classify_potential(only_e) == 0
    # an averaged energy cannot take the forces form (its Forces module differentiates the mean)
    m = M.build_model("schnet")
    m.output_modules[0].aggregation_mode = "avg"
    assert M.classify_potential(m) == 1


def test_to_device_keeps_the_plan_meta_on_the_host_and_install_refuses_a_moved_one():
    from schnetpack_amd import data as D
    batch = {"_idx_i": torch.zeros(4, dtype=torch.int64), "_spk_plan_meta": torch.arange(8), "_spk_rowptr": torch.zeros(3, dtype=torch.int32)}
    moved = D.to_device(batch, torch.device("meta"))
    assert moved["_spk_plan_meta"].device.type == "cpu" and moved["_idx_i"].device.type == "meta"
    bad = dict(moved)
    bad["_spk_plan_meta"] = batch["_spk_plan_meta"].to("meta")
    with pytest.raises(ValueError, match="keep it on the host"):
        D.install_plan(bad)
    assert D.install_plan({"_idx_i": batch["_idx_i"]}) is False          # a batch without a host-made plan


def test_bench_books_every_modelled_kernel_with_a_lower_bound():
    B = _bench()
    for kind in ("schnet", "painn"):
        algo = B.algorithmic_work(kind, 77944, 5376, 256, 128, 3, 20)
        assert algo, kind
        for tag, (bound, work, executed, bmin) in algo.items():
            assert bound in ("mfma", "hbm") and work > 0 and 0 < executed <= 1.0, tag
            assert bmin is None or 0 < bmin, tag
    assert any(t.startswith("painn_mol") for t in B.algorithmic_work("painn", 77944, 5376, 256, 128, 3, 20))
    assert 0.05 <= B.RAMP_S <= 0.5          # the untimed clock ramp stays a bounded fraction of a second


def test_documented_second_order_of_the_filter_node():
    """HISTORY.md 4.12 writes down the first and second derivative of the SchNet filter node (schnet.py:60-62) for the fused training
    node that is not built yet: the formulas, restated in float64, against torch autograd."""
    import math
    torch.manual_seed(0)
    E, K, nf, rc, dt = 9, 6, 8, 5.0, torch.float64
    d = (0.5 + 3 * torch.rand(E, dtype=dt)).requires_grad_(True)
    mu, c = torch.linspace(0, 4, K, dtype=dt), -0.5 / 0.3 ** 2
    W1, b1 = torch.randn(nf, K, dtype=dt, requires_grad=True), torch.randn(nf, dtype=dt, requires_grad=True)
    W2, b2 = torch.randn(nf, nf, dtype=dt, requires_grad=True), torch.randn(nf, dtype=dt, requires_grad=True)
    G, h = torch.randn(E, nf, dtype=dt, requires_grad=True), torch.randn(E, dtype=dt)
    t = d[:, None] - mu[None]
    phi = torch.exp(c * t * t)
    dphi = 2 * c * t * phi
    ddphi = 2 * c * phi + 2 * c * t * dphi
    fc = 0.5 * (torch.cos(d * math.pi / rc) + 1)
    dfc = -0.5 * math.pi / rc * torch.sin(d * math.pi / rc)
    ddfc = -0.5 * (math.pi / rc) ** 2 * torch.cos(d * math.pi / rc)
    a = phi @ W1.t() + b1
    z = torch.nn.functional.softplus(a) - math.log(2.0)
    g = z @ W2.t() + b2
    W = g * fc[:, None]
    gd_auto, = torch.autograd.grad(W, d, G, create_graph=True)
    sig = torch.sigmoid(a)
    ap, app = dphi @ W1.t(), ddphi @ W1.t()
    gp = (sig * ap) @ W2.t()
    assert torch.allclose((G * (fc[:, None] * gp + dfc[:, None] * g)).sum(1), gd_auto, rtol=1e-12, atol=1e-12)
    auto = torch.autograd.grad((h * gd_auto).sum(), (G, d, W1, b1, W2, b2))
    sigp = sig * (1 - sig)
    gpp = (sigp * ap * ap + sig * app) @ W2.t()
    u, v = h[:, None] * fc[:, None] * G, h[:, None] * dfc[:, None] * G
    ub, vb = u @ W2, v @ W2
    ab, pb = vb * sig + ub * sigp * ap, ub * sig
    mine = (h[:, None] * (fc[:, None] * gp + dfc[:, None] * g),
            h * (G * (fc[:, None] * gpp + 2 * dfc[:, None] * gp + ddfc[:, None] * g)).sum(1),
            ab.t() @ phi + pb.t() @ dphi, ab.sum(0), u.t() @ (sig * ap) + v.t() @ z, v.sum(0))
    for x, y in zip(mine, auto):
        assert torch.allclose(x.detach(), y, rtol=1e-10, atol=1e-12)
