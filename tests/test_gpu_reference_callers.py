"""The REFERENCE's own callers on the device: after ``schnetpack_amd.install.install()`` the
reference's ``NeuralNetworkPotential`` (model/base.py:174-190), ``Atomwise`` (atomistic/atomwise.py:69-88) and
``Forces`` (atomistic/response.py:59-92) -- their code, not the mirrors in ``schnetpack_amd/{model,atomistic}.py`` --
run on top of the HIP-backed ``SchNet`` / ``PaiNN`` / ``PairwiseDistances`` / ``Dense`` / ``scatter_add`` on ``cuda:0``
and must reproduce the same reference model evaluated with the reference's own classes on the host CPU.

The reference package is imported through ``oracle/refshim.py``: from ``/root/reference`` where that exists, from its
byte-compiled build ``oracle/_ref`` on the GPU box (``oracle/build_ref.py``).  Tolerance: 1e-5 relative (north_star).
"""
import sys

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import build_ref, refshim
from schnetpack_amd import synthetic as S

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not refshim.available(), reason="neither /root/reference nor oracle/_ref present")]
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture()
def ns():
    """Reference namespace; whatever a test installs is removed again."""
    import schnetpack_amd.install as inst
    n = refshim.load()
    sys.modules["ase.data"].atomic_masses = np.ones(119)      # transform/atomistic.py:4 reads it at import
    yield n
    inst.uninstall()


def _ref_inputs(b, device="cpu"):
    n_mol = int(b["n_mol"])
    d = {"_atomic_numbers": b["Z"], "_positions": b["R"].clone(), "_idx_i": b["idx_i"],
         "_idx_j": b["idx_j"], "_offsets": b["offsets"], "_idx_m": b["idx_m"],
         "_cell": b["cell"].reshape(1, 3, 3) if "cell" in b else torch.zeros(n_mol, 3, 3),
         "_pbc": torch.zeros(3 * n_mol, dtype=torch.bool),
         "_n_atoms": torch.bincount(b["idx_m"], minlength=n_mol)}
    return {k: v.to(device) for k, v in d.items()}


def _build_reference_model(ns, kind, seed=0, radial="gaussian", **rep_kw):
    """configs/model/nnp.yaml:4-8 + experiment/md17.yaml:30-38 with whatever classes the reference namespace holds NOW."""
    import schnetpack as spk       # the shim package
    torch.manual_seed(seed)
    rb = spk.nn.GaussianRBF(20, 5.0) if radial == "gaussian" else spk.nn.BesselRBF(20, 5.0)
    cf = spk.nn.CosineCutoff(5.0)
    rep_cls = sys.modules["schnetpack.representation." + kind].__dict__["SchNet" if kind == "schnet" else "PaiNN"]
    rep = rep_cls(128, 3, rb, cf, **rep_kw)
    aw = sys.modules["schnetpack.atomistic.atomwise"].Atomwise(n_in=128, output_key="energy")
    pd = sys.modules["schnetpack.atomistic.distances"].PairwiseDistances()
    return ns.model.NeuralNetworkPotential(rep, input_modules=[pd], output_modules=[aw, ns.response.Forces()])


@pytest.mark.parametrize("fused_head", [False, True])
@pytest.mark.parametrize("kind,radial,rep_kw", [("schnet", "gaussian", {}), ("painn", "gaussian", {}),
                                                 ("painn", "bessel", {"shared_filters": True}),
                                                 ("schnet", "bessel", {"n_filters": 64})])
def test_reference_callers_on_hip_classes_match_reference_on_cpu(dev, ns, kind, radial, rep_kw, fused_head):
    import schnetpack_amd.install as inst
    from schnetpack_amd import atomistic as A, nn as N, representation as R
    b = S.molecule_batch("aspirin", 6, seed=21)
    m_ref = _build_reference_model(ns, kind, radial=radial, **rep_kw).eval()
    assert type(m_ref.representation).__module__.startswith("schnetpack.representation")
    out_ref = m_ref(_ref_inputs(b))

    inst.install(sys.modules["schnetpack"], fused_head=fused_head)
    m_hip = _build_reference_model(ns, kind, radial=radial, **rep_kw)
    # the callers are the reference's code, the hot path is ours
    assert type(m_hip) is ns.model.NeuralNetworkPotential and type(m_hip.output_modules[1]) is ns.response.Forces
    assert isinstance(m_hip.representation, R.SchNet if kind == "schnet" else R.PaiNN)
    assert isinstance(m_hip.input_modules[0], A.PairwiseDistances)
    assert isinstance(m_hip.output_modules[0], A.Atomwise) == fused_head
    assert isinstance(m_hip.output_modules[0].outnet[0], N.Dense)
    sd_ref, sd_hip = m_ref.state_dict(), m_hip.state_dict()
    assert set(sd_ref) == set(sd_hip) and all(torch.equal(sd_ref[k], sd_hip[k]) for k in sd_ref)   # same seeded init
    m_hip = m_hip.to(dev).eval()
    out = m_hip(_ref_inputs(b, dev))
    assert rel_err(out["energy"].cpu(), out_ref["energy"]) < TOL
    assert rel_err(out["forces"].cpu(), out_ref["forces"]) < TOL


def test_unpickled_lammps_example_model_runs_on_hip_classes(dev, ns):
    """interfaces/lammps/examples/aspirin/best_model (PaiNN, 2 interactions, AddOffsets + CastTo64 postprocessors):
    unpickled once with the reference's classes (CPU) and once, after install(), into the mirrors (device)."""
    import schnetpack_amd.install as inst
    from schnetpack_amd import representation as R
    path = build_ref.data_path("lammps_aspirin_best_model")
    load_model = sys.modules["schnetpack.utils"].load_model if hasattr(sys.modules["schnetpack.utils"], "load_model") \
        else __import__("schnetpack.utils.compatibility", fromlist=["load_model"]).load_model
    b = S.molecule_batch("aspirin", 3, seed=5, jitter=0.03)
    m_ref = load_model(path).eval()
    assert not isinstance(m_ref.representation, R.PaiNN)
    out_ref = m_ref(_ref_inputs(b))

    inst.install(sys.modules["schnetpack"])
    m_hip = load_model(path)
    assert isinstance(m_hip.representation, R.PaiNN)
    m_hip = m_hip.to(dev).eval()
    out = m_hip(_ref_inputs(b, dev))
    assert out["energy"].dtype == out_ref["energy"].dtype            # CastTo64 postprocessor ran in both
    assert rel_err(out["forces"].cpu(), out_ref["forces"]) < TOL
    # energies are ~ -4e5 kcal/mol totals: compare to 2 fp32 ulp of the total like tests/test_deploy.py
    assert float((out["energy"].cpu() - out_ref["energy"]).abs().max()) <= 2 * 2.0 ** -23 * float(out_ref["energy"].abs().max()) + 1e-4


def test_reference_callers_training_mode_weight_gradients(dev, ns):
    """Reference Forces(create_graph=training) + reference Atomwise around the HIP classes in train() mode: the
    force-matching loss and its parameter gradients equal the all-reference model's on the CPU."""
    import schnetpack_amd.install as inst
    b = S.molecule_batch("aspirin", 2, seed=8)
    g = torch.Generator().manual_seed(3)
    Et, Ft = torch.randn(2, generator=g), torch.randn(b["Z"].shape[0], 3, generator=g)

    def loss_of(model, device):
        out = model(_ref_inputs(b, device))
        return 0.01 * ((out["energy"] - Et.to(device)) ** 2).mean() + 0.99 * ((out["forces"] - Ft.to(device)) ** 2).mean()

    m_ref = _build_reference_model(ns, "painn").double().train()
    bd = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
    out = m_ref(_ref_inputs(bd))
    l_ref = 0.01 * ((out["energy"] - Et.double()) ** 2).mean() + 0.99 * ((out["forces"] - Ft.double()) ** 2).mean()
    l_ref.backward()
    inst.install(sys.modules["schnetpack"])
    m_hip = _build_reference_model(ns, "painn").to(dev).train()
    l_hip = loss_of(m_hip, dev)
    l_hip.backward()
    assert abs(float(l_hip) - float(l_ref)) / abs(float(l_ref)) < 1e-5
    gr = dict(m_ref.named_parameters())
    worst = 0.0
    for k, p in m_hip.named_parameters():
        if gr[k].grad is None:
            continue
        assert p.grad is not None, k
        worst = max(worst, rel_err(p.grad.cpu(), gr[k].grad))
    assert worst < 2e-4, worst     # fp32 second-order accumulations against the fp64 reference


def test_unpickled_model_with_fused_head_install(dev, ns):
    """Round-2 ADVICE: after install(fused_head=True) a reference-pickled model maps its Atomwise onto the mirror class without
    running __init__ -- __setstate__ has to supply `_fused_head`, `_head_act`, `n_molecules_key`."""
    import schnetpack_amd.install as inst
    from schnetpack_amd import atomistic as A
    path = build_ref.data_path("lammps_aspirin_best_model")
    load_model = sys.modules["schnetpack.utils"].load_model if hasattr(sys.modules["schnetpack.utils"], "load_model") \
        else __import__("schnetpack.utils.compatibility", fromlist=["load_model"]).load_model
    b = S.molecule_batch("aspirin", 3, seed=5, jitter=0.03)
    out_ref = load_model(path).eval()(_ref_inputs(b))
    inst.install(sys.modules["schnetpack"], fused_head=True)
    m_hip = load_model(path)
    head = m_hip.output_modules[0]
    assert isinstance(head, A.Atomwise) and head._fused_head and head.n_molecules_key == "_n_molecules"
    out = m_hip.to(dev).eval()(_ref_inputs(b, dev))
    assert rel_err(out["forces"].cpu(), out_ref["forces"]) < TOL


def test_fused_potential_install_routes_the_reference_model_to_the_two_launch_operator(dev, ns):
    """install(fused_head=True, fused_potential=True): the REFERENCE's NeuralNetworkPotential.forward hands the standard
    potential (PairwiseDistances -> SchNet -> Atomwise -> Forces) to the fused operators in eval mode -- same energies and forces
    as the reference on the CPU, exactly two profiled launches per call, SchNet and PaiNN -- and keeps its own forward for
    everything else (training mode, other compositions)."""
    import schnetpack_amd.install as inst
    from schnetpack_amd import _lib
    b = S.molecule_batch("aspirin", 6, seed=21)
    m_ref = _build_reference_model(ns, "schnet").eval()
    out_ref = m_ref(_ref_inputs(b))
    out_pref = _build_reference_model(ns, "painn").eval()(_ref_inputs(b))
    inst.install(sys.modules["schnetpack"], fused_head=True, fused_potential=True)
    m = _build_reference_model(ns, "schnet").to(dev).eval()
    assert type(m) is ns.model.NeuralNetworkPotential and getattr(type(m).__call__, "_spk_hip_patched", False)
    assert not hasattr(type(m).forward, "_spk_hip_patched")      # the class forward stays the reference's (scriptable)
    out = m(_ref_inputs(b, dev))            # first call: plan
    _lib.profile_enable(True)
    _lib.profile_report()
    out = m(_ref_inputs(b, dev))
    prof = _lib.profile_report()
    _lib.profile_enable(False)
    assert m.__dict__["_spk_hip_mode"] == 2
    assert set(prof) == {"schnet_mol_fwd", "schnet_mol_bwd"}, prof
    assert set(out) == set(out_ref)
    assert rel_err(out["energy"].cpu(), out_ref["energy"]) < TOL
    assert rel_err(out["forces"].cpu(), out_ref["forces"]) < TOL
    # training mode: the reference's forward (differentiable primitives, create_graph)
    m.train()
    o = m(_ref_inputs(b, dev))
    assert o["forces"].requires_grad
    # the PaiNN standard potential is routed the same way (torch.ops.spk_hip.painn_potential_forces) ...
    mp_ = _build_reference_model(ns, "painn").to(dev).eval()
    op = mp_(_ref_inputs(b, dev))
    _lib.profile_enable(True)
    _lib.profile_report()
    op = mp_(_ref_inputs(b, dev))
    prof_p = _lib.profile_report()
    _lib.profile_enable(False)
    assert mp_.__dict__["_spk_hip_mode"] == 2 and set(prof_p) == {"painn_mol_fwd", "painn_mol_bwd"}, prof_p
    assert rel_err(op["energy"].cpu(), out_pref["energy"]) < TOL and rel_err(op["forces"].cpu(), out_pref["forces"]) < TOL
    # ... and a model that is not the standard potential (stress requested) is left alone
    ms = _build_reference_model(ns, "schnet").to(dev).eval()
    ms.output_modules[1].calc_stress = True
    from schnetpack_amd import model as M
    assert M.classify_potential(ms) == 0
    inst.uninstall()
    assert not getattr(ns.model.NeuralNetworkPotential.__call__, "_spk_hip_patched", False)
