"""CPU: pin the oracle against the LIVE reference modules (from /root/reference, or from its byte-compiled
build under oracle/_ref where /root/reference does not exist; skipped when neither is there)."""
import pytest
import torch

from oracle import refshim, spk_oracle as O
from schnetpack_amd import synthetic as S

pytestmark = pytest.mark.skipif(not refshim.available(), reason="reference sources not present")


def _ref_inputs(b):
    n_mol = int(b["n_mol"])
    return {"_atomic_numbers": b["Z"], "_positions": b["R"].clone(), "_idx_i": b["idx_i"],
            "_idx_j": b["idx_j"], "_offsets": b["offsets"], "_idx_m": b["idx_m"],
            "_cell": torch.zeros(n_mol, 3, 3), "_pbc": torch.zeros(3 * n_mol, dtype=torch.bool),
            "_n_atoms": torch.bincount(b["idx_m"], minlength=n_mol)}


@pytest.mark.parametrize("kind", ["schnet", "painn"])
@pytest.mark.parametrize("shared", [False, True])
def test_seeded_init_and_force_call(kind, shared):
    ns = refshim.load()
    torch.manual_seed(0)
    rb, cf = ns.nn.GaussianRBF(20, 5.0), ns.nn.CosineCutoff(5.0)
    if kind == "schnet":
        if shared:
            pytest.skip("shared_interactions aliases modules; covered by module tests")
        rep = ns.schnet.SchNet(128, 3, rb, cf)
        p = O.init_schnet_params()
    else:
        rep = ns.painn.PaiNN(128, 3, rb, cf, shared_filters=shared)
        p = O.init_painn_params(shared_filters=shared)
    sd = rep.state_dict()
    assert set(sd) == set(p)
    assert all(torch.equal(sd[k], p[k]) for k in sd)
    torch.manual_seed(1)
    aw = ns.atomwise.Atomwise(n_in=128, output_key="energy")
    head = O.init_atomwise_params(128, seed=1)
    assert all(torch.equal(v, head[k]) for k, v in aw.state_dict().items())
    model = ns.model.NeuralNetworkPotential(rep, input_modules=[ns.distances.PairwiseDistances()],
                                            output_modules=[aw, ns.response.Forces()])
    model.eval()
    b = S.molecule_batch("aspirin", 4, seed=11)
    out = model(_ref_inputs(b))
    o = O.energy_and_forces(kind, p, head, b, 3, shared_filters=shared)
    assert (out["energy"] - o["energy"]).abs().max() / out["energy"].abs().max() < 2e-6
    assert (out["forces"] - o["forces"]).abs().max() / out["forces"].abs().max() < 5e-6


def test_reference_golden_nn_tests_hold_for_oracle():
    """The reference L0 modules and the oracle functions agree on random input."""
    ns = refshim.load()
    d = torch.rand(50) * 6
    torch.testing.assert_close(ns.nn.GaussianRBF(20, 5.0)(d), O.gaussian_rbf(d, *O.gaussian_rbf_params(20, 5.0)))
    torch.testing.assert_close(ns.nn.BesselRBF(20, 5.0)(d), O.bessel_rbf(d, O.bessel_rbf_params(20, 5.0).float()))
    torch.testing.assert_close(ns.nn.CosineCutoff(5.0)(d), O.cosine_cutoff(d, 5.0))
    x = torch.randn(100) * 10
    torch.testing.assert_close(ns.nn.shifted_softplus(x), O.shifted_softplus(x))


def test_synthetic_neighbor_list_matches_reference_torch_list():
    """Index parity of the synthetic generator's list vs TorchNeighborList (bit-exact as sets
    per centre atom; the reference's argsort is not stable so order inside a row may differ)."""
    ns = refshim.load()
    if ns.neighborlist is None:
        pytest.skip("neighborlist module not importable")
    import numpy as np
    b = S.molecule_batch("aspirin", 1, seed=2)
    nl = ns.neighborlist.TorchNeighborList(cutoff=5.0)
    inp = {"_atomic_numbers": b["Z"], "_positions": b["R"], "_cell": torch.zeros(3, 3),
           "_pbc": torch.zeros(3, dtype=torch.bool)}
    out = nl(inp)
    ref = sorted(zip(out["_idx_i"].tolist(), out["_idx_j"].tolist()))
    mine = sorted(zip(b["idx_i"].tolist(), b["idx_j"].tolist()))
    assert ref == mine
    assert bool((out["_idx_i"][1:] >= out["_idx_i"][:-1]).all())


def test_install_hook_patches_reference_namespace_and_pickles_resolve_to_mirrors():
    """schnetpack_amd.install routes the hot-path names of the (shim-loaded) reference package to
    the HIP-backed mirrors; the shipped whole-model pickle then unpickles into the mirror classes
    with identical parameters."""
    import sys
    import numpy as np
    import schnetpack_amd.install as inst
    from schnetpack_amd import representation as R, nn as N
    ns = refshim.load()
    spk = sys.modules["schnetpack"]
    spk.representation.SchNet = ns.schnet.SchNet
    spk.representation.PaiNN = ns.painn.PaiNN
    saved = {(m, k): getattr(sys.modules[m], k) for m, k in [
        ("schnetpack.representation.painn", "PaiNN"), ("schnetpack.representation.painn", "PaiNNInteraction"),
        ("schnetpack.representation.painn", "PaiNNMixing"), ("schnetpack.representation.schnet", "SchNet"),
        ("schnetpack.representation.schnet", "SchNetInteraction"), ("schnetpack.nn", "scatter_add"),
        ("schnetpack.nn", "Dense"), ("schnetpack.nn", "GaussianRBF"), ("schnetpack.nn", "BesselRBF"),
        ("schnetpack.nn", "CosineCutoff"), ("schnetpack.nn.base", "Dense"), ("schnetpack.nn.radial", "GaussianRBF"),
        ("schnetpack.nn.radial", "BesselRBF"), ("schnetpack.nn.cutoff", "CosineCutoff"),
        ("schnetpack.nn.scatter", "scatter_add"), ("schnetpack.atomistic.distances", "PairwiseDistances"),
        ("schnetpack.atomistic.atomwise", "scatter_add") if hasattr(sys.modules["schnetpack.atomistic.atomwise"], "scatter_add") else ("schnetpack.nn", "scatter_add")]}
    aw_mod = sys.modules["schnetpack.atomistic.atomwise"]
    saved[("schnetpack.atomistic.atomwise", "Atomwise")] = aw_mod.Atomwise
    try:
        log = inst.install(spk)          # defaults since round 4: fused head + the standard potential routed at __call__
        assert "schnetpack.representation.painn.PaiNN" in log and "schnetpack.nn.scatter_add" in log
        assert "schnetpack.model.base.NeuralNetworkPotential.__call__" in log and "schnetpack.atomistic.atomwise.Atomwise" in log
        assert not hasattr(sys.modules["schnetpack.model.base"].NeuralNetworkPotential.forward, "_spk_hip_patched")
        assert sys.modules["schnetpack.representation.painn"].PaiNN is R.PaiNN
        assert spk.nn.scatter_add is N.scatter_add
        sys.modules["ase.data"].atomic_masses = np.ones(119)
        from oracle import build_ref
        path = build_ref.data_path("lammps_aspirin_best_model")
        m = torch.load(path, map_location="cpu", weights_only=False)
        assert isinstance(m.representation, R.PaiNN)
        assert isinstance(m.representation.interactions[0].interatomic_context_net[0], N.Dense)
        assert m.representation._fusable() and m.representation._eps() == pytest.approx(1e-8)
        ms, keep = None, None
        assert m.representation.filter_net.weight.shape == (768, 20)
        # the unpickled reference model (reference NeuralNetworkPotential + Atomwise + Forces + AddOffsets around the
        # mirror representation) exports for the torch-free runtime: what `spkdeploy best_model deployed` is for
        import struct
        from schnetpack_amd import deploy
        blob = deploy.export_potential(m.eval())
        ints = struct.unpack("<16i", blob[8:72])
        flts = struct.unpack("<4f", blob[72:88])
        assert ints[:6] == (1, 1, 128, 128, 2, 20) and ints[7] == 64 and ints[10] == 1 and ints[11] == 0
        assert flts[0] == pytest.approx(5.0) and flts[2] == pytest.approx(float(m.postprocessors[1].mean), rel=1e-6)
        # opt-in extras: fused energy head and the device neighbour lists
        from schnetpack_amd import atomistic as A, neighborlist as NL
        log2 = inst.install(spk, fused_head=True, neighbor_lists=True)
        assert aw_mod.Atomwise is A.Atomwise and "schnetpack.atomistic.atomwise.Atomwise" in log2
        if ns.neighborlist is not None:
            assert sys.modules["schnetpack.transform.neighborlist"].HipNeighborList is NL.HipNeighborList
            delattr(sys.modules["schnetpack.transform.neighborlist"], "HipNeighborList")
    finally:
        inst.uninstall()
        assert "__call__" not in sys.modules["schnetpack.model.base"].NeuralNetworkPotential.__dict__
        for (mod, k), v in saved.items():
            setattr(sys.modules[mod], k, v)
        spk.representation.SchNet = ns.schnet.SchNet
        spk.representation.PaiNN = ns.painn.PaiNN


def test_nbl_oracle_equals_live_torch_neighbor_list():
    """oracle/nbl_oracle.py against the reference's TorchNeighborList class itself
    (transform/neighborlist.py:438-553) on seeded systems, incl. the transform's forward()."""
    from oracle import nbl_oracle as NB
    ns = refshim.load()
    if ns.neighborlist is None:
        pytest.skip("reference neighbour-list module not importable: %s" % ns.neighborlist_error)
    g = torch.Generator().manual_seed(5)
    cases = [
        (torch.rand(45, 3, generator=g) * 7.0, torch.diag(torch.tensor([7.0, 6.0, 8.0])), [True, True, True], 5.0),
        (torch.rand(30, 3, generator=g) * 9.0, torch.tensor([[9.0, 0, 0], [2.0, 8.0, 0], [1.0, -1.0, 9.5]]), [True, False, True], 4.0),
        (torch.rand(8, 3, generator=g) * 3.0, torch.diag(torch.tensor([3.2, 3.4, 3.1])), [True, True, True], 5.0),
        (torch.randn(50, 3, generator=g) * 3.0, torch.zeros(3, 3), [False, False, False], 3.5),
    ]
    for R, cell, pbc, rc in cases:
        pbc = torch.tensor(pbc)
        tnl = ns.neighborlist.TorchNeighborList(rc)
        out = tnl({"_atomic_numbers": torch.ones(R.shape[0], dtype=torch.long), "_positions": R, "_cell": cell.reshape(1, 3, 3), "_pbc": pbc})
        i, j, off = out["_idx_i"], out["_idx_j"], out["_offsets"]
        assert bool((i[1:] >= i[:-1]).all())
        S = torch.round(off @ torch.linalg.inv(cell)).long() if bool(pbc.any()) else torch.zeros(i.shape[0], 3, dtype=torch.long)
        order = NB.canonical_order(i, j, S)
        oi, oj, oS, oo = NB.neighbor_list(R, cell, pbc, rc)
        assert torch.equal(i[order], oi) and torch.equal(j[order], oj) and torch.equal(S[order], oS)
        assert torch.allclose(off[order], oo, atol=1e-6)
