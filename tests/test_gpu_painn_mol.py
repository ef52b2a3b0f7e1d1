"""Molecule-resident PaiNN kernels (schnetpack_amd/csrc/spk_painn_mol.hip): batches of small molecules are block diagonal
(data/loader.py:35-46), so a group of <= 32 atoms runs all interactions (painn.py:207-256) inside one workgroup with q / mu / context
rows in LDS.  Checked against the CPU oracle in float64 and against the general driver (``SPK_NO_PAINN_MOL`` disables the molecule
path; the profile tags assert which path ran).  Tolerance 1e-5 relative (north_star)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S
from test_gpu_mol import _mixed_batch

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run(batch, dev, n_int=3, n_rbf=20, radial="gaussian", general=False, potential=False, **rep_kw):
    """potential=False: representation operator + head + Forces (the kernels' plain instances); True: the standard potential as ONE
    operator (pair vectors, embedding rows, head and forces inside the two launches)."""
    from schnetpack_amd import _lib, model as M
    rep = O.init_painn_params(128, n_int, n_rbf, 5.0, radial=radial, **rep_kw)
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model("painn", 128, n_int, n_rbf, 5.0, radial, **rep_kw)
    M.load_reference_params(m, rep, head)
    m = m.to(dev).eval()
    if general:
        os.environ["SPK_NO_PAINN_MOL"] = "1"
    if not potential:
        os.environ["SPK_NO_POTENTIAL"] = "1"
    try:
        _lib.profile_enable(True)
        _lib.profile_report()
        inp = M.batch_to_inputs(batch, dev)
        out = m(inp)
        res = (out["energy"].detach().cpu(), out["forces"].detach().cpu(), inp["scalar_representation"].detach().cpu(),
               inp["vector_representation"].detach().cpu())
        tags = _lib.profile_report()
    finally:
        _lib.profile_enable(False)
        os.environ.pop("SPK_NO_PAINN_MOL", None)
        os.environ.pop("SPK_NO_POTENTIAL", None)
    return res, tags, (rep, head)


@pytest.mark.parametrize("sizes,n_int,n_rbf,radial,rep_kw", [
    (["aspirin"] * 7, 3, 20, "gaussian", {}),
    (["ethanol", "aspirin", "atom", "ethanol", "dimer", "ethanol", "ethanol", "aspirin", "atom", "atom"], 3, 20, "gaussian", {}),
    (["ethanol"] * 40, 2, 16, "bessel", {}),
    (["aspirin", "dimer"] * 3, 1, 8, "gaussian", {}),
    (["aspirin"] * 300, 3, 20, "gaussian", {}),          # more groups than compute units: the workgroups loop
    (["aspirin"] * 5, 2, 20, "bessel", {"shared_filters": True}),
])
def test_molecule_resident_painn_matches_oracle_and_general_driver(dev, sizes, n_int, n_rbf, radial, rep_kw):
    b = _mixed_batch(3, sizes)
    (e, f, x, v), tags, (rep, head) = _run(b, dev, n_int, n_rbf, radial, **rep_kw)
    assert "painn_mol_fwd" in tags and not any(t.startswith(("painn_msg_fwd", "painn_mixing_fwd")) for t in tags), tags      # the path under test ran
    assert "painn_mol_bwd" in tags and not any(t.startswith(("painn_msg_bwd", "painn_mixing_bwd", "chain")) for t in tags), tags
    ref = O.energy_and_forces("painn", rep, head, b, n_int, need_rep=True, dtype=torch.float64, shared_filters=bool(rep_kw.get("shared_filters")))
    assert rel_err(x, ref["scalar_representation"]) < TOL
    assert rel_err(v, ref["vector_representation"]) < TOL
    assert rel_err(e, ref["energy"]) < TOL and rel_err(f, ref["forces"]) < TOL
    (e2, f2, x2, v2), tags2, _ = _run(b, dev, n_int, n_rbf, radial, general=True, **rep_kw)
    assert "painn_mol_fwd" not in tags2 and any(t.startswith("painn_msg_fwd") for t in tags2), tags2
    assert rel_err(x, x2) < 2e-6 and rel_err(v, v2) < 2e-6 and rel_err(f, f2) < 5e-6


@pytest.mark.parametrize("sizes,n_int,n_rbf", [
    (["aspirin"] * 7, 3, 20),
    (["ethanol", "aspirin", "atom", "ethanol", "dimer", "ethanol", "ethanol", "aspirin", "atom", "atom"], 2, 16),
    (["aspirin"] * 300, 3, 20),
])
@pytest.mark.parametrize("assign", ["snake", "30"])
def test_tuning_switches_keep_parity(dev, sizes, n_int, n_rbf, assign):
    """The switches of the tuning runs stay correct code: the message on the matrix core (SPK_PM_TILED=1, an experiment that is off
    by default: HISTORY.md 4.3a) and the static atom -> wave assignments (SPK_PM_ASSIGN) against the float64 oracle and, bit for bit
    where the summation order is the same, against the default path."""
    b = _mixed_batch(5, sizes)
    (e0, f0, x0, v0), _, (rep, head) = _run(b, dev, n_int, n_rbf)
    ref = O.energy_and_forces("painn", rep, head, b, n_int, need_rep=True, dtype=torch.float64)
    os.environ["SPK_PM_ASSIGN"] = assign
    try:
        (e1, f1, x1, v1), tags, _ = _run(b, dev, n_int, n_rbf)
        os.environ["SPK_PM_TILED"] = "1"
        (e2, f2, x2, v2), tags2, _ = _run(b, dev, n_int, n_rbf)
    finally:
        os.environ.pop("SPK_PM_ASSIGN", None)
        os.environ.pop("SPK_PM_TILED", None)
    assert "painn_mol_fwd" in tags and "painn_mol_fwd" in tags2
    assert torch.equal(x0, x1) and torch.equal(v0, v1) and torch.equal(f0, f1)          # which wave takes a row does not change the row
    assert rel_err(x2, ref["scalar_representation"]) < TOL and rel_err(v2, ref["vector_representation"]) < TOL
    assert rel_err(e2, ref["energy"]) < TOL and rel_err(f2, ref["forces"]) < TOL
    assert not torch.equal(v0, v2)                                                         # (the tiled form really ran: another summation order)


@pytest.mark.parametrize("sizes,n_int,n_rbf,radial", [
    (["aspirin"] * 7, 3, 20, "gaussian"),
    (["ethanol", "aspirin", "atom", "ethanol", "dimer", "ethanol", "ethanol", "aspirin", "atom", "atom"], 3, 20, "gaussian"),
    (["ethanol"] * 40, 2, 16, "bessel"),
    (["aspirin"] * 300, 3, 20, "gaussian"),
])
def test_standard_potential_in_two_launches(dev, sizes, n_int, n_rbf, radial):
    """PairwiseDistances -> PaiNN -> Atomwise -> Forces as ONE operator (torch.ops.spk_hip.painn_potential_forces): exactly the two
    molecule kernels run, energies / forces / representations match the float64 oracle and the operator-by-operator path."""
    b = _mixed_batch(7, sizes)
    (e, f, x, v), tags, (rep, head) = _run(b, dev, n_int, n_rbf, radial, potential=True)
    assert set(tags) == {"painn_mol_fwd", "painn_mol_bwd"}, tags
    ref = O.energy_and_forces("painn", rep, head, b, n_int, need_rep=True, dtype=torch.float64)
    assert rel_err(x, ref["scalar_representation"]) < TOL and rel_err(v, ref["vector_representation"]) < TOL
    assert rel_err(e, ref["energy"]) < TOL and rel_err(f, ref["forces"]) < TOL
    (e2, f2, x2, v2), tags2, _ = _run(b, dev, n_int, n_rbf, radial, potential=False)
    assert len(tags2) > 2
    assert rel_err(e, e2) < 2e-6 and rel_err(f, f2) < 5e-6 and rel_err(x, x2) < 2e-6 and rel_err(v, v2) < 2e-6
    (e3, f3, _, _), _, _ = _run(b, dev, n_int, n_rbf, radial, potential=True)
    assert torch.equal(f, f3) and torch.equal(e, e3)                  # no atomics on the way to the forces


@pytest.mark.parametrize("kind", ["painn", "schnet"])
def test_two_launch_potential_on_small_periodic_cells(dev, kind):
    """Batches of SMALL PERIODIC systems (24-atom water cells, cutoff 2.8 A: 40 % of the pairs cross a cell face, some atom pairs are
    neighbours through two images) are block-diagonal too: the two-launch potential forms r_ij = R_j - R_i + offset inside the
    forward launch and returns forces through the reverse-edge map -- against the float64 oracle."""
    from schnetpack_amd import _lib, model as M
    cutoff = 2.8
    systems = []
    for k in range(5):
        w = S.water_box(n_side=2, cutoff=cutoff, seed=30 + k)
        systems.append({"Z": w["Z"], "R": np.asarray(w["R"]), "idx_i": np.asarray(w["idx_i"]), "idx_j": np.asarray(w["idx_j"]), "offsets": np.asarray(w["offsets"])})
    b = S.collate(systems)
    assert float(b["offsets"].abs().sum()) > 0
    rep = (O.init_painn_params if kind == "painn" else O.init_schnet_params)(128, 3, 20, cutoff)
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model(kind, 128, 3, 20, cutoff)
    M.load_reference_params(m, rep, head)
    m = m.to(dev).eval()
    _lib.profile_enable(True)
    _lib.profile_report()
    try:
        inp = M.batch_to_inputs(b, dev)
        out = m(inp)
        tags = set(_lib.profile_report())
    finally:
        _lib.profile_enable(False)
    assert tags == {kind + "_mol_fwd", kind + "_mol_bwd"}, tags
    ref = O.energy_and_forces(kind, rep, head, b, 3, dtype=torch.float64)
    assert rel_err(out["energy"].detach().cpu(), ref["energy"]) < TOL and rel_err(out["forces"].detach().cpu(), ref["forces"]) < TOL


def test_molecule_resident_painn_is_deterministic(dev):
    """No atomics anywhere in the two launches: representation AND forces are bit-reproducible."""
    b = S.molecule_batch("aspirin", 64, seed=9)
    (e1, f1, x1, v1), tags, _ = _run(b, dev)
    (e2, f2, x2, v2), _, _ = _run(b, dev)
    assert "painn_mol_fwd" in tags and "painn_mol_bwd" in tags
    assert torch.equal(x1, x2) and torch.equal(v1, v2)
    assert torch.equal(f1, f2)


@pytest.mark.parametrize("which", ["forward_only", "backward_only"])
def test_either_half_combines_with_the_general_driver(dev, which):
    """The saved tensors have the layout of the general driver: molecule-resident forward + general backward and general forward +
    molecule-resident backward give the same forces."""
    b = _mixed_batch(11, ["aspirin", "ethanol", "aspirin", "dimer", "atom", "aspirin"])
    env = "SPK_NO_PAINN_MOL_BWD" if which == "forward_only" else "SPK_NO_PAINN_MOL_FWD"
    os.environ[env] = "1"
    try:
        (e, f, x, v), tags, (rep, head) = _run(b, dev)
    finally:
        os.environ.pop(env, None)
    assert ("painn_mol_fwd" in tags) == (which == "forward_only") and ("painn_mol_bwd" in tags) == (which == "backward_only"), tags
    ref = O.energy_and_forces("painn", rep, head, b, 3, dtype=torch.float64)
    assert rel_err(f, ref["forces"]) < TOL


def test_gradient_of_a_vector_readout_and_of_the_embedding(dev):
    """The backward launch with dL/dmu_L != 0 and dL/dq0 wanted (not the eval force path): gradients of
    sum(w_q . q) + sum(w_mu . mu) w.r.t. r_ij and the embedding rows against torch autograd of the oracle in float64."""
    from schnetpack_amd import _lib, model as M
    b = _mixed_batch(5, ["aspirin", "ethanol", "aspirin"])
    rep_p = O.init_painn_params(128, 2, 20, 5.0)
    m = M.build_model("painn", 128, 2, 20, 5.0)
    M.load_reference_params(m, rep_p, O.init_atomwise_params(128, seed=1))
    rep = m.representation.to(dev).eval()
    g = torch.Generator().manual_seed(0)
    N = b["Z"].shape[0]
    wq, wm = torch.randn(N, 128, generator=g), torch.randn(N, 3, 128, generator=g)
    r = (b["R"][b["idx_j"]] - b["R"][b["idx_i"]] + b["offsets"]).float()
    # oracle, float64
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in rep_p.items()}
    r64 = r.double().requires_grad_(True)
    emb = p64["embedding.weight"].clone().requires_grad_(True)
    q, mu = O.painn_representation(b["Z"], r64, b["idx_i"], b["idx_j"], dict(p64, **{"embedding.weight": emb}), 2)
    (gr_ref, gemb_ref) = torch.autograd.grad([(q * wq.double()).sum() + (mu * wm.double()).sum()], [r64, emb])
    # device: the fused eval operator, gradient w.r.t. r_ij and the embedding table
    rd = r.to(dev).requires_grad_(True)
    rep.embedding.weight.requires_grad_(True)
    _lib.profile_enable(True); _lib.profile_report()
    try:
        out = rep({"_atomic_numbers": b["Z"].to(dev), "_Rij": rd, "_idx_i": b["idx_i"].to(dev), "_idx_j": b["idx_j"].to(dev)})
        loss = (out["scalar_representation"] * wq.to(dev)).sum() + (out["vector_representation"] * wm.to(dev)).sum()
        gr, gemb = torch.autograd.grad([loss], [rd, rep.embedding.weight])
        tags = _lib.profile_report()
    finally:
        _lib.profile_enable(False)
    assert "painn_mol_bwd" in tags, tags
    assert rel_err(gr.cpu(), gr_ref) < TOL
    assert rel_err(gemb.cpu(), gemb_ref) < TOL


def test_skin_list_pairs_beyond_the_cutoff_contribute_nothing(dev):
    """An MD list with a 2 A skin holds pairs with f_c = 0: same representation and forces as the exact list."""
    rng = np.random.RandomState(1)
    systems_skin, systems_exact = [], []
    for _ in range(6):
        R = np.asarray(S.ASPIRIN_R) + 0.05 * rng.randn(21, 3)
        for lst, rc in ((systems_skin, 7.0), (systems_exact, 5.0)):
            ii, jj = S.neighbor_pairs_open(R, rc)
            lst.append({"Z": S.ASPIRIN_Z, "R": R, "idx_i": ii, "idx_j": jj})
    bs, be = S.collate(systems_skin), S.collate(systems_exact)
    assert bs["idx_i"].shape[0] > be["idx_i"].shape[0]
    (e, f, x, v), tags, (rep, head) = _run(bs, dev)
    assert "painn_mol_fwd" in tags
    ref = O.energy_and_forces("painn", rep, head, be, 3, need_rep=True, dtype=torch.float64)
    assert rel_err(x, ref["scalar_representation"]) < TOL and rel_err(f, ref["forces"]) < TOL


def test_large_molecules_and_wide_bases_fall_back_to_the_general_driver(dev):
    rng = np.random.RandomState(0)
    R = np.concatenate([np.asarray(S.ASPIRIN_R), np.asarray(S.ASPIRIN_R) + np.array([4.0, 0.0, 0.0])]) + 0.05 * rng.randn(42, 3)
    ii, jj = S.neighbor_pairs_open(R, 5.0)
    b = S.collate([{"Z": S.ASPIRIN_Z * 2, "R": R, "idx_i": ii, "idx_j": jj}] * 3)
    (e, f, x, v), tags, (rep, head) = _run(b, dev)
    assert "painn_mol_fwd" not in tags
    assert rel_err(f, O.energy_and_forces("painn", rep, head, b, 3)["forces"]) < TOL
    b2 = _mixed_batch(2, ["aspirin"] * 4)
    (e, f, x, v), tags, (rep, head) = _run(b2, dev, 3, 32)      # n_rbf = 32 > 20: filter weights do not fit the registers
    assert "painn_mol_fwd" not in tags
    assert rel_err(f, O.energy_and_forces("painn", rep, head, b2, 3)["forces"]) < TOL


@pytest.mark.parametrize("kind", ["painn", "schnet"])
def test_atomic_number_beyond_the_embedding_table_poisons_its_molecule_and_reads_nothing(dev, kind):
    """Round-3 ADVICE: the two-launch potentials look the embedding rows up by Z inside the forward launch.  The reference's
    nn.Embedding raises an IndexError for Z >= max_z (representation/painn.py:155, schnet.py:127); a device kernel cannot raise, so
    an out-of-range Z reads NO row (it used to be an out-of-bounds read in the PaiNN kernel) and poisons exactly its molecule with
    NaN -- the other molecules of the batch are untouched."""
    from schnetpack_amd import _lib, model as M
    b = S.molecule_batch("aspirin", 3, seed=12)
    rep = (O.init_painn_params if kind == "painn" else O.init_schnet_params)()
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model(kind)
    M.load_reference_params(m, rep, head)
    m = m.to(dev).eval()
    good = m(M.batch_to_inputs(b, dev))
    e_good, f_good = good["energy"].detach().cpu(), good["forces"].detach().cpu()
    for bad_z in (100, 250, -3):
        bb = dict(b)
        bb["Z"] = b["Z"].clone()
        bb["Z"][21 + 4] = bad_z                       # an atom of the SECOND molecule
        _lib.profile_enable(True); _lib.profile_report()
        out = m(M.batch_to_inputs(bb, dev))
        tags = set(_lib.profile_report()); _lib.profile_enable(False)
        assert tags == {kind + "_mol_fwd", kind + "_mol_bwd"}, tags
        e, f = out["energy"].detach().cpu(), out["forces"].detach().cpu()
        assert torch.isnan(e[1]) and torch.isnan(f[21:42]).all()
        assert rel_err(e[[0, 2]], e_good[[0, 2]]) < 1e-6
        assert rel_err(f[:21], f_good[:21]) < 1e-6 and rel_err(f[42:], f_good[42:]) < 1e-6


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_headline_batches_repeat_bit_for_bit_over_forty_calls(dev, kind):
    """Round 6 guard (profiles/r06_box_split_glitch.md): the split-precision path puts f16 matrix instructions next to packed fp32 vector work, a
    combination that was seen to glitch in one kernel at a rate of 1e-4 per tile iteration.  The two-launch force calls of configs[1] / configs[2]
    (256 aspirin frames: 1 792 pair tiles x 3 interactions per call) are checked over forty calls: PaiNN has a fixed summation order between
    positions and forces -- every bit must agree; the SchNet kernels hand pair tiles to whichever wave is free, so their sums differ in the last
    bit (1.4e-7 of the largest force, with the fp32 matrix path just the same: scripts/r06_mol_repro.py) -- bound 1e-6.  A glitch was 1e-3 .. 1e-1."""
    from schnetpack_amd import model as M
    b = S.molecule_batch("aspirin", 256, seed=11)
    rep = (O.init_schnet_params if kind == "schnet" else O.init_painn_params)()
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model(kind)
    M.load_reference_params(m, rep, head)
    m = m.to(dev).eval()
    inp = M.batch_to_inputs(b, dev)
    f0 = m(dict(inp))["forces"].detach().clone()
    scale = float(f0.abs().max())
    for _ in range(40):
        f = m(dict(inp))["forces"].detach()
        if kind == "painn":
            assert torch.equal(f, f0)
        else:
            assert float((f - f0).abs().max()) < 1e-6 * scale
