"""Molecule-resident SchNet kernels (schnetpack_amd/csrc/spk_schnet_mol.hip): batches of small molecules are block diagonal
(data/loader.py:35-46), so a group of <= 32 atoms runs all interactions inside one workgroup.  Checked against the CPU oracle
(representation/schnet.py:147-173) and against the general driver (SPK_VARIANT_MFMA_DIRECTED never takes the molecule path).
Tolerance 1e-5 relative (north_star)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import spk_oracle as O
from schnetpack_amd import synthetic as S

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _mixed_batch(seed, sizes):
    """Molecules of different kinds in one batch: aspirin (21 atoms), ethanol (9), single atoms, pairs."""
    rng = np.random.RandomState(seed)
    systems = []
    for kind in sizes:
        if kind == "aspirin":
            Z, R = S.ASPIRIN_Z, np.asarray(S.ASPIRIN_R) + 0.05 * rng.randn(21, 3)
        elif kind == "ethanol":
            Z, R = S.ETHANOL_Z, np.asarray(S.ETHANOL_R) + 0.05 * rng.randn(9, 3)
        elif kind == "atom":
            Z, R = [8], rng.randn(1, 3)
        else:  # "dimer"
            Z, R = [1, 1], np.array([[0.0, 0.0, 0.0], [0.74 + 0.05 * rng.randn(), 0.0, 0.0]])
        ii, jj = S.neighbor_pairs_open(np.asarray(R), 5.0)
        systems.append({"Z": Z, "R": R, "idx_i": ii, "idx_j": jj})
    return S.collate(systems)


def _run(batch, dev, n_int=3, n_rbf=20, radial="gaussian", directed=False):
    from schnetpack_amd import _lib, model as M
    rep = O.init_schnet_params(128, n_int, n_rbf, 5.0, radial=radial)
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model("schnet", 128, n_int, n_rbf, 5.0, radial)
    M.load_reference_params(m, rep, head)
    m = m.to(dev).eval()
    _lib.set_variant(_lib.VARIANT_MFMA_DIRECTED if directed else _lib.VARIANT_AUTO)
    try:
        _lib.profile_enable(True)
        _lib.profile_report()
        inp = M.batch_to_inputs(batch, dev)
        out = m(inp)
        res = (out["energy"].detach().cpu(), out["forces"].detach().cpu(), inp["scalar_representation"].detach().cpu())
        tags = _lib.profile_report()
    finally:
        _lib.profile_enable(False)
        _lib.set_variant(_lib.VARIANT_AUTO)
    return res, tags, (rep, head)


@pytest.mark.parametrize("sizes,n_int,n_rbf,radial", [
    (["aspirin"] * 7, 3, 20, "gaussian"),
    (["ethanol", "aspirin", "atom", "ethanol", "dimer", "ethanol", "ethanol", "aspirin", "atom", "atom"], 3, 20, "gaussian"),
    (["ethanol"] * 40, 2, 16, "bessel"),
    (["aspirin", "dimer"] * 3, 1, 8, "gaussian"),
    (["aspirin"] * 300, 3, 20, "gaussian"),          # more groups than compute units: the workgroups loop
    (["aspirin"] * 5, 4, 32, "gaussian"),
])
def test_molecule_resident_forward_matches_oracle(dev, sizes, n_int, n_rbf, radial):
    b = _mixed_batch(3, sizes)
    (e, f, x), tags, (rep, head) = _run(b, dev, n_int, n_rbf, radial)
    assert "schnet_mol_fwd" in tags and not any(t.startswith("cfconv_fwd") for t in tags), tags      # the path under test ran
    assert ("schnet_mol_bwd" in tags) == (n_rbf <= 24), tags
    ref = O.energy_and_forces("schnet", rep, head, b, n_int, need_rep=True)
    assert rel_err(x, ref["scalar_representation"]) < TOL
    assert rel_err(e, ref["energy"]) < TOL and rel_err(f, ref["forces"]) < TOL
    # the general driver on the same batch
    (e2, f2, x2), tags2, _ = _run(b, dev, n_int, n_rbf, radial, directed=True)
    assert "schnet_mol_fwd" not in tags2
    assert rel_err(x, x2) < 2e-6 and rel_err(f, f2) < 5e-6


def test_molecule_resident_forward_is_deterministic(dev):
    """Row sums instead of float atomics: the representation is bit-reproducible (the reference's index_add on the CPU is)."""
    b = S.molecule_batch("aspirin", 64, seed=9)
    (e1, f1, x1), tags, _ = _run(b, dev)
    (e2, f2, x2), _, _ = _run(b, dev)
    assert "schnet_mol_fwd" in tags
    assert torch.equal(x1, x2)        # (the energy head sums molecules with one float atomic per block: not compared bitwise)


def test_large_molecules_fall_back_to_the_general_driver(dev):
    """A block of more than 32 atoms (two aspirin molecules bonded into one 42-atom system) is not eligible."""
    rng = np.random.RandomState(0)
    R = np.concatenate([np.asarray(S.ASPIRIN_R), np.asarray(S.ASPIRIN_R) + np.array([4.0, 0.0, 0.0])]) + 0.05 * rng.randn(42, 3)
    ii, jj = S.neighbor_pairs_open(R, 5.0)
    b = S.collate([{"Z": S.ASPIRIN_Z * 2, "R": R, "idx_i": ii, "idx_j": jj}] * 3)
    (e, f, x), tags, (rep, head) = _run(b, dev)
    assert "schnet_mol_fwd" not in tags
    ref = O.energy_and_forces("schnet", rep, head, b, 3)
    assert rel_err(f, ref["forces"]) < TOL


@pytest.mark.parametrize("sizes,agg", [(["aspirin"] * 9, "sum"), (["ethanol", "aspirin", "atom", "dimer", "ethanol", "atom", "atom", "ethanol"], "sum"),
                                        (["ethanol"] * 13, "avg")])
def test_standard_potential_in_two_launches_equals_the_module_by_module_path(dev, sizes, agg):
    """NeuralNetworkPotential([PairwiseDistances], SchNet, [Atomwise, Forces]) in eval mode is ONE operator (spk_hip::schnet_potential):
    pair vectors + representation + energy head in one launch, head + representation + dE/dR in the backward launch.  Same
    energies / forces / representation as the separate modules, and as the oracle."""
    from schnetpack_amd import _lib, model as M
    b = _mixed_batch(5, sizes)
    rep, head = O.init_schnet_params(128, 3, 20, 5.0), O.init_atomwise_params(128, seed=1)
    m = M.build_model("schnet", 128, 3, 20, 5.0)
    M.load_reference_params(m, rep, head)
    m.output_modules[0].aggregation_mode = agg
    m = m.to(dev).eval()
    assert m._potential
    res = {}
    for fused in (True, False):
        # (the flags are fixed at construction; the aggregation mode was changed after it)
        m._potential = fused
        m._potential_forces = fused and agg == "sum"       # energies AND forces straight from the two launches
        _lib.profile_enable(True)
        _lib.profile_report()
        try:
            inp = M.batch_to_inputs(b, dev)
            inp["_n_atoms"] = torch.bincount(b["idx_m"], minlength=int(b["n_mol"])).to(dev)
            out = m(inp)
            res[fused] = (out["energy"].detach().cpu(), out["forces"].detach().cpu(), inp["scalar_representation"].detach().cpu(), _lib.profile_report())
        finally:
            _lib.profile_enable(False)
    m._potential, m._potential_forces = True, agg == "sum"
    tags = res[True][3]
    assert "schnet_mol_fwd" in tags and "schnet_mol_bwd" in tags and "atomwise_fwd" not in tags and "atomwise_bwd" not in tags, tags
    assert not any(t.startswith("pairwise") for t in tags), tags
    assert "atomwise_fwd" in res[False][3]
    if agg == "sum":       # the forces route and the autograd route of the fused operator (same kernels; the per-pair sums of the
        # backward meet in LDS in task order, so two runs agree to rounding, not bit for bit)
        m._potential_forces = False
        inp = M.batch_to_inputs(b, dev)
        out = m(inp)
        m._potential_forces = True
        assert rel_err(out["forces"].detach().cpu(), res[True][1]) < 2e-6 and rel_err(out["energy"].detach().cpu(), res[True][0]) < 1e-6
    for a, c in zip(res[True][:3], res[False][:3]):
        assert rel_err(a, c) < 2e-6
    ref = O.energy_and_forces("schnet", rep, head, b, 3)
    e_ref = ref["energy"] / torch.bincount(b["idx_m"]).double() if agg == "avg" else ref["energy"]
    f_ref = ref["forces"] if agg == "sum" else None
    assert rel_err(res[True][0], e_ref) < TOL
    if f_ref is not None:
        assert rel_err(res[True][1], f_ref) < TOL


def test_standard_potential_gradient_reaches_the_embedding_rows(dev):
    """dE/dx0 (embedding rows as a leaf) and dE/dR from one backward of the fused operator against the separate operators."""
    from schnetpack_amd import model as M
    b = _mixed_batch(7, ["aspirin", "ethanol", "ethanol", "dimer"])
    m = M.build_model("schnet").to(dev).eval()
    inp = M.batch_to_inputs(b, dev)
    rep, head = m.representation, m.output_modules[0]
    kind, p0, p1 = rep.radial_basis.kernel_params()
    l0, l1 = head.outnet[0], head.outnet[1]
    got = []
    for fused in (True, False):
        x0 = rep.embed(inp).detach().requires_grad_(True)
        R = inp["_positions"].detach().clone().requires_grad_(True)
        if fused:
            E, x = torch.ops.spk_hip.schnet_potential(x0, R, inp["_offsets"], inp["_idx_i"], inp["_idx_j"], inp["_idx_m"], int(b["n_mol"]),
                                                      rep.interaction_weights(), [l0.weight, l0.bias, l1.weight, l1.bias], rep.n_filters, kind, p0, p1,
                                                      rep.cutoff_fn.cutoff_value(), head._head_act)
        else:
            r = torch.ops.spk_hip.pairwise(R, inp["_idx_i"], inp["_idx_j"], inp["_offsets"])
            x = torch.ops.spk_hip.schnet(x0, r, inp["_idx_i"], inp["_idx_j"], rep.interaction_weights(), rep.n_filters, kind, p0, p1, rep.cutoff_fn.cutoff_value())
            E = torch.ops.spk_hip.atomwise(x, l0.weight, l0.bias, l1.weight, l1.bias, inp["_idx_m"], int(b["n_mol"]), head._head_act)[0]
        w = torch.linspace(0.5, 1.5, E.shape[0], device=dev)
        gx0, gR = torch.autograd.grad((E * w).sum() + 0.01 * (x ** 2).sum(), [x0, R])
        got.append((E.detach().cpu(), gx0.cpu(), gR.cpu()))
    for a, c in zip(*got):
        assert rel_err(a, c) < 3e-6


def test_groups_whose_pairs_all_lie_beyond_the_cutoff(dev):
    """A list with a skin can hold a molecule whose every pair is outside the model cutoff (here: two atoms 6 A apart in a 7 A
    list).  Such a group gets no pair tile at all: finite results, zero forces on its atoms, the energies of isolated atoms --
    and its neighbours in the batch are unaffected."""
    from schnetpack_amd import model as M
    far = {"Z": [8, 1], "R": np.array([[0.0, 0.0, 0.0], [6.0, 0.0, 0.0]]), "idx_i": np.array([0, 1]), "idx_j": np.array([1, 0])}
    rng = np.random.RandomState(0)
    asp = {"Z": S.ASPIRIN_Z, "R": np.asarray(S.ASPIRIN_R) + 0.05 * rng.randn(21, 3)}
    ii, jj = S.neighbor_pairs_open(asp["R"], 7.0)            # a skin list for the aspirin too
    asp.update(idx_i=ii, idx_j=jj)
    b = S.collate([asp, far, dict(asp), far])
    m = M.build_model("schnet").to(dev).eval()
    out = m(M.batch_to_inputs(b, dev))
    e, f = out["energy"].detach().cpu(), out["forces"].detach().cpu()
    assert torch.isfinite(e).all() and torch.isfinite(f).all()
    assert float(f[21:23].abs().max()) == 0.0 and float(f[44:46].abs().max()) == 0.0
    assert abs(float(e[1] - e[3])) < 1e-6 * abs(float(e[1])) and abs(float(e[0] - e[2])) < 2e-6 * abs(float(e[0]))
    # the same molecules with the exact 5 A lists
    ii5, jj5 = S.neighbor_pairs_open(asp["R"], 5.0)
    b5 = S.collate([dict(asp, idx_i=ii5, idx_j=jj5), {"Z": [8], "R": np.zeros((1, 3)), "idx_i": np.zeros(0, int), "idx_j": np.zeros(0, int)}])
    out5 = m(M.batch_to_inputs(b5, dev))
    assert rel_err(e[0:1], out5["energy"].detach().cpu()[0:1]) < 2e-6
    assert rel_err(f[:21], out5["forces"].detach().cpu()[:21]) < 5e-6


def test_energy_store_versus_accumulation_and_custom_embeddings(dev):
    """schnet_potential_forces: (i) molecules that each lie inside one group get their energies STORED (no clearing launch),
    a molecule that straddles two groups (two disconnected fragments under one idx_m) takes the accumulate route -- same numbers as
    the separate modules either way; (ii) the in-launch embedding lookup equals the module's own lookup (x0 path)."""
    from schnetpack_amd import model as M
    b = _mixed_batch(11, ["aspirin", "ethanol", "aspirin", "ethanol", "ethanol"])
    m = M.build_model("schnet").to(dev).eval()
    assert m._potential_forces

    def run(batch, modular):
        m._potential, m._potential_forces = (not modular), (not modular)
        try:
            out = m(M.batch_to_inputs(batch, dev))
        finally:
            m._potential, m._potential_forces = True, True
        return out["energy"].detach().cpu(), out["forces"].detach().cpu()

    e1, f1 = run(b, False)
    e0, f0 = run(b, True)
    assert rel_err(e1, e0) < 2e-6 and rel_err(f1, f0) < 5e-6
    # merge molecules 1 and 2 under one id (an "ethanol + aspirin" complex of two disconnected fragments: 30 atoms, two groups)
    b2 = dict(b)
    idx_m = b["idx_m"].clone()
    idx_m[idx_m >= 2] -= 1
    b2["idx_m"], b2["n_mol"] = idx_m, int(b["n_mol"]) - 1
    e1, f1 = run(b2, False)
    e0, f0 = run(b2, True)
    assert e1.shape[0] == 4 and rel_err(e1, e0) < 2e-6 and rel_err(f1, f0) < 5e-6
    # x0 route: an embedding module that is not a plain table
    class Shifted(torch.nn.Embedding):
        def forward(self, z):
            return super().forward(z) + 0.0
    emb = Shifted(100, 128).to(dev)
    emb.load_state_dict(m.representation.embedding.state_dict())
    plain = m.representation.embedding
    m.representation.embedding = emb
    try:
        e2, f2 = run(b, False)
    finally:
        m.representation.embedding = plain
    e1, f1 = run(b, False)
    assert rel_err(e2, e1) < 1e-6 and rel_err(f2, f1) < 2e-6
