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
