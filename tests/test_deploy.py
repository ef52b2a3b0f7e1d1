"""Deployment runtime: flat weight file + spk_potential_* (include/spk_hip.h).

CPU tests: file layout written by schnetpack_amd.deploy, loader argument errors (no device involved).
GPU tests: the torch-free runtime against the reference's deployed models (tests/golden/deploy_painn.npz, made by
oracle/make_golden.py from spkdeploy-processed reference pickles fed LAMMPS-style inputs) and against the
package's own torch-side force call."""
import struct

import numpy as np
import pytest
import torch

from conftest import load_npz
from schnetpack_amd import deploy, model as M, synthetic as S
from schnetpack_amd._lib import SpkHipError


class _Offsets(torch.nn.Module):
    """Stand-in with the attributes of the reference's AddOffsets (transform/atomistic.py:217-283)."""

    def __init__(self, mean=0.0, atomref=None, is_extensive=True):
        super().__init__()
        self._property = "energy"
        self.add_mean = mean != 0.0
        self.add_atomrefs = atomref is not None
        self.is_extensive = is_extensive
        self.register_buffer("mean", torch.tensor(float(mean)))
        self.register_buffer("atomref", atomref if atomref is not None else torch.zeros(100))


_Offsets.__name__ = "AddOffsets"


def _painn_from_golden(prefix_rep, prefix_head, g, n_interactions, cutoff, mean):
    m = M.build_model("painn", 128, n_interactions, 20, cutoff)
    rep = {k[len(prefix_rep):]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(prefix_rep)}
    head = {k[len(prefix_head):]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(prefix_head)}
    M.load_reference_params(m, rep, head)
    m.postprocessors = torch.nn.ModuleList([_Offsets(mean)])
    return m.eval()


def _parse(blob):
    assert blob[:8] == deploy.MAGIC
    ints = struct.unpack("<16i", blob[8:72])
    flts = struct.unpack("<4f", blob[72:88])
    n_t = ints[12]
    table = {}
    for k in range(n_t):
        o = 88 + 48 * k
        name = blob[o:o + 32].rstrip(b"\0").decode()
        n, off = struct.unpack("<qq", blob[o + 32:o + 48])
        table[name] = (n, off)
    data0 = (88 + 48 * n_t + 63) // 64 * 64
    return ints, flts, table, data0


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_export_layout_round_trips_every_weight(kind):
    torch.manual_seed(3)
    m = M.build_model(kind, 128, 2, 20, 4.5).eval()
    ar = torch.linspace(-1, 1, 100)
    m.postprocessors = torch.nn.ModuleList([_Offsets(-2.5, ar)])
    blob = deploy.export_potential(m)
    ints, flts, table, data0 = _parse(blob)
    assert ints[:5] == (1, 0 if kind == "schnet" else 1, 128, 128, 2) and ints[5] == 20 and ints[7] == 64 and ints[9] == 100
    assert ints[10] == 1 and ints[11] == 100
    assert flts[0] == pytest.approx(4.5) and flts[2] == pytest.approx(-2.5)

    def tensor(name):
        n, off = table[name]
        assert off % 16 == 0
        return np.frombuffer(blob, "<f4", n, data0 + 4 * off)

    rep = m.representation
    assert np.array_equal(tensor("embedding"), rep.embedding.weight.detach().numpy().ravel())
    assert np.array_equal(tensor("atomref"), ar.numpy())
    assert np.array_equal(tensor("head_w1"), m.output_modules[0].outnet[0].weight.detach().numpy().ravel())
    if kind == "schnet":
        assert np.array_equal(tensor("l1.fn_w2"), rep.interactions[1].filter_network[1].weight.detach().numpy().ravel())
        assert np.array_equal(tensor("rbf_p0"), rep.radial_basis.offsets.numpy())
    else:
        assert np.array_equal(tensor("filt_w"), rep.filter_net.weight.detach().numpy().ravel())
        assert np.array_equal(tensor("l0.mix_w"), rep.mixing[0].mu_channel_mix.weight.detach().numpy().ravel())
    n_param = sum(p.numel() for p in rep.parameters()) + sum(p.numel() for p in m.output_modules[0].parameters())
    assert sum(n for n, _ in table.values()) == n_param + 40 + 100     # + rbf_p0/p1 + atomref
    # shared filters are expanded per interaction
    if kind == "painn":
        ms = M.build_model("painn", 128, 3, 20, 5.0, shared_filters=True).eval()
        _, _, t2, _ = _parse(deploy.export_potential(ms))
        assert t2["filt_w"][0] == 3 * 384 * 20


def test_export_refuses_what_the_runtime_cannot_run():
    m = M.build_model("schnet", 128, 1).eval()
    m.postprocessors = torch.nn.ModuleList([torch.nn.Identity()])
    with pytest.raises(ValueError, match="postprocessor"):
        deploy.export_potential(m)
    m = M.build_model("schnet", 128, 1).eval()
    m.output_modules = torch.nn.ModuleList([m.output_modules[0]])
    with pytest.raises(ValueError, match="Forces"):
        deploy.export_potential(m)
    m = M.build_model("painn", 128, 1, activation=torch.tanh).eval()
    with pytest.raises(ValueError, match="fused"):
        deploy.export_potential(m)


def test_loader_rejects_bad_files_before_touching_the_device():
    blob = deploy.export_potential(M.build_model("schnet", 128, 1).eval())
    with pytest.raises(SpkHipError, match="bad magic"):
        deploy.DeployedPotential(b"NOTSPK00" + blob[8:])
    with pytest.raises(SpkHipError, match="too short"):
        deploy.DeployedPotential(blob[:40])
    with pytest.raises(SpkHipError, match="truncated|outside the file"):
        deploy.DeployedPotential(blob[:4000])
    bad = bytearray(blob)
    bad[8:12] = struct.pack("<i", 7)
    with pytest.raises(SpkHipError, match="version"):
        deploy.DeployedPotential(bytes(bad))
    bad = bytearray(blob)
    bad[8 + 4 * 7:8 + 4 * 8] = struct.pack("<i", 48)       # head hidden width the file's tensors do not have
    with pytest.raises(SpkHipError, match="head_w1|fused"):
        deploy.DeployedPotential(bytes(bad))
    with pytest.raises(SpkHipError, match="cannot open"):
        deploy.DeployedPotential("/nonexistent/model.spkm")
    if not torch.cuda.is_available():
        with pytest.raises(SpkHipError):                    # a valid file still needs the device: no CPU fallback
            deploy.DeployedPotential(blob)


# ------------------------------------------------------------------------------------------- GPU
def _deploy_cases():
    g = load_npz("deploy_painn.npz")
    gp = load_npz("painn_aspirin_pretrained.npz")
    out = []
    for name in ("aspirin", "ethanol"):
        if name == "aspirin":
            m = _painn_from_golden("w_rep.", "w_head.", gp, int(g["aspirin_n_interactions"]), float(g["aspirin_cutoff"]), float(g["aspirin_mean"]))
        else:
            m = _painn_from_golden("ethanol_w_rep.", "ethanol_w_head.", g, int(g["ethanol_n_interactions"]), float(g["ethanol_cutoff"]),
                                   float(g["ethanol_mean"]))
        out.append((name, m, g))
    return out


def _close(E, F, g, t):
    Er, Fr = np.asarray(g[t + "energy"]), np.asarray(g[t + "forces"])
    # energies carry the -4e5 offset in fp32: 1 ulp = 0.03; forces are the sensitive quantity
    assert abs(float(E[0]) - float(Er.reshape(-1)[0])) <= 0.07, (t, E, Er)
    assert np.abs(F - Fr).max() / np.abs(Fr).max() < 1e-5, t


@pytest.mark.gpu
def test_runtime_matches_reference_deployed_models_lammps_style(tmp_path):
    """Unsorted edge order + image-shift offsets exactly as pair_schnetpack.cpp hands them over; outputs vs the
    spkdeploy-processed reference model (scripted for the aspirin model)."""
    for name, m, g in _deploy_cases():
        path = tmp_path / (name + ".spkm")
        deploy.export_potential(m, str(path))
        pot = deploy.DeployedPotential(str(path))
        assert pot.cutoff == pytest.approx(float(g[name + "_cutoff"]))
        assert pot.info["kind"] == 1 and pot.info["n_interactions"] == int(g[name + "_n_interactions"])
        for tag in ("free", "pbc"):
            t = "%s_%s_" % (name, tag)
            E, F = pot.compute(g[t + "Z"], g[t + "R"], g[t + "idx_i"], g[t + "idx_j"], g[t + "offsets"])
            _close(E, F, g, t)
            # the runtime's own device neighbour list gives the same answer, with and without a skin
            for skin in (0.0, 1.0):
                E2, F2 = pot.compute_cell(g[t + "Z"], g[t + "R"], g[t + "cell"][None], g[t + "pbc"][None], skin=skin)
                _close(E2, F2, g, t)
                assert pot.last_stats["rebuilt"]
                if skin == 0.0:
                    assert pot.last_stats["pairs"] == g[t + "idx_i"].shape[0]
        pot.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_runtime_matches_torch_side_force_call_on_a_batch(kind):
    """Batch of molecules (idx_m, n_mol > 1), seeded weights, atomref + mean offsets; torch-side HIP path as the
    second opinion (itself pinned to the reference by test_gpu_models.py)."""
    torch.manual_seed(5)
    m = M.build_model(kind, 128, 3).eval()
    ar = torch.randn(100)
    m.postprocessors = torch.nn.ModuleList([_Offsets(0.75, ar)])
    pot = deploy.DeployedPotential(deploy.export_potential(m))
    b = S.molecule_batch("aspirin", 6, seed=9)
    dev = torch.device("cuda:0")
    md = m.to(dev)
    out = md(M.batch_to_inputs(b, dev))
    n_at = torch.bincount(b["idx_m"], minlength=6).float()
    E_ref = out["energy"].detach().cpu() + 0.75 * n_at + torch.zeros(6).index_add_(0, b["idx_m"], ar[b["Z"]])
    F_ref = out["forces"].detach().cpu().numpy()
    g = torch.Generator().manual_seed(1)
    perm = torch.randperm(b["idx_i"].shape[0], generator=g)
    for order in (slice(None), perm):
        E, F = pot.compute(b["Z"].numpy(), b["R"].numpy(), b["idx_i"][order].numpy(), b["idx_j"][order].numpy(),
                           b["offsets"][order].numpy(), b["idx_m"].numpy(), 6)
        assert np.abs(E - E_ref.numpy()).max() / np.abs(E_ref.numpy()).max() < 1e-5
        assert np.abs(F - F_ref).max() / np.abs(F_ref).max() < 1e-5
    E, F = pot.compute_cell(b["Z"].numpy(), b["R"].numpy(), idx_m=b["idx_m"].numpy(), n_mol=6)
    assert np.abs(F - F_ref).max() / np.abs(F_ref).max() < 1e-5
    assert pot.last_stats["pairs"] == b["idx_i"].shape[0]


@pytest.mark.gpu
def test_runtime_skin_list_is_kept_until_an_atom_moves_half_the_skin():
    torch.manual_seed(2)
    m = M.build_model("schnet", 128, 2).eval()
    pot = deploy.DeployedPotential(deploy.export_potential(m))
    wb = S.water_box(n_side=4, seed=1)
    Z, R = wb["Z"].numpy(), wb["R"].numpy().astype(np.float32)
    cell, pbc = wb["cell"].numpy().reshape(1, 3, 3), np.ones((1, 3), np.uint8)
    E0, F0 = pot.compute_cell(Z, R, cell, pbc, skin=0.0)
    E1, F1 = pot.compute_cell(Z, R, cell, pbc, skin=0.6)
    assert pot.last_stats["rebuilt"] and np.abs(F1 - F0).max() / np.abs(F0).max() < 1e-5
    n_skin = pot.last_stats["pairs"]
    rng = np.random.RandomState(0)
    R2 = R + rng.uniform(-0.1, 0.1, R.shape).astype(np.float32)          # |dR| < 0.18 < skin / 2
    E2, F2 = pot.compute_cell(Z, R2, cell, pbc, skin=0.6)
    assert not pot.last_stats["rebuilt"] and pot.last_stats["pairs"] == n_skin
    E2x, F2x = pot.compute_cell(Z, R2, cell, pbc, skin=0.0)              # exact list of the moved geometry
    assert abs(float(E2[0] - E2x[0])) / abs(float(E2x[0])) < 1e-5
    assert np.abs(F2 - F2x).max() / np.abs(F2x).max() < 1e-5
    R3 = R2.copy()
    R3[7] += np.array([0.25, 0.2, 0.0], np.float32)                      # beyond skin / 2 of the list's reference
    pot.compute_cell(Z, R2, cell, pbc, skin=0.6)
    pot.compute_cell(Z, R3, cell, pbc, skin=0.6)
    assert pot.last_stats["rebuilt"]


@pytest.mark.gpu
def test_runtime_argument_errors():
    m = M.build_model("schnet", 128, 1).eval()
    pot = deploy.DeployedPotential(deploy.export_potential(m))
    Z = np.array([6, 1, 1], np.int64)
    R = np.array([[0, 0, 0], [1.0, 0, 0], [0, 1.0, 0]], np.float32)
    ii, jj = np.array([0, 1, 0, 2], np.int64), np.array([1, 0, 2, 0], np.int64)
    E, F = pot.compute(Z, R, ii, jj)
    assert np.isfinite(E).all() and np.abs(F.sum(0)).max() < 1e-4
    with pytest.raises(SpkHipError, match="out of range"):
        pot.compute(Z, R, ii, np.array([1, 0, 3, 0], np.int64))
    with pytest.raises(SpkHipError, match="atomic number"):
        pot.compute(np.array([6, 1, 100], np.int64), R, ii, jj)
    with pytest.raises(SpkHipError, match="idx_m"):
        pot.compute(Z, R, ii, jj, idx_m=np.array([0, 1, 0], np.int64), n_mol=2)
    with pytest.raises(SpkHipError, match="cell"):
        pot.compute_cell(Z, R, None, np.ones((1, 3), np.uint8))
    E0, F0 = pot.compute(Z, R, np.zeros(0, np.int64), np.zeros(0, np.int64))     # no neighbours at all
    assert np.abs(F0).max() == 0.0


@pytest.mark.gpu
def test_plain_c_program_links_the_library_alone(tmp_path):
    """examples/native/spk_run.c (gcc, no Python / torch in the process) loads the file and reproduces the
    ctypes-driven runtime; what a LAMMPS pair style would do."""
    import os
    import subprocess
    from schnetpack_amd.csrc import build as B
    exe = B.RUN_BIN
    assert os.path.exists(exe), "build() compiles examples/native/spk_run.c"
    torch.manual_seed(4)
    m = M.build_model("painn", 128, 2).eval()
    m.postprocessors = torch.nn.ModuleList([_Offsets(-3.0)])
    path = tmp_path / "m.spkm"
    deploy.export_potential(m, str(path))
    wb = S.water_box(n_side=3, seed=2, cutoff=4.0)     # L >= 2 cutoff for the host generator; the model cutoff is 5
    Z, R = wb["Z"].numpy(), wb["R"].numpy().astype(np.float32)
    n = Z.shape[0]
    cell, pbc = wb["cell"].numpy().astype(np.float32).reshape(1, 3, 3), np.ones((1, 3), np.uint8)
    with open(tmp_path / "sys.bin", "wb") as f:
        f.write(np.array([n, 1, 1], np.int64).tobytes() + Z.astype(np.int64).tobytes() + np.zeros(n, np.int64).tobytes()
                + R.tobytes() + cell.tobytes() + pbc.tobytes())
    res = subprocess.run([exe, str(path), str(tmp_path / "sys.bin"), "3"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    E, F = None, np.zeros((n, 3), np.float32)
    for line in res.stdout.splitlines():
        w = line.split()
        if w[0] == "E":
            E = np.float32(w[2])
        elif w[0] == "F":
            F[int(w[1])] = [np.float32(x) for x in w[2:5]]
    pot = deploy.DeployedPotential(str(path))
    E2, F2 = pot.compute_cell(Z, R, cell, pbc)
    # same kernels, same inputs: equal up to the order of the few float atomics of the molecule sum
    assert abs(float(E) - float(E2[0])) <= 2e-6 * abs(float(E2[0])) and np.abs(F - F2).max() <= 2e-6 * np.abs(F2).max()
    assert "PaiNN" in res.stderr
