"""GPU: the TorchScript-able boundary on the device.

* scripted mirrors == eager mirrors == oracle (reference tests/nn/test_schnet.py:83-96 runs the scripted module);
* the `spkdeploy` archives built by oracle/build_ref.py (reference NeuralNetworkPotential + Atomwise + Forces + AddOffsets
  code scripted around the HIP classes) are loaded the way interfaces/lammps/pair_schnetpack.cpp:128 does -- only the two
  shared libraries, `torch.jit.load(..., device)` -- fed the LAMMPS-style input dict (:285-301: one system, permuted edges,
  image shifts as offsets) and compared with the reference-generated fixtures tests/golden/deploy_painn.npz;
* stress through the reference's Strain + Forces(calc_stress=True) (atomistic/response.py:77-90, 434-464);
* the autograd contract of the eval operators on real tensors.
Tolerance 1e-5 relative (north_star)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_npz, rel_err
from oracle import build_ref, refshim, spk_oracle as O
from schnetpack_amd import model as M, synthetic as S

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a ROCm device"
    return torch.device("cuda:0")


def _model(kind, dev, **kw):
    rep = O.init_schnet_params() if kind == "schnet" else O.init_painn_params()
    head = O.init_atomwise_params(128, seed=1)
    m = M.build_model(kind, **kw)
    M.load_reference_params(m, rep, head)
    return m.to(dev).eval(), rep, head


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_scripted_model_equals_eager_and_oracle(dev, kind, tmp_path):
    b = S.molecule_batch("aspirin", 5, seed=17)
    m, rep, head = _model(kind, dev)
    ref = O.energy_and_forces(kind, rep, head, b, 3)
    sm = torch.jit.script(m)
    assert ("spk_hip::schnet" if kind == "schnet" else "spk_hip::painn") in str(sm.representation.graph)
    inp = M.batch_to_inputs(b, dev)
    inp["_n_molecules"] = torch.tensor(int(b["n_mol"]))      # scripted dicts hold tensors only
    out_e = m(dict(inp))
    for _ in range(3):                                       # incl. the profiling / optimised executor passes
        out_s = sm(dict(inp))
    assert rel_err(out_s["forces"].cpu(), ref["forces"]) < TOL and rel_err(out_s["energy"].cpu(), ref["energy"]) < TOL
    # (SchNet's pair kernels sum with float atomics: the order, hence the last bits, vary from call to call)
    assert rel_err(out_s["energy"].cpu(), out_e["energy"].cpu()) < 2e-6 and rel_err(out_s["forces"].cpu(), out_e["forces"].cpu()) < 2e-6
    # save -> load -> run; without the host-side molecule count (the reference's int(idx_m[-1]) + 1 path)
    p = str(tmp_path / "m.pt")
    torch.jit.save(sm, p)
    lm = torch.jit.load(p, map_location=dev)
    inp.pop("_n_molecules")
    out_l = lm(dict(inp))
    assert rel_err(out_l["forces"].cpu(), ref["forces"]) < TOL
    # training mode scripts and runs too (primitive path, create_graph)
    st = torch.jit.script(m.train())
    out_t = st(dict(inp))
    assert rel_err(out_t["forces"].detach().cpu(), ref["forces"]) < TOL
    assert out_t["forces"].requires_grad


def test_spkdeploy_archives_load_like_lammps_and_match_reference_fixtures(dev):
    paths = {n: build_ref.deployed_path(n) for n in build_ref.DEPLOYED}
    if not all(os.path.exists(p) for p in paths.values()):
        pytest.skip("oracle/_ref/deployed/*.pt not built (needs /root/reference at build time)")
    g = load_npz("deploy_painn.npz")
    for name, p in paths.items():
        extra = {"cutoff": ""}
        jm = torch.jit.load(p, map_location=dev, _extra_files=extra)       # pair_schnetpack.cpp:125-131
        assert float(extra["cutoff"]) == pytest.approx(float(g[name + "_cutoff"]))
        for tag in ("free", "pbc"):
            t = "%s_%s_" % (name, tag)
            n = int(g[t + "Z"].shape[0])
            inp = {"_positions": torch.from_numpy(g[t + "R"]).to(dev), "_idx_i": torch.from_numpy(g[t + "idx_i"]).to(dev),
                   "_idx_j": torch.from_numpy(g[t + "idx_j"]).to(dev), "_idx_m": torch.zeros(n, dtype=torch.long, device=dev),
                   "_offsets": torch.from_numpy(g[t + "offsets"]).to(dev), "_cell": torch.from_numpy(g[t + "cell"]).to(dev),
                   "_n_atoms": torch.tensor([n], device=dev), "_atomic_numbers": torch.from_numpy(g[t + "Z"]).to(dev)}
            out = jm(inp)
            Er, Fr = np.asarray(g[t + "energy"]).reshape(-1), np.asarray(g[t + "forces"])
            F = out["forces"].detach().cpu().numpy()
            assert np.abs(F - Fr).max() / np.abs(Fr).max() < TOL, t
            # energies carry the -4e5 kcal/mol offset in fp32: 1 ulp = 0.03
            assert abs(float(out["energy"].detach().cpu().reshape(-1)[0]) - float(Er[0])) <= 0.07, t


@pytest.mark.skipif(not refshim.available(), reason="neither /root/reference nor oracle/_ref present")
@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_stress_through_reference_strain_module(dev, kind):
    """Strain (atomistic/response.py:434-464) makes positions, cell and OFFSETS functions of a strain tensor; Forces(calc_stress=True)
    differentiates the energy w.r.t. it: exercises d r_ij / d offsets of the HIP PairwiseDistances."""
    import schnetpack_amd.install as inst
    ns = refshim.load()
    sys.modules["ase.data"].atomic_masses = np.ones(119)
    spk = sys.modules["schnetpack"]
    b = S.water_box(n_side=4, seed=2)           # 192 atoms, periodic, L = 12.4 A

    def inputs(device):
        return {"_atomic_numbers": b["Z"].to(device), "_positions": b["R"].clone().to(device), "_idx_i": b["idx_i"].to(device),
                "_idx_j": b["idx_j"].to(device), "_offsets": b["offsets"].clone().to(device), "_idx_m": b["idx_m"].to(device),
                "_cell": b["cell"].reshape(1, 3, 3).clone().to(device), "_pbc": torch.ones(3, dtype=torch.bool, device=device),
                "_n_atoms": torch.tensor([b["Z"].shape[0]], device=device)}

    def build():
        torch.manual_seed(0)
        rb, cf = spk.nn.GaussianRBF(20, 5.0), spk.nn.CosineCutoff(5.0)
        rep_cls = sys.modules["schnetpack.representation." + kind].__dict__["SchNet" if kind == "schnet" else "PaiNN"]
        aw = sys.modules["schnetpack.atomistic.atomwise"].Atomwise(n_in=128, output_key="energy")
        pd = sys.modules["schnetpack.atomistic.distances"].PairwiseDistances()
        return ns.model.NeuralNetworkPotential(rep_cls(128, 3, rb, cf), input_modules=[ns.response.Strain(), pd],
                                               output_modules=[aw, ns.response.Forces(calc_forces=True, calc_stress=True)])
    try:
        out_ref = build().double().eval()({k: (v.double() if v.is_floating_point() else v) for k, v in inputs("cpu").items()})
        inst.install(spk)
        m = build().to(dev).eval()
        out = m(inputs(dev))
    finally:
        inst.uninstall()
    assert out["stress"].shape == (1, 3, 3)
    assert rel_err(out["stress"].cpu(), out_ref["stress"]) < TOL
    assert rel_err(out["forces"].cpu(), out_ref["forces"]) < TOL


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_eval_operator_autograd_contract_on_device(dev, kind):
    b = S.molecule_batch("aspirin", 2, seed=4)
    m, rep, head = _model(kind, dev)
    out = m(M.batch_to_inputs(b, dev))
    with pytest.raises(RuntimeError, match="training mode"):
        out["energy"].sum().backward()                       # parameter gradients in eval mode: loud, not silently absent
    # embedding rows ARE differentiated by the fused operator (dL/dx0): a frozen-weight model can still train its embedding
    for p in m.parameters():
        p.requires_grad_(False)
    m.representation.embedding.weight.requires_grad_(True)
    inp = M.batch_to_inputs(b, dev)
    x = m.representation(m.input_modules[0](inp))["scalar_representation"]
    w = torch.randn(x.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    (x * w).sum().backward()
    g_emb = m.representation.embedding.weight.grad.cpu()
    # oracle gradient w.r.t. the embedding table
    rp = dict(rep)
    rp["embedding.weight"] = rep["embedding.weight"].clone().double().requires_grad_(True)
    rp = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() and k != "embedding.weight" else v) for k, v in rp.items()}
    r_ij = O.pairwise_vectors(b["R"].double(), b["idx_i"], b["idx_j"], b["offsets"].double())
    xo = O.schnet_representation(b["Z"], r_ij, b["idx_i"], b["idx_j"], rp, 3) if kind == "schnet" else \
        O.painn_representation(b["Z"], r_ij, b["idx_i"], b["idx_j"], rp, 3)[0]
    (go,) = torch.autograd.grad((xo * w.cpu().double()).sum(), [rp["embedding.weight"]])
    assert rel_err(g_emb, go) < TOL


def _write_system_bin(path, g, t):
    """LAMMPS-style single system (golden fixture `t`) in the little-endian layout examples/native/spk_jit_client.cpp reads."""
    n, E = int(g[t + "Z"].shape[0]), int(g[t + "idx_i"].shape[0])
    with open(path, "wb") as f:
        f.write(np.asarray([n, E], dtype="<i8").tobytes())
        f.write(np.asarray(g[t + "Z"], dtype="<i8").tobytes())
        f.write(np.asarray(g[t + "R"], dtype="<f4").tobytes())
        f.write(np.asarray(g[t + "idx_i"], dtype="<i8").tobytes())
        f.write(np.asarray(g[t + "idx_j"], dtype="<i8").tobytes())
        f.write(np.asarray(g[t + "offsets"], dtype="<f4").tobytes())
        f.write(np.asarray(g[t + "cell"], dtype="<f4").reshape(9).tobytes())


def test_cpp_libtorch_client_loads_the_archives_like_lammps(dev, tmp_path):
    """The actual contract of interfaces/lammps/pair_schnetpack.cpp:125-131, :328: a C++ process -- no Python, no `import
    schnetpack_amd` -- dlopens the two operator libraries, torch::jit::load(path, device, {"cutoff": ""}), forward on the dict of
    one system, and reproduces the reference-generated fixtures (free molecule and periodic cell, both shipped PaiNN models)."""
    import subprocess
    from schnetpack_amd.csrc import build as B
    exe = B.JIT_BIN
    if not os.path.exists(exe):
        pytest.skip("spk_jit_client not built")
    paths = {n: build_ref.deployed_path(n) for n in build_ref.DEPLOYED}
    if not all(os.path.exists(p) for p in paths.values()):
        pytest.skip("oracle/_ref/deployed/*.pt not built (needs /root/reference at build time)")
    g = load_npz("deploy_painn.npz")
    for name, p in paths.items():
        for tag in ("free", "pbc"):
            t = "%s_%s_" % (name, tag)
            sysf = str(tmp_path / (t + "system.bin"))
            _write_system_bin(sysf, g, t)
            r = subprocess.run([exe, p, sysf, B.LIB, B.TORCH_LIB, "cuda:0"], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-2000:]
            lines = r.stdout.strip().splitlines()
            assert lines[0].startswith("cutoff") and float(lines[0].split()[1]) == pytest.approx(float(g[name + "_cutoff"]))
            E = float(lines[1].split()[1])
            F = np.array([[float(x) for x in ln.split()] for ln in lines[2:]])
            Er, Fr = np.asarray(g[t + "energy"]).reshape(-1), np.asarray(g[t + "forces"])
            assert F.shape == Fr.shape
            assert np.abs(F - Fr).max() / np.abs(Fr).max() < TOL, t
            assert abs(E - float(Er[0])) <= 0.07, t
