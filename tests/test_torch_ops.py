"""CPU: the TORCH_LIBRARY(spk_hip) boundary (schnetpack_amd/csrc/spk_torch.cpp) without a
device -- registration, the loud CPU refusal, Meta kernels (a whole force call runs on the ``meta`` device), autograd
contract of the fused eval operators, TorchScript of the module mirrors (reference tests/nn/test_schnet.py:83-96) and of
the reference's own NeuralNetworkPotential around them (src/scripts/spkdeploy:16-40)."""
import io
import os
import sys

import pytest
import torch

from schnetpack_amd import model as M, torchops
from schnetpack_amd.nn import BesselRBF, CosineCutoff, GaussianRBF
from schnetpack_amd.representation import PaiNN, SchNet


def _meta_inputs(N=12, E=40, n_mol=2, with_count=True):
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="meta")
    d = {"_atomic_numbers": z(N, dt=torch.long), "_positions": z(N, 3), "_idx_i": z(E, dt=torch.long), "_idx_j": z(E, dt=torch.long),
         "_offsets": z(E, 3), "_idx_m": z(N, dt=torch.long)}
    if with_count:
        d["_n_molecules"] = torch.tensor(n_mol)
    return d


def test_every_operator_is_registered_with_device_and_meta_kernels():
    for name in torchops.OPERATORS:
        assert hasattr(torch.ops.spk_hip, name), name
    has = torch._C._dispatch_has_kernel_for_dispatch_key
    for name in ["scatter_add", "gather", "pairwise", "dense", "radial_cutoff", "schnet", "painn", "atomwise",
                 "schnet_forward", "schnet_backward", "painn_forward", "painn_backward", "atomwise_forward", "atomwise_backward"]:
        assert has("spk_hip::" + name, "CUDA"), name          # "CUDA" = the ROCm device key of PyTorch-ROCm
        assert has("spk_hip::" + name, "Meta"), name
    for name in ["scatter_add", "gather", "pairwise", "dense", "radial_cutoff", "schnet", "painn", "atomwise"]:
        assert has("spk_hip::" + name, "AutogradCUDA") or has("spk_hip::" + name, "Autograd"), name
    s = str(torch.ops.spk_hip.schnet.default._schema)
    assert "Tensor[] weights" in s and "Tensor? rbf_p1" in s and s.endswith("-> Tensor")


def test_cpu_tensors_are_refused_loudly():
    x, idx = torch.randn(6, 4), torch.tensor([0, 0, 1, 1, 2, 2])
    for call in (lambda: torch.ops.spk_hip.scatter_add(x, idx, 3, 0), lambda: torch.ops.spk_hip.gather(x, idx, 0),
                 lambda: torch.ops.spk_hip.dense(x, torch.randn(8, 4), None, 1),
                 lambda: torch.ops.spk_hip.pairwise(torch.randn(3, 3), idx[:2], idx[2:4], None),
                 lambda: torch.ops.spk_hip.schnet_forward(x, torch.randn(6, 3), idx, idx, [x] * 9, 4, 0, x[0], None, 5.0, False)):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()


def test_meta_kernels_infer_shapes():
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="meta")
    idx = z(10, dt=torch.long)
    assert torch.ops.spk_hip.scatter_add(z(2, 10, 4), idx, 3, 1).shape == (2, 3, 4)
    assert torch.ops.spk_hip.gather(z(5, 3, 4), idx, 0).shape == (10, 3, 4)
    assert torch.ops.spk_hip.pairwise(z(5, 3), idx, idx, z(10, 3)).shape == (10, 3)
    assert torch.ops.spk_hip.dense(z(7, 2, 16), z(32, 16), z(32), 2).shape == (7, 2, 32)
    phi, fc = torch.ops.spk_hip.radial_cutoff(z(9), 0, z(20), z(20), 5.0, True, True)
    assert phi.shape == (9, 20) and fc.shape == (9,)
    q, mu = torch.ops.spk_hip.painn(z(5, 64), z(10, 3), idx, idx, [z(1)] * 11, False, 1e-8, 0, z(20), z(20), 5.0)
    assert q.shape == (5, 64) and mu.shape == (5, 3, 64)
    E, ya = torch.ops.spk_hip.atomwise(z(5, 64), z(32, 64), z(32), z(1, 32), z(1), z(5, dt=torch.long), 2, 2)
    assert E.shape == (2,) and ya.shape == (5, 1)


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_whole_force_call_on_the_meta_device(kind):
    """Eager and scripted, eval and training mode: control flow + shapes of the complete call without a device."""
    m = M.build_model(kind).to("meta")
    for mode in ("eval", "train"):
        getattr(m, mode)()
        out = m(_meta_inputs())
        assert out["energy"].shape == (2,) and out["forces"].shape == (12, 3)
        sm = torch.jit.script(m)
        out = sm(_meta_inputs())
        assert out["energy"].shape == (2,) and out["forces"].shape == (12, 3)
    # without the host-side molecule count the reference's int(idx_m[-1]) + 1 runs (atomistic/atomwise.py:80)
    code = torch.jit.script(m.output_modules[0]).code
    assert "_n_molecules" in code and "idx_m" in code


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_eval_mode_parameter_gradients_raise_and_training_mode_has_them(kind):
    """model.eval() + loss.backward() used to yield silently missing parameter gradients."""
    m = M.build_model(kind).to("meta").eval()
    out = m(_meta_inputs())
    with pytest.raises(RuntimeError, match="training mode"):
        out["energy"].sum().backward()
    # ... while Forces' own first-order gradient w.r.t. the positions (autograd.grad, atomistic/response.py:63-68) is served
    assert out["forces"].shape == (12, 3)
    # second order in eval mode (create_graph) is refused as well instead of silently dropping terms
    inp = _meta_inputs()
    inp["_positions"].requires_grad_()
    inp = m.input_modules[0](inp)
    x = m.representation(inp)["scalar_representation"]
    with pytest.raises(RuntimeError, match="training mode"):
        torch.autograd.grad(x.sum(), inp["_positions"], create_graph=True)
    m.train()
    out = m(_meta_inputs())
    (out["energy"].sum() + out["forces"].sum()).backward()
    assert all(p.grad is not None for p in m.parameters())


@pytest.mark.parametrize("rep_cls,radial,kw", [(SchNet, GaussianRBF, {}), (SchNet, BesselRBF, {"n_filters": 64}),
                                                 (PaiNN, GaussianRBF, {}), (PaiNN, BesselRBF, {"shared_filters": True, "shared_interactions": True})])
def test_representations_script_like_the_reference(rep_cls, radial, kw):
    """reference tests/nn/test_schnet.py:83-96: torch.jit.script(SchNet(...)) and call it; the scripted graph holds the fused operator."""
    rep = rep_cls(128, 3, radial(20, 5.0), CosineCutoff(5.0), **kw).to("meta").eval()
    srep = torch.jit.script(rep)
    inp = _meta_inputs()
    inp["_Rij"] = torch.zeros(40, 3, device="meta")
    out = srep(inp)
    assert out["scalar_representation"].shape == (12, 128)
    if rep_cls is PaiNN:
        assert out["vector_representation"].shape == (12, 3, 128)
    assert ("spk_hip::schnet" if rep_cls is SchNet else "spk_hip::painn") in str(srep.graph)
    buf = io.BytesIO()
    torch.jit.save(srep, buf)
    buf.seek(0)
    again = torch.jit.load(buf)
    assert again(dict(inp))["scalar_representation"].shape == (12, 128)


def test_trainable_rbf_keeps_the_fused_eval_path_and_still_scripts():
    """GaussianRBF(trainable=True) (nn/radial.py:40-45): in eval mode offsets / widths are plain operands of the fused operator; in
    training mode the closed operators spk_hip::radial_d / radial_c carry their gradients (tests/test_train_autograd.py)."""
    rep = SchNet(64, 2, GaussianRBF(16, 5.0, trainable=True), CosineCutoff(5.0)).to("meta").eval()
    assert rep._fused is True and isinstance(rep.radial_basis.offsets, torch.nn.Parameter)
    srep = torch.jit.script(rep)
    assert "spk_hip::schnet(" in str(srep.graph) and "spk_hip::radial_d" in str(srep.radial_basis.graph)
    inp = _meta_inputs()
    inp["_Rij"] = torch.zeros(40, 3, device="meta")
    assert srep(inp)["scalar_representation"].shape == (12, 64)


def test_reference_model_with_installed_hip_classes_scripts_like_spkdeploy(tmp_path):
    """The reference's NeuralNetworkPotential / Atomwise / Forces code (scripted from its source) around the HIP classes
    -- the archive `spkdeploy` writes and pair_schnetpack.cpp:128 loads; needs the reference SOURCE (build container)."""
    from oracle import build_ref, refshim
    if not refshim.available() or refshim.sourceless():
        pytest.skip("reference source text not present")
    paths = build_ref.build_deployed(verbose=False)
    assert len(paths) == 2
    for p in paths:
        extra = {"cutoff": ""}
        jm = torch.jit.load(p, map_location="cpu", _extra_files=extra)
        assert float(extra["cutoff"]) == 5.0                           # spkdeploy:36 / pair_schnetpack.cpp:129-131
        assert "spk_hip::painn" in str(jm.representation.graph)
        assert "spk_hip::pairwise" in str(getattr(jm.input_modules, "0").graph)
        assert type(jm).__name__ == "RecursiveScriptModule" and jm.original_name == "NeuralNetworkPotential"


def test_cpp_libtorch_client_builds_loads_the_archive_and_runs_it_on_the_host(tmp_path):
    """examples/native/spk_jit_client.cpp (the C++ side of interfaces/lammps/pair_schnetpack.cpp:125-131, :328) on the build box:
    it builds against libtorch, dlopens the two operator libraries, loads a `spkdeploy` archive with its cutoff metadata and -- with
    no ROCm device here -- evaluates it on the HOST: the scripted mirrors take their ATen route (nn/fallback.py; until round 5 the
    first operator refused the CPU tensors) and reproduce the reference-generated fixture of the free molecule."""
    import os
    import subprocess
    import numpy as np
    import pytest
    from conftest import load_npz
    from oracle import build_ref
    from schnetpack_amd.csrc import build as B
    exe = B.build_jit_client(verbose=False)
    assert exe and os.path.exists(exe)
    name = build_ref.DEPLOYED[0] if isinstance(build_ref.DEPLOYED, (list, tuple)) else sorted(build_ref.DEPLOYED)[0]
    p = build_ref.deployed_path(name)
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/deployed/*.pt not built")
    g = load_npz("deploy_painn.npz")
    t = "%s_free_" % name
    sysf = str(tmp_path / "system.bin")
    Z, R = np.asarray(g[t + "Z"]), np.asarray(g[t + "R"])
    ii, jj, off = np.asarray(g[t + "idx_i"]), np.asarray(g[t + "idx_j"]), np.asarray(g[t + "offsets"])
    cell = np.asarray(g[t + "cell"]) if (t + "cell") in g else np.zeros(9)
    with open(sysf, "wb") as f:
        f.write(np.asarray([Z.shape[0], ii.shape[0]], dtype="<i8").tobytes())
        f.write(Z.astype("<i8").tobytes())
        f.write(R.astype("<f4").tobytes())
        f.write(ii.astype("<i8").tobytes())
        f.write(jj.astype("<i8").tobytes())
        f.write(off.astype("<f4").tobytes())
        f.write(np.asarray(cell, dtype="<f4").reshape(-1)[:9].tobytes())
    r = subprocess.run([exe, p, sysf, B.LIB, B.TORCH_LIB, "cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    lines = r.stdout.strip().splitlines()
    F = np.array([[float(x) for x in ln.split()] for ln in lines[2:]])
    Fr = np.asarray(g[t + "forces"])
    assert F.shape == Fr.shape and np.abs(F - Fr).max() / np.abs(Fr).max() < 1e-5
