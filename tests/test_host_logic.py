"""CPU tests: the C-ABI library builds/loads and exports every symbol that include/spk_hip.h
declares (no compute calls without a GPU); host-side mirrors keep the reference's constructor
signatures, parameter names and seeded initialisation; the product path refuses CPU tensors."""
import os
import re
import subprocess

import pytest
import torch

from conftest import ROOT
from oracle import spk_oracle as O


def header_functions():
    src = open(os.path.join(ROOT, "include", "spk_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spk_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from schnetpack_amd.csrc import build as b
    lib = b.build(verbose=False)
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    declared = header_functions()
    assert len(declared) >= 30
    missing = [f for f in declared if f not in exported]
    assert not missing, missing
    # the ctypes table binds exactly the declared set
    from schnetpack_amd import _lib
    assert sorted(_lib.exported_symbols()) == declared
    handle = _lib.lib()
    assert handle.spk_version() >= 100
    assert handle.spk_get_variant() == _lib.VARIANT_AUTO


def test_library_contains_gfx950_code_object():
    from schnetpack_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    assert b"k_cfconv_mfma" in data and b"k_painn_msg_row" in data and b"k_dense_mfma" in data


def test_compute_entry_points_fail_without_gpu_or_with_cpu_tensors():
    """The operators (torch.ops.spk_hip.*: the CPU dispatch key is a loud refusal) and the C ABI (ops.* through ctypes) never compute on
    the host -- the product path fails loudly without the device.  The MODULE mirrors route host / non-float32 tensors to the reference's
    own ATen formulas instead (SURVEY.md section 8(b) error convention; round 6, nn/fallback.py, tests/test_aten_fallback.py)."""
    from schnetpack_amd import ops, torchops  # noqa: F401
    from schnetpack_amd._lib import SpkHipError
    from schnetpack_amd.nn import CosineCutoff, Dense, GaussianRBF, scatter_add
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.spk_hip.scatter_add(torch.ones(4, 2), torch.tensor([0, 0, 1, 1]), 2, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.spk_hip.dense(torch.ones(2, 8), torch.ones(32, 8), None, 0)
    with pytest.raises(SpkHipError):
        ops.gather(torch.ones(3, 2), torch.tensor([0, 1]))
    with pytest.raises(SpkHipError):
        ops.scatter_add(torch.ones(4, 2), torch.tensor([0, 0, 1, 1]), 2)
    # the mirrors on host tensors: the reference's formulas
    assert torch.equal(scatter_add(torch.ones(4, 2), torch.tensor([0, 0, 1, 1]), 2), torch.full((2, 2), 2.0))
    lin = Dense(8, 32)
    assert torch.allclose(lin(torch.ones(2, 8)), torch.nn.functional.linear(torch.ones(2, 8), lin.weight, lin.bias))
    assert GaussianRBF(20, 5.0).eval()(torch.ones(3)).shape == (3, 20)
    assert torch.allclose(CosineCutoff(5.0).eval()(torch.ones(3)), 0.5 * (torch.cos(torch.ones(3) * torch.pi / 5.0) + 1.0))


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_module_mirrors_reference_names_and_seeded_init(kind):
    from schnetpack_amd.nn import CosineCutoff, GaussianRBF
    from schnetpack_amd.representation import PaiNN, SchNet
    torch.manual_seed(0)
    if kind == "schnet":
        rep = SchNet(128, 3, GaussianRBF(20, 5.0), CosineCutoff(5.0))
        p = O.init_schnet_params()
    else:
        rep = PaiNN(128, 3, GaussianRBF(20, 5.0), CosineCutoff(5.0))
        p = O.init_painn_params()
    sd = rep.state_dict()
    assert set(sd) == set(p)
    assert all(torch.equal(sd[k], p[k]) for k in sd)  # same RNG stream as the reference ctor
    assert float(rep.cutoff) == 5.0 and rep.n_atom_basis == 128
    assert rep.radial_basis.n_rbf == 20
    assert len(rep.interactions) == 3


def test_shared_interactions_and_filters_alias_like_reference():
    from schnetpack_amd.nn import CosineCutoff, GaussianRBF
    from schnetpack_amd.representation import PaiNN, SchNet
    s = SchNet(64, 3, GaussianRBF(20, 5.0), CosineCutoff(5.0), n_filters=32, shared_interactions=True)
    assert s.interactions[0] is s.interactions[2]
    assert s.interactions[0].in2f.weight.shape == (32, 64)
    p = PaiNN(64, 2, GaussianRBF(20, 5.0), CosineCutoff(5.0), shared_filters=True)
    assert p.filter_net.weight.shape == (3 * 64, 20)
    p2 = PaiNN(64, 2, GaussianRBF(20, 5.0), CosineCutoff(5.0))
    assert p2.filter_net.weight.shape == (2 * 3 * 64, 20)


def test_golden_pretrained_state_dict_loads_into_mirror():
    from conftest import golden_params, load_golden
    from schnetpack_amd import model as M
    _, _, meta = load_golden("painn_aspirin_pretrained.npz")
    rep_p, head_p = golden_params(meta)
    m = M.build_model("painn", n_interactions=int(meta["n_interactions"]))
    M.load_reference_params(m, rep_p, head_p)
    assert torch.equal(m.representation.filter_net.weight, rep_p["filter_net.weight"])
    assert m.representation.filter_net.weight.shape[0] == 2 * 384


def test_gaussian_and_bessel_buffers_match_reference_formulas():
    from schnetpack_amd.nn import BesselRBF, CosineCutoff, GaussianRBF
    g = GaussianRBF(20, 5.0)
    off, w = O.gaussian_rbf_params(20, 5.0)
    assert torch.equal(g.offsets, off) and torch.equal(g.widths, w)
    b = BesselRBF(7, 3.0)
    assert torch.allclose(b.freqs.float(), O.bessel_rbf_params(7, 3.0).float())
    c = CosineCutoff(1.8)
    assert c.cutoff.shape == (1,) and abs(float(c.cutoff) - 1.8) < 1e-6
    gt = GaussianRBF(5, 2.0, trainable=True)
    assert len(list(gt.parameters())) == 2


def test_shard_frames_partitions_exactly():
    from schnetpack_amd.parallel import shard_frames
    for total in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            chunks = [shard_frames(total, r, world) for r in range(world)]
            assert chunks[0][0] == 0 and chunks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
            sizes = [hi - lo for lo, hi in chunks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_frames(4, 2, 2)


def test_synthetic_batch_shapes_cfg2():
    from schnetpack_amd import synthetic as S
    b = S.molecule_batch("aspirin", 16, seed=0)
    assert b["Z"].shape[0] == 16 * 21 and b["idx_m"][-1] == 15
    assert b["idx_i"].dtype == torch.int64
    deg = torch.bincount(b["idx_i"], minlength=336).float().mean()
    assert 12.0 < float(deg) < 17.0  # ~14.5 neighbours within 5 A


def test_bench_refuses_to_run_without_a_device():
    """bench.py measures the HIP path only: on a box without a ROCm device it exits non-zero and prints no metric line."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("a ROCm device is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    assert '"metric"' not in res.stdout
    assert "no CPU fallback" in res.stderr


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` without a launcher starts 2 ranks itself (torch.distributed.run, 127.0.0.1) and the line
    says n_gpus = 2; here as a dry run over gloo (rank wiring only -- the measured path needs devices)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPK_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["backend"] == "gloo"
    assert line["max_over_ranks"] == 2.0 and line["units_over_ranks"] == 2.0
    # a launcher that started a different number of ranks than --gpus asks for is an error, not a silent 1-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=600, env=env2)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr
    # without devices the real multi-GPU run refuses loudly
    if not torch.cuda.is_available():
        env3 = {k: v for k, v in env.items() if k != "SPK_BENCH_BACKEND"}
        res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env3)
        assert res.returncode != 0 and "device(s) visible" in res.stderr and '"metric"' not in res.stdout


def test_bench_ranks_kernel_families_not_template_variants():
    """The dominant-kernel rule of bench.py groups the compile-time variants of a kernel (text check: the rule is
    part of the measurement contract described in DESIGN.md section 5)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    assert 'replace("_geom", "")' in src and 'replace("_mu0", "")' in src


def test_complete_intramolecular_pair_list():
    """md.NVESimulation.complete_pair_list: every ordered pair of every molecule, rows ascending, neighbours ascending."""
    from schnetpack_amd.md import NVESimulation
    n_atoms = torch.tensor([3, 1, 4, 2])
    idx_m = torch.repeat_interleave(torch.arange(4), n_atoms)
    ii, jj = NVESimulation.complete_pair_list(idx_m, n_atoms)
    want = [(i, j) for i in range(10) for j in range(10) if i != j and idx_m[i] == idx_m[j]]
    assert list(zip(ii.tolist(), jj.tolist())) == want
