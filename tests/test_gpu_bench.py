"""bench.py's contract on a real device: `--gpus N` starts N ranks by itself, and the default N=1 line carries
`roofline`, `cpu_baseline`, the `md` (ns/day) half of BASELINE's metric and the padded-neighbour `sweep`."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=900, with_detail=False):
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(prefix="spk_bench_"), "detail.json")
    args = list(args) + ["--detail", detail]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    assert res.stdout.strip().splitlines()[-1] == lines[0]          # the LAST stdout line is the JSON line ...
    assert len(lines[0]) < 8192                                     # ... and fits the driver's 8 KB tail
    line = json.loads(lines[0])
    assert line["detail"] == detail
    return (line, json.load(open(detail))) if with_detail else line


def test_gpus_2_runs_two_ranks():
    """Two ranks (sharing the one device of this box over gloo; nccl = RCCL needs one device per rank) through the
    same self-spawn path `bench.py --gpus 8` takes on an 8-GPU node."""
    assert torch.cuda.is_available()
    line = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--frames", "32", "--no-md", "--no-sweep", "--no-pmc", "--no-cpu-baseline"],
                {"SPK_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["config"]["backend"] == "gloo"
    assert line["value"] > 0 and line["scaling"] == "weak" and line["steps"] == 5


def test_default_line_has_both_halves_of_the_metric():
    line, detail = _run(["--steps", "10", "--warmup", "3", "--md-steps", "40", "--water-side", "10", "--no-pmc", "--cpu-reps", "2"], with_detail=True)
    assert line["n_gpus"] == 1 and line["dtype"] == "f32" and line["value"] > 0
    rf = line["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and 0 < rf["frac"] <= 1.0 and rf["peak"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["value"] > 0 and cpu["parity_rel_forces"] < 1e-5
    md = line["md"]
    assert md["aspirin"]["ns_per_day"] > 0 and md["water"]["ns_per_day"] > 0 and md["aspirin"]["trajectories"] == 256
    assert any("frac_of_peak" in v for v in detail["kernels"].values())      # every modelled kernel carries its own roofline fraction (detail file)
    rows = detail["sweep"]["rows"]
    for model in ("schnet", "painn"):
        assert [r["k"] for r in rows if r["list"] == "symmetric" and r["model"] == model] == [16, 32, 64]
        assert [r["k"] for r in rows if r["list"] == "asymmetric" and r["model"] == model] == [32]
    assert all(r["M_edge_messages_per_s"] > 0 and r["E"] == r["N"] * r["k"] for r in rows)
    assert len(line["sweep"]["rows"]) == len(rows) and "columns" in line["sweep"]
    # the scatter_add roofline distinguishes the Infinity-Cache-resident replay from the DRAM-streaming cases
    sc = line["scatter_add"]
    assert sc["cache_resident"]["frac"] > 0 and sc["dram_rotating"]["working_set_MB"] > 512 and sc["frac_dram"] == min(
        v["frac"] for k, v in sc.items() if k.startswith("dram_"))
    assert sc["dram_rotating"]["frac_of_measured_copy"] <= 1.05
    # the edge of the molecule regime is a row of the line
    assert {r[1] for r in line["molecule_cliff"]["rows"]} == {21, 29, 42, 60}
