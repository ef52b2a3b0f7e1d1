"""RCCL at world size 1 (/root/reference/src/schnetpack/configs/trainer/ddp_trainer.yaml:4-7 is the cfg-4 contract).

The GPU box has ONE device, so no scaling curve can be measured here -- but every `nccl` branch of the multi-GPU path can execute:
`init_process_group("nccl", device_id=...)`, barriers and the max-over-ranks reductions of bench.py, the flat gradient all-reduce between the
two training graphs (`FlatGradAllReduce` -> RCCL on the stream the graphs replay on), the bead all-gather of `md.RPMDSimulation`.  The ranks are
started exactly the way the driver starts them (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _torchrun(args, timeout=600):
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    detail = os.path.join(tempfile.mkdtemp(prefix="spk_rccl_"), "detail.json")
    env = dict(os.environ, SPK_BENCH_FORCE_DIST="1", SPK_MD_FORCE_COLLECTIVES="1", SPK_BENCH_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--detail", detail] + args
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0]), json.load(open(detail))


def test_eval_line_through_rccl_barriers_and_reductions():
    line, _ = _torchrun(["--steps", "5", "--warmup", "2", "--frames", "32", "--no-md", "--no-sweep", "--no-pmc", "--no-cpu-baseline", "--no-painn", "--no-train", "--no-drop-in"])
    assert line["n_gpus"] == 1 and line["config"]["backend"] == "nccl (RCCL)" and line["config"]["world_size"] == 1
    assert line["value"] > 0 and line["config"]["multi_gpu_measured"] is False


@pytest.mark.parametrize("kind", ["schnet", "painn"])
def test_training_step_all_reduces_its_flat_bucket_through_rccl(kind):
    """configs[3]'s step: backward graph | ONE RCCL all-reduce of the flat bucket | optimizer graph -- and the loss still goes down."""
    line, detail = _torchrun(["--mode", "train", "--kind", kind, "--steps", "30", "--warmup", "4", "--no-pmc", "--no-cpu-baseline"])
    cfg = detail["config"]
    assert cfg["backend"] == "nccl (RCCL)" and cfg["allreduce_between_graphs"] is True and cfg["parallelism"] == "dp1"
    assert line["value"] > 0 and cfg["last_loss"] == cfg["last_loss"] and cfg["last_loss"] < 10 * cfg["first_loss"]
    # the collective really ran (round-5 review: at world size 1 it used to be skipped): one all-reduce per timed step, issued on the
    # flat bucket itself (no staging copy), of the whole bucket (PaiNN + head 589 057 floats = 2.36 MB, SchNet + head 226 945)
    assert cfg["allreduce_calls_timed"] == 30, cfg
    assert cfg["allreduce_buffer_is_flat_bucket"] is True
    assert cfg["allreduce_floats"] == (226945 if kind == "schnet" else 589057), cfg["allreduce_floats"]


@pytest.mark.parametrize("exchange", ["forces", "state"])
def test_bead_parallel_ring_polymer_all_gathers_through_rccl(exchange):
    """md.RPMDSimulation's bead exchange (one all-gather of the forces resp. three of the state per step) on the RCCL backend."""
    line, detail = _torchrun(["--mode", "md", "--workload", "aspirin", "--frames", "4", "--beads", "4", "--bead-parallel", exchange, "--steps", "20", "--warmup", "4"])
    cfg = detail["config"]
    assert cfg["backend"] == "nccl (RCCL)" and cfg["beads"] == 4 and cfg["beads_per_rank"] == 4
    assert cfg["collectives_per_step"] == (1 if exchange == "forces" else 3)
    assert line["value"] > 0 and line["scaling"] == "strong"
