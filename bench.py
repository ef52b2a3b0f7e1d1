#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): M edge-messages/s of the eval-mode force call
(PairwiseDistances -> SchNet(128, 3 interactions, 20 Gaussians, cosine cutoff 5 A) -> Atomwise ->
Forces) on a 256-frame MD17-aspirin batch (configs[1]; N = 5376 atoms, E ~ 77.9k directed edges),
synthetic jittered frames, seeded random-init weights, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE).  A "step" is one complete force
call (forward + first-order backward to the positions) over one batch.  The path shards by
independent molecules: every rank owns its own 256-frame batch (weak scaling), no data-path
collective.  edge-messages/s = E * n_interactions * steps / time, summed over ranks.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the launch
stream in a separate eager pass) and `cpu_baseline` (the CPU oracle = torch restatement of the
reference, timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # same guide: dense f16 / bf16 matrix peak (16 x the fp32 matrix rate); the split-precision path spends 3 f16 products per fp32 product
HBM_PEAK_GBS = 8000.0          # same guide, HBM3E spec peak
RAMP_S = 0.15                  # untimed clock ramp in front of the warm-up steps of the eval legs (seconds of replays)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--kind", default="schnet", choices=["schnet", "painn"])
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--workload", default="aspirin", choices=["aspirin", "water"],
                    help="aspirin: configs[1]/[2] (256-frame MD17 batch, the default); water: configs[4] per-GPU "
                         "share (one 32k-atom bulk-water PBC replica / PIMD bead per GPU)")
    ap.add_argument("--water-side", type=int, default=22, help="molecules per box edge (22 -> 31 944 atoms)")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the force call in a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=15)
    ap.add_argument("--md-shell", type=float, default=2.0, help="--mode md: neighbour-list skin in Angstrom")
    ap.add_argument("--beads", type=int, default=1, help="--mode md: ring-polymer MD with this many beads per GPU (folded into the batch)")
    ap.add_argument("--mode", default="eval", choices=["eval", "train", "md"],
                    help="eval: the headline force call (default).  train: configs[3] — one AdamW step of the force-matching "
                         "loss on --train-frames aspirin frames per GPU, gradients averaged with one flat all-reduce")
    ap.add_argument("--train-frames", type=int, default=8)
    ap.add_argument("--variant", default="auto", choices=["auto", "simple", "mfma", "directed", "pair", "mol"])
    ap.add_argument("--no-md", action="store_true", help="skip the `md` sub-object (ns/day of the on-device NVE loop) of the default line")
    ap.add_argument("--no-sweep", action="store_true", help="skip the `sweep` sub-object (padded-neighbour k = 16/32/64 graphs)")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect the HBM counters of the dominant kernel in this run")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the few eager force calls rocprofv3 --pmc wraps
    ap.add_argument("--md-steps", type=int, default=200)
    ap.add_argument("--thermostat", default="auto", choices=["auto", "none", "pile"],
                    help="--mode md --beads B: PILE-L NVT (300 K, tau = 100 fs; md_configs/dynamics/thermostat/pile_local.yaml).  auto = pile for ring polymers")
    ap.add_argument("--bead-parallel", nargs="?", const="state", default=None, choices=["state", "forces"],
                    help="--mode md --beads B --gpus N: ONE ring polymer of B beads spread over the N ranks (B / N beads each) instead of N "
                         "independent replicas.  state: ranks hold only their beads, all-gather of (q, p) in the main step + one of p per "
                         "thermostat application; forces: integrator state replicated, the one all-gather of a step carries the forces")
    ap.add_argument("--no-painn", action="store_true", help="skip the `painn` sub-object (configs[2]) of the default line")
    ap.add_argument("--no-train", action="store_true", help="skip the `train` sub-object (configs[3] per-GPU share) of the default line")
    ap.add_argument("--no-pimd", action="store_true", help="skip `md.water_pimd` (configs[4]: PaiNN, 8 beads, RPMD + PILE-L) of the default line")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the `drop_in` sub-object (the reference's own NeuralNetworkPotential after install())")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the FULL record of the run goes (every note, per-kernel table and counter detail); the stdout line is its compact, numbers-only face")
    ap.add_argument("--dry-run", action="store_true", help="rank wiring only (launcher, process group, barrier, max-over-ranks reduction), no device work: for the CPU test of --gpus N")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


LINE_LIMIT = 8192              # bytes: the driver keeps an 8 KB tail of stdout -- the final line must fit in it


def _sig(v, n=5):
    """Floats to n significant digits (the full precision stays in the detail file)."""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (n, v))


def _pick(d, keys):
    return {k: _sig(d[k]) for k in keys if isinstance(d, dict) and d.get(k) is not None}


def compact_roofline(rf, brief=False):
    """Numbers only: the contract's fields + the counter-derived ones; every explanation lives in the detail file."""
    if not isinstance(rf, dict):
        return None
    out = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac"))
    if isinstance(out.get("kernel"), str):
        out["kernel"] = out["kernel"][:48]
    out["traffic"] = _sig(rf.get("traffic"))
    out.update(_pick(rf, ("avg_launch_us", "frac_by_convention", "issued_frac_of_peak", "mfma_busy_frac", "traffic_over_B_min", "frac_of_split_path_roof")))
    if isinstance(rf.get("mfma_inputs"), str):
        out["mfma_inputs"] = "f16x2" if rf["mfma_inputs"].startswith("f16x2") else "f32"
    if not brief:
        out.update(_pick(rf, ("peak_split_path",)))
        out.update(_pick(rf, ("algorithmic_per_launch", "algorithmic_per_step", "achieved_by_convention", "executed_frac_of_peak", "issued_over_useful",
                              "B_min_bytes_per_launch")))
    m = rf.get("measured")
    if isinstance(m, dict):
        out.update(_pick(m, ("bound_measured",)))
        if "hbm_frac_of_peak" in m:
            out["hbm_frac_measured"] = _sig(m["hbm_frac_of_peak"])
        if "mfma_frac_of_peak" in m and not brief:
            out["mfma_issued_frac_measured"] = _sig(m["mfma_frac_of_peak"])
    elif rf.get("traffic_frac_of_hbm_peak") is not None:
        out["hbm_frac_measured"] = _sig(rf["traffic_frac_of_hbm_peak"])
    fc = rf.get("force_call")
    if isinstance(fc, dict) and not brief:
        out["force_call"] = _pick(fc, ("frac", "issued_frac"))
    return out


def compact_cpu(c, brief=False):
    if not isinstance(c, dict):
        return None
    out = _pick(c, ("value", "unit", "cores", "kind"))
    if brief:
        out.pop("unit", None)
        out.update(_pick(c, ("parity_rel_forces", "first_loss_rel_diff")))
        return out
    out["sample"] = str(c.get("sample", ""))[:72]
    out.update(_pick(c, ("parity_rel_forces", "parity_rms_forces", "parity_rel_energy", "first_loss_rel_diff")))
    return out


def compact_leg(o, extra=()):
    """A sub-object of the line (painn / water.* / train.*): value, time, compact roofline and CPU baseline."""
    if not isinstance(o, dict):
        return None
    if "error" in o:
        return {"error": str(o["error"])[:120]}
    out = _pick(o, ("value", "unit", "ms_per_step", "steps", "launches_per_step", "n_atoms", "n_edges") + tuple(extra))
    out["roofline"] = compact_roofline(o.get("roofline"), brief=True)
    out["cpu_baseline"] = compact_cpu(o.get("cpu_baseline"), brief=True)
    return out


def compact_line(full, detail_path):
    """The ONE stdout line: the contract's keys + numbers-only sub-objects, < LINE_LIMIT bytes; `detail` names the file with the rest."""
    line = {k: _sig(full.get(k), 7) for k in ("metric", "value", "value_without_ramp", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    line["config"] = dict(_pick(cfg, ("workload", "model", "n_atoms", "n_edges", "frames_per_s", "parallelism", "world_size", "backend", "hip_graph", "variant", "ramp_s",
                                      "multi_gpu_measured", "beads", "beads_per_rank", "thermostat", "collectives_per_step", "trajectories_per_gpu",
                                      "aggregate_ns_per_day", "kinetic_temperature_K", "first_loss", "last_loss", "allreduce_between_graphs",
                                      "allreduce_calls_timed", "allreduce_buffer_is_flat_bucket", "split_precision_matrix_path", "water_atoms", "water_pairs")))
    if full.get("launches_per_step") is not None:
        line["launches_per_step"] = full["launches_per_step"]
    if isinstance(line["config"].get("workload"), str) and len(line["config"]["workload"]) > 100:
        line["config"]["workload"] = line["config"]["workload"][:97] + "..."
    if isinstance(line["config"].get("parallelism"), str):
        line["config"]["parallelism"] = line["config"]["parallelism"][:64]
    line["roofline"] = compact_roofline(full.get("roofline"))
    plw = full.get("parity_ledger_worst")
    if isinstance(plw, dict) and isinstance(plw.get("worst_of_the_1e-5_comparisons"), dict):
        w_ = plw["worst_of_the_1e-5_comparisons"]
        line["parity_ledger_worst"] = {"quantity": w_["quantity"], "fixture": str(w_["fixture"])[:40], "value": _sig(w_["value"]), "bound": w_["bound"], "records": plw.get("records")}
    line["cpu_baseline"] = compact_cpu(full.get("cpu_baseline"))
    if full.get("painn") is not None:
        line["painn"] = compact_leg(full["painn"])
    if isinstance(full.get("water"), dict):
        line["water"] = {k: compact_leg(v) for k, v in full["water"].items()}
    if isinstance(full.get("train"), dict):
        line["train"] = {k: compact_leg(v) for k, v in full["train"].items()}
    if isinstance(full.get("md"), dict):
        line["md"] = {k: (_pick(v, ("ns_per_day", "ms_per_step", "n_atoms", "trajectories", "beads", "dt_fs", "rebuilds", "kinetic_temperature_K")) if "error" not in v else {"error": str(v["error"])[:120]})
                      for k, v in full["md"].items() if isinstance(v, dict)}
    sw = full.get("sweep")
    if isinstance(sw, dict) and "rows" in sw:
        line["sweep"] = {"kind": sw.get("kind"), "N": sw["rows"][0]["N"] if sw["rows"] else None,
                         "rows": [[r.get("model", sw.get("kind")), r["list"][:4], r["k"], _sig(r["ms_fwd_bwd"]), _sig(r["M_edge_messages_per_s"])] for r in sw["rows"]],
                         "columns": ["model", "list", "k", "ms_fwd_bwd", "M_edge_messages_per_s"]}
    elif isinstance(sw, dict):
        line["sweep"] = {"error": str(sw.get("error"))[:120]}
    di = full.get("drop_in")
    if isinstance(di, dict):
        line["drop_in"] = {k: _pick(v, ("ms_per_call", "M_edge_messages_per_s", "rel_diff_forces_vs_mirror_model")) for k, v in di.items() if isinstance(v, dict)}
    sc = full.get("scatter_add")
    if isinstance(sc, dict):
        line["scatter_add"] = {k: (_pick(v, ("shape", "us", "achieved", "frac", "frac_of_measured_copy", "working_set_MB")) if isinstance(v, dict) else _sig(v)) for k, v in sc.items()
                               if k not in ("note",)}
    nb = full.get("neighbor_list")
    if isinstance(nb, dict):
        line["neighbor_list"] = _pick(nb, ("pairs", "matches_input_list", "build_ms", "M_pairs_per_s", "cpu_oracle_ms"))
    mc = full.get("molecule_cliff")
    if isinstance(mc, dict):
        line["molecule_cliff"] = mc if "error" not in mc else {"error": str(mc["error"])[:120]}
    line["detail"] = detail_path
    # the bound is a hard one: shed the optional sub-objects (least important first) rather than print a line the driver cannot parse
    for drop in ("neighbor_list", "drop_in", "molecule_cliff", "sweep", "scatter_add", "md", "train", "water", "painn"):
        if len(json.dumps(line)) < LINE_LIMIT - 256:
            break
        line.pop(drop, None)
        line.setdefault("shed_for_size", []).append(drop)
    return line


def emit(full, detail_path):
    """Write the full record next to the script, print its compact face as the last stdout line."""
    try:
        with open(detail_path, "w") as fh:
            json.dump(full, fh, indent=1, default=str)
        shown = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
    except OSError as exc:  # pragma: no cover
        sys.stderr.write("[bench] could not write %s: %s\n" % (detail_path, exc))
        shown = None
    line = compact_line(full, shown)
    s = json.dumps(line, allow_nan=False)
    assert len(s) < LINE_LIMIT, len(s)
    print(s, flush=True)
    return line


def dry_run(args, rank, world):
    """The multi-rank control flow of the bench without device work (CPU test of `--gpus N`): process group, barrier,
    max-over-ranks time, sum-over-ranks units; rank 0 prints a line marked `dry_run` that carries no measurement."""
    import torch.distributed as dist
    tmax, units = 1.0 + rank, 1.0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SPK_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
        dist.barrier()
        t = torch.tensor([tmax, units], dtype=torch.float64)
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tmax, units = float(tm[0]), float(t[1])
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "metric": None, "value": None, "n_gpus": world, "max_over_ranks": tmax, "units_over_ranks": units,
                          "backend": dist.get_backend() if world > 1 else None}))
    if world > 1:
        dist.destroy_process_group()


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks exactly the way the driver would
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), one rank per GPU
    over RCCL, and pass rank 0's JSON line through.  Fails loudly when the node has fewer devices."""
    import subprocess
    backend = os.environ.get("SPK_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if backend == "nccl" and have < args.gpus and not args.dry_run:
        raise SystemExit("bench.py --gpus %d: only %d ROCm device(s) visible (one rank per GPU; set SPK_BENCH_BACKEND=gloo to "
                         "let ranks share a device for a control-flow test)" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


# rocprofv3 kernel-name patterns of the profile tags (spk_profile_report) -- used to attribute PMC counters
PMC_TAGS = [
    ("schnet_mol_fwd", r"k_schnet_mol_fwd<"), ("schnet_mol_bwd", r"k_schnet_mol_bwd<"),
    ("painn_mol_fwd", r"k_painn_mol_fwd<"), ("painn_mol_bwd", r"k_painn_mol_bwd<"),
    ("cfconv_fwd_rowtile", r"k_cfconv_rowtile_fwd<"), ("cfconv_fwd_pair", r"k_cfconv_pair<[^>]*false, false, false>|k_cfconv_pair_sp<"), ("cfconv_bwd_pair_gs_geom", r"k_cfconv_pair_t<[^>]*true, true, true>|k_cfconv_pair_t_sp<\d+, \d+, true>"),
    ("cfconv_bwd_pair_gs", r"k_cfconv_pair_t<[^>]*true, true, false>|k_cfconv_pair_t_sp<\d+, \d+, false>"), ("cfconv_bwd_pair", r"k_cfconv_pair_t<[^>]*true, false"),
    ("cfconv_fwd_mol", r"k_cfconv_mol<[^>]*false>"), ("cfconv_bwd_mol", r"k_cfconv_mol<[^>]*true>"),
    ("cfconv_fwd_mfma", r"k_cfconv_mfma<[^>]*false, (true|false)>"), ("cfconv_bwd_mfma_sym", r"k_cfconv_mfma<[^>]*true, true>"),
    ("cfconv_bwd_mfma_atomic", r"k_cfconv_mfma<[^>]*true, false>"),
    ("painn_msg_fwd_row", r"k_painn_msg_row<\d+, \d+, false, false, false[,>]"), ("painn_msg_bwd_row", r"k_painn_msg_row<\d+, \d+, true, false, false[,>]"),
    ("painn_msg_fwd_row_mu0", r"k_painn_msg_row<\d+, \d+, false, false, true[,>]"), ("painn_msg_bwd_row_geom", r"k_painn_msg_row<\d+, \d+, true, true"),
    ("painn_msg_fwd_tile_mu0", r"k_painn_msg_tile<\d+, \d+, true"), ("painn_msg_fwd_tile", r"k_painn_msg_tile<"),
    ("painn_msg_bwd_tile_geom", r"k_painn_msg_tile_bwd<\d+, \d+, true"), ("painn_msg_bwd_tile", r"k_painn_msg_tile_bwd<\d+, \d+, false"),
    # row-tile kernels (round 6): <F, KPB, geometry, sums, mu == 0, batch, skin> / <F, KPB, mu == 0, batch, skin>
    ("painn_msg_bwd_rowtile_geom", r"k_painn_msg_rowtile_bwd<\d+, \d+, true, false, true"), ("painn_msg_bwd_rowtile_g", r"k_painn_msg_rowtile_bwd<\d+, \d+, true, false, false"),
    ("painn_msg_bwd_rowtile_t", r"k_painn_msg_rowtile_bwd<\d+, \d+, false, true"),
    ("painn_msg_fwd_rowtile_mu0", r"k_painn_msg_rowtile_fwd<\d+, \d+, true"), ("painn_msg_fwd_rowtile", r"k_painn_msg_rowtile_fwd<\d+, \d+, false"),
    ("painn_mixing_fwd", r"k_painn_mixing_fwd"), ("painn_mixing_bwd", r"k_painn_mixing_bwd"),
    ("dense_chain", r"k_dense_chain"), ("scatter_add_segsum", r"k_segsum<4, \d+>"),
]


def csrc_digest():
    """sha256 (16 hex digits) over the kernel sources: ties a PMC record to the kernel revision it was taken from."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "schnetpack_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def parity_ledger_worst():
    """Worst record of the last parity ledger the GPU test-suite wrote (tests/conftest.py::record_parity -> gpurun_out/parity_ledger.json, copied to
    profiles/rNN_parity_ledger.json): which quantity of which fixture came closest to north_star's 1e-5."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_ledger.json")))
    live = os.path.join(ROOT, "gpurun_out", "parity_ledger.json")
    path = live if os.path.exists(live) else (cands[-1] if cands else None)
    if path is None:
        return None
    try:
        led = json.load(open(path))
        recs = led["records"] if isinstance(led, dict) else led
        # the closest approach to ITS OWN bound decides (the 1e-5 comparisons of energies / forces / representations and the looser weight-gradient ones)
        worst, worst_1e5 = None, None
        for r in recs:
            v, tol = r.get("max_rel"), r.get("tolerance")
            if not isinstance(v, (int, float)) or not tol:
                continue
            if worst is None or v / tol > worst["max_rel"] / worst["tolerance"]:
                worst = r
            if tol <= 1e-5 and (worst_1e5 is None or v > worst_1e5["max_rel"]):
                worst_1e5 = r
        pick = lambda r: None if r is None else {"quantity": r["quantity"], "fixture": r["fixture"], "variant": r["variant"], "value": r["max_rel"], "bound": r["tolerance"]}
        return {"closest_to_its_bound": pick(worst), "worst_of_the_1e-5_comparisons": pick(worst_1e5), "records": len(recs), "source": os.path.relpath(path, ROOT)}
    except Exception as exc:  # pragma: no cover
        return {"error": str(exc)[:120]}


def collect_pmc(args, kind, workload, timeout_s=170):
    """HBM-side traffic AND matrix-core work per launch of every hot kernel, measured IN THIS RUN: three `rocprofv3 --pmc` passes
    (FETCH_SIZE and WRITE_SIZE do not share a pass on gfx950; the third carries SQ_INSTS_VALU_MFMA_MOPS_F32, SQ_VALU_MFMA_BUSY_CYCLES
    and GRBM_GUI_ACTIVE; kernel-trace only) over a child of this script that runs five eager force calls of the same workload.
    Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: KiB per dispatch, FETCH_SIZE x 2 on gfx950; one MOPS
    unit = 512 FLOP; busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs.
    Returns {tag: {"read_bytes", "write_bytes", "mfma_mops", "mfma_busy_cycles", "gui_active", "kernel_name"}} or None."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    res = {}
    tmp = tempfile.mkdtemp(prefix="spk_pmc_", dir="/tmp")
    try:
        # (round 6: the split-precision path issues v_mfma_f32_32x32x16_f16 -- counted by SQ_INSTS_VALU_MFMA_MOPS_F16, same 512-FLOP unit --
        #  next to the fp32 instructions of the phases that stayed on v_mfma_f32_32x32x2_f32)
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"):
            out = os.path.join(tmp, counter.split()[0])
            cmd = [prof, "--kernel-trace", "--output-format", "csv", "--pmc"] + counter.split() + ["-d", out, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--kind", kind, "--workload", workload,
                   "--frames", str(args.frames), "--water-side", str(args.water_side), "--variant", args.variant]
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)
                p.wait()
                return None
            if rc != 0:
                if counter.startswith("SQ_"):
                    continue            # the traffic passes are the contract's; the MFMA pass is extra
                return None
            per_disp, names = {}, {}          # (tag, dispatch id) -> {counter: value, "_ns": duration}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    cname = row.get("Counter_Name")
                    if cname not in counter.split():
                        continue
                    name = row["Kernel_Name"]
                    for tag, pat in PMC_TAGS:
                        if re.search(pat, name):
                            d = per_disp.setdefault((tag, row.get("Dispatch_Id")), {})
                            d[cname] = d.get(cname, 0.0) + float(row["Counter_Value"])
                            try:
                                d["_ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                            except (KeyError, ValueError):
                                pass
                            names[tag] = name[:100]
                            break
            med = lambda v: sorted(v)[len(v) // 2]
            # per launch = the MEDIAN over the dispatches of the child (3 force calls x launches per call): one slow first dispatch does not move it
            for tag in {t for t, _ in per_disp}:
                ds = [d for (t, _), d in per_disp.items() if t == tag]
                r = res.setdefault(tag, {"kernel_name": names[tag]})
                for cname in counter.split():
                    vals = [d[cname] for d in ds if cname in d]
                    if not vals:
                        continue
                    per = med(vals)
                    if cname == "FETCH_SIZE":
                        r["read_bytes"] = 2.0 * 1024.0 * per
                    elif cname == "WRITE_SIZE":
                        r["write_bytes"] = 1024.0 * per
                    else:
                        r[{"SQ_INSTS_VALU_MFMA_MOPS_F32": "mfma_mops", "SQ_INSTS_VALU_MFMA_MOPS_F16": "mfma_mops_f16", "SQ_VALU_MFMA_BUSY_CYCLES": "mfma_busy_cycles",
                           "GRBM_GUI_ACTIVE": "gui_active"}[cname]] = per
                    r["launches_" + cname] = len(vals)
                if counter.startswith("SQ_"):
                    # MFMA-pipe busy share per dispatch, two normalisations: GRBM_GUI_ACTIVE (summed over the 8 XCDs) and the dispatch's own duration at the
                    # 2.4 GHz peak engine clock (a lower bound when the clock is lower).  The GUI count of a short run has been seen 6 x too high (VERDICT
                    # round 4: 0.032 vs 0.212 for one binary): it is used only when the clock it implies (GUI / 8 / duration) is a possible one
                    rat_g, rat_t, clk = [], [], []
                    for d in ds:
                        b, g, ns = d.get("SQ_VALU_MFMA_BUSY_CYCLES"), d.get("GRBM_GUI_ACTIVE"), d.get("_ns")
                        if b is not None and g:
                            rat_g.append(b / (1024.0 * g / 8.0))
                        if b is not None and ns:
                            rat_t.append(b / (1024.0 * ns * 2.4))
                        if g and ns:
                            clk.append(g / 8.0 / ns)
                    if rat_g:
                        r["mfma_busy_frac_gui"] = med(rat_g)
                    if rat_t:
                        r["mfma_busy_frac_time"] = med(rat_t)
                    if clk:
                        r["implied_clock_ghz"] = med(clk)
    except Exception as exc:  # pragma: no cover - depends on the profiler
        sys.stderr.write("[bench] PMC pass failed: %s\n" % exc)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res or None


PMC_TRAIN_STEPS = 4


def collect_train_pmc(args, kind, timeout_s=120):
    """HBM-side bytes of ONE training step, measured in this run: two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only)
    over a child of this script that runs PMC_TRAIN_STEPS eager steps of the same configuration; every dispatch of the child is summed
    (library and framework kernels alike) and divided by the number of steps.  KiB per dispatch, FETCH_SIZE x 2 on gfx950
    (/opt/skills/guides/MI355X_MICROARCH.md).  Returns {"read_bytes", "write_bytes", "dispatches_per_step"} or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    res = {}
    tmp = tempfile.mkdtemp(prefix="spk_pmc_train_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--kernel-trace", "--output-format", "csv", "--pmc", counter, "-d", out, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--mode", "train", "--kind", kind, "--train-frames", str(args.train_frames)]
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)
                p.wait()
                return None
            if rc != 0:
                return None
            total, n = 0.0, 0
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter:
                        total += float(row["Counter_Value"])
                        n += 1
            if n == 0:
                return None
            res["read_bytes" if counter == "FETCH_SIZE" else "write_bytes"] = (2.0 if counter == "FETCH_SIZE" else 1.0) * 1024.0 * total / PMC_TRAIN_STEPS
            res["dispatches_per_step"] = round(n / PMC_TRAIN_STEPS, 1)
    except Exception as exc:  # pragma: no cover - depends on the profiler
        sys.stderr.write("[bench] training PMC pass failed: %s\n" % exc)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def reference_model(kind, rep_p, head_p, F, n_int, n_rbf, cutoff):
    """The REFERENCE's own modules (NeuralNetworkPotential + PairwiseDistances + SchNet/PaiNN + Atomwise + Forces,
    configs/model/nnp.yaml) through oracle/refshim.py (from /root/reference, or its byte-compiled build oracle/_ref on
    the GPU box) with the given weights, on the CPU.  None when the reference is not available."""
    try:
        from oracle import refshim              # test infrastructure: used only for the cpu_baseline leg
        if not refshim.available():
            return None
        ns = refshim.load()
        rb, cf = ns.nn.GaussianRBF(n_rbf, cutoff), ns.nn.CosineCutoff(cutoff)
        rep = (ns.schnet.SchNet if kind == "schnet" else ns.painn.PaiNN)(F, n_int, rb, cf)
        aw = ns.atomwise.Atomwise(n_in=F, output_key="energy")
        m = ns.model.NeuralNetworkPotential(rep, input_modules=[ns.distances.PairwiseDistances()], output_modules=[aw, ns.response.Forces()])
        m.representation.load_state_dict(rep_p)
        m.output_modules[0].load_state_dict(head_p)
        return m.eval()
    except Exception as exc:  # pragma: no cover
        sys.stderr.write("[bench] reference modules unavailable (%s); cpu_baseline falls back to the oracle port\n" % exc)
        return None


def reference_inputs(batch):
    n_mol = int(batch["n_mol"])
    return {"_atomic_numbers": batch["Z"], "_positions": batch["R"].clone(), "_idx_i": batch["idx_i"], "_idx_j": batch["idx_j"],
            "_offsets": batch["offsets"], "_idx_m": batch["idx_m"],
            "_cell": batch["cell"].reshape(1, 3, 3) if "cell" in batch else torch.zeros(n_mol, 3, 3),
            "_pbc": torch.zeros(3 * n_mol, dtype=torch.bool), "_n_atoms": torch.bincount(batch["idx_m"], minlength=n_mol)}


def sweep_measure(model, dev, kind, n_atoms=16384, degrees=(16, 32, 64), reps=10):
    """north_star's padded-neighbour sweep: fixed-degree graphs of stated (N, E = N k, F), representation forward +
    first-order backward w.r.t. r_ij (what a force call asks of the hot path), eager launches timed with events.
    `symmetric`: every atom is linked to its k nearest ring neighbours with antisymmetric pair vectors (the full,
    symmetric, i-sorted structure every reference neighbour list has); `asymmetric`: random directed graph (general path)."""
    from schnetpack_amd import synthetic as S
    rep = model.representation
    F = rep.n_atom_basis
    n_int = len(rep.interactions)
    rows = []
    for sym in (True, False):
        for k in (degrees if sym else degrees[1:2]):
            b = S.ring_graph_batch(n_atoms, k, seed=k) if sym else S.random_graph_batch(n_atoms, k, seed=k)
            inp = {"_atomic_numbers": b["Z"].to(dev), "_idx_i": b["idx_i"].to(dev), "_idx_j": b["idx_j"].to(dev)}
            r = b["r_ij"].to(dev).requires_grad_(True)

            def call():
                d = dict(inp)
                d["_Rij"] = r
                x = rep(d)["scalar_representation"]
                return torch.autograd.grad([x.sum()], [r])[0]
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            E = int(b["idx_i"].shape[0])
            rows.append({"model": kind, "list": "symmetric" if sym else "asymmetric", "N": n_atoms, "k": k, "E": E, "F": F,
                         "ms_fwd_bwd": round(ms, 4), "M_edge_messages_per_s": round(E * n_int / ms / 1e3, 1)})
    return {"kind": kind, "what": "representation forward + backward w.r.t. r_ij on fixed-degree graphs, eager launches, per GPU", "rows": rows}


def cliff_measure(models, dev, frames=128, sizes=(21, 29, 42, 60), reps=10):
    """The edge of the molecule regime: eval force calls on batches of compact synthetic molecules of 21 .. 60
    atoms (synthetic.blob_molecule_batch; 29 = the largest QM9 molecule, 42+ = MD22-sized).  A group of <= 32 atoms with <= 384 pairs runs in
    the molecule-resident launches; beyond that the general kernels take over.  Rows: [kind, atoms per molecule, E, ms per call,
    M edge-messages/s, dominant kernel of the call] -- the last column names the path."""
    from schnetpack_amd import _lib, model as M, synthetic as S
    rows = []
    for n in sizes:
        b = S.blob_molecule_batch(n, frames, seed=n)
        inp = M.batch_to_inputs(b, dev)
        E = int(b["idx_i"].shape[0])
        for kind, model in models:
            def call():
                return model(dict(inp))["forces"].detach()
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            _lib.profile_enable(True)
            _lib.profile_report()
            call()
            prof = _lib.profile_report()
            _lib.profile_enable(False)
            dom = max(prof, key=lambda t: prof[t][1]) if prof else None
            rows.append([kind, n, E, _sig(ms), _sig(E * 3 / ms / 1e3), dom])
    return {"frames": frames, "columns": ["kind", "atoms_per_molecule", "E", "ms_per_call", "M_edge_messages_per_s", "dominant_kernel"], "rows": rows}


_WATER = {}


def cached_water_box(n_side, seed):
    """The synthetic bulk-water box (host-side construction takes seconds: built once per run)."""
    from schnetpack_amd import synthetic as S
    if (n_side, seed) not in _WATER:
        _WATER[(n_side, seed)] = S.water_box(n_side=n_side, seed=seed)
    return _WATER[(n_side, seed)]


def algorithmic_work(kind, E, N, n_mol, F, n_int, n_rbf):
    """tag -> (bound, ALGORITHMIC work per launch, executed share of it, perfect-reuse lower bound of the HBM bytes or None).
    SURVEY.md section 8(d) per-unit figures x units per launch: cfconv forward E * 2 (n_rbf nf + nf^2)
    FLOP per directed edge-message, backward 2x; PaiNN message forward E * 3100 + N * 4096 B (no-reuse gather convention),
    backward 2x.  The pair kernels EXECUTE half of the forward figure (one filter per undirected pair) and the saved-filter
    backward 352/608 of that.  B_min (section 8(d), perfect reuse): every atom row read / written once + 28 B of geometry and
    indices per edge -- SchNet cfconv N * 2 * 4F + E * 28, PaiNN message N * (4096 + 2 * 12F) + E * 28."""
    nf = F
    flop_fwd = 2.0 * E * (n_rbf * nf + nf * nf)
    msg_bytes = E * 3100.0 + N * 4096.0
    bmin_cf = N * 8.0 * F + E * 28.0
    bmin_msg = N * (4096.0 + 24.0 * F) + E * 28.0
    bmin_msg0 = N * (4096.0 + 12.0 * F) + E * 28.0
    flop_dense = 2.0 * N * 3 * F * F
    # molecule-resident launches: per pair tile of 32 undirected pairs: hidden layer once (4 x 4 KPB MFMAs), GEMM 2 (4 x 64),
    # incidence accumulation (4 x 32); per molecule and interaction three Dense layers on a 32-row tile (3 x 4 x 64); 1 MFMA = 4096 FLOP
    tiles_mol = (E // 2 + 31 * n_mol) // 32
    kpb = (n_rbf + 7) // 8
    exec_mol = n_int * 4096.0 * (tiles_mol * 4 * (4 * kpb + 64 + 32) + n_mol * 3 * 4 * 64)
    exec_mol_bwd = n_int * 4096.0 * (tiles_mol * 2 * (8 * kpb + 128) + n_mol * 3 * 4 * 64)
    mol_f, mol_b = n_int * (flop_fwd + flop_dense), n_int * (2 * flop_fwd + flop_dense)
    gs = 0.5 * 352.0 / 608.0
    # molecule-resident PaiNN launches: every interaction's message AND mixing / context nets in one launch -- booked, like the
    # kernels they replace, by the no-reuse gather convention: forward sum over interactions of (message + mixing) bytes, backward 2x
    pmol = n_int * (msg_bytes + N * 4096.0)
    pmol_min = n_int * (bmin_msg + N * 4096.0)
    return {
        "painn_mol_fwd": ("hbm", pmol, 1.0, pmol_min), "painn_mol_bwd": ("hbm", 2 * pmol, 1.0, 2 * pmol_min),
        "schnet_mol_fwd": ("mfma", mol_f, exec_mol / mol_f, n_int * bmin_cf), "schnet_mol_bwd": ("mfma", mol_b, exec_mol_bwd / mol_b, 2 * n_int * bmin_cf),
        "cfconv_fwd_mfma": ("mfma", flop_fwd, 1.0, bmin_cf), "cfconv_fwd_simple": ("mfma", flop_fwd, 1.0, bmin_cf),
        "cfconv_fwd_pair": ("mfma", flop_fwd, 0.5, bmin_cf),
        "cfconv_fwd_rowtile": ("mfma", flop_fwd, 1.0, bmin_cf + 0.5 * E * 4.0 * F),      # (round 6) one filter per DIRECTED edge; writes the saved filters of the pairs

        "cfconv_bwd_mfma_sym": ("mfma", 2 * flop_fwd, 1.0, 2 * bmin_cf), "cfconv_bwd_mfma_atomic": ("mfma", 2 * flop_fwd, 1.0, 2 * bmin_cf),
        "cfconv_bwd_simple": ("mfma", 2 * flop_fwd, 1.0, 2 * bmin_cf), "cfconv_bwd_pair": ("mfma", 2 * flop_fwd, 0.5, 2 * bmin_cf),
        "cfconv_bwd_pair_gs": ("mfma", 2 * flop_fwd, gs, 2 * bmin_cf), "cfconv_bwd_pair_gs_geom": ("mfma", 2 * flop_fwd, gs, 2 * bmin_cf),
        "painn_msg_fwd_row": ("hbm", msg_bytes, 1.0, bmin_msg), "painn_msg_fwd_simple": ("hbm", msg_bytes, 1.0, bmin_msg),
        "painn_msg_bwd_row": ("hbm", 2 * msg_bytes, 1.0, 2 * bmin_msg), "painn_msg_bwd_simple": ("hbm", 2 * msg_bytes, 1.0, 2 * bmin_msg),
        "painn_msg_fwd_tile": ("hbm", msg_bytes, 1.0, bmin_msg),
        # first-interaction variants: mu == 0 (its 3F floats per neighbour are not gathered); geometry-only backward
        # (c and mu of the neighbour: the forward's bytes; with mu == 0 only c)
        "painn_msg_fwd_row_mu0": ("hbm", E * 1564.0 + N * 4096.0, 1.0, bmin_msg0), "painn_msg_fwd_tile_mu0": ("hbm", E * 1564.0 + N * 4096.0, 1.0, bmin_msg0),
        "painn_msg_bwd_row_geom": ("hbm", E * 1564.0 + N * 4096.0, 1.0, bmin_msg0), "painn_msg_bwd_tile_geom": ("hbm", E * 1564.0 + N * 4096.0, 1.0, bmin_msg0),
        "painn_msg_bwd_tile": ("hbm", 2 * msg_bytes, 1.0, 2 * bmin_msg),
        # row-tile kernels (round 6): the backward of SURVEY.md 8(d) (2 x the forward's no-reuse bytes) runs as a geometry launch (neighbour c and mu:
        # the forward's bytes) and a transposed-sums launch (neighbour gq and gmu: 2 060 B per edge)
        "painn_msg_fwd_rowtile": ("hbm", msg_bytes, 1.0, bmin_msg), "painn_msg_fwd_rowtile_mu0": ("hbm", E * 1564.0 + N * 4096.0, 1.0, bmin_msg0),
        "painn_msg_bwd_rowtile_g": ("hbm", msg_bytes, 1.0, bmin_msg), "painn_msg_bwd_rowtile_geom": ("hbm", E * 1564.0 + N * 4096.0, 1.0, bmin_msg0),
        "painn_msg_bwd_rowtile_t": ("hbm", E * 2076.0 + N * 4096.0, 1.0, N * (4096.0 + 16.0 * F) + E * 28.0),
    }


def eval_leg(args, kind, workload, model, rep_p, head_p, rank, world, dev, dist, steps, warmup, with_pmc=True, with_cpu=True, cpu_reps=15):
    """One eval-mode force-call measurement (the headline leg, and the `painn` sub-object of the default line): timed
    graph replays bracketed by barriers, per-kernel HIP-event pass, roofline of the dominant kernel (+ in-run PMC traffic),
    the reference on the host cores with the parity of this very batch.  Returns a dict; rank != 0 gets only the timing."""
    from schnetpack_amd import _lib, model as M, synthetic as S
    from schnetpack_amd.parallel import shard_frames
    n_int, F, n_rbf, cutoff = 3, 128, 20, 5.0
    # weak scaling: rank r owns frames [r*frames, (r+1)*frames) of one global seeded trajectory
    lo, hi = shard_frames(args.frames * world, rank, world)
    if workload == "water":
        batch = cached_water_box(args.water_side, rank)   # one replica (bead) per rank
    else:
        batch = S.molecule_batch("aspirin", hi - lo, seed=1000 + rank if world > 1 else 0)
    E = int(batch["idx_i"].shape[0])
    N = int(batch["Z"].shape[0])
    inp = M.batch_to_inputs(batch, dev)

    def force_call():
        out = model(dict(inp))
        # detach: a live autograd graph from an earlier (default-stream) call would be pulled into
        # the HIP-graph capture through the AccumulateGrad node of the positions
        return out["energy"].detach(), out["forces"].detach()

    if args.pmc_child:          # wrapped by `rocprofv3 --pmc` from collect_pmc(): a few eager calls, no output
        for _ in range(5):
            force_call()
        torch.cuda.synchronize()
        return None

    for _ in range(max(warmup, 3)):
        e_ref, f_ref = force_call()
    torch.cuda.synchronize()

    graph = None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    force_call()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                ge, gf = force_call()
            g.replay()
            torch.cuda.synchronize()
            err = float((gf - f_ref).abs().max() / f_ref.abs().max())
            if not (err < 1e-5):
                raise RuntimeError("graph replay deviates from eager: %g" % err)
            graph = g
        except Exception as exc:  # pragma: no cover - depends on the runtime
            sys.stderr.write("[bench] HIP graph capture unavailable (%s); running eager\n" % exc)
            graph = None
            torch.cuda.synchronize()

    step = graph.replay if graph is not None else force_call
    # the same K timed steps WITHOUT the clock ramp first (reported beside the headline as value_without_ramp: what a freshly
    # started process measures over these few milliseconds)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt_cold = time.perf_counter() - t0
    # clock ramp, untimed and BEFORE the W warm-up steps: a freshly started process reaches its sustained clocks only after some
    # tens of milliseconds of work -- with the driver's --steps 20 --warmup 5 the whole measurement is 6 ms long and came out 8 %
    # below the same binary at --steps 200 (one box, alternating runs).  Bounded: RAMP_S seconds of replays, reported in `config`
    t_r = time.perf_counter()
    n_ramp = 0
    while time.perf_counter() - t_r < RAMP_S:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        n_ramp += 20
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if graph is not None:   # the LAST timed replay must still reproduce the eager result (not only the first one)
        err = float((gf - f_ref).abs().max() / f_ref.abs().max())
        if not (err < 1e-5):
            raise RuntimeError("graph replay deviates from eager after the timed loop: %g" % err)
    E_total = E
    if dist is not None:
        rdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        tt = torch.tensor([dt], device=rdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        et = torch.tensor([E], device=rdev, dtype=torch.float64)
        dist.all_reduce(et, op=dist.ReduceOp.SUM)
        E_total = int(et.item())
    value = E_total * n_int * steps / dt / 1e6
    res = {"value": value, "value_without_ramp": E * n_int * steps / dt_cold / 1e6, "dt": dt, "steps": steps, "E": E, "N": N, "frames": hi - lo, "graph": graph is not None, "batch": batch, "inp": inp, "n_ramp": n_ramp,
           "e_ref": e_ref, "f_ref": f_ref, "n_int": n_int, "F": F, "n_rbf": n_rbf, "cutoff": cutoff}
    if rank != 0:
        return res

    # ---------------- per-kernel timing (separate eager pass, HIP events on the launch stream)
    _lib.profile_enable(True)
    _lib.profile_report()
    psteps = min(steps, 20)
    for _ in range(psteps):
        force_call()
    prof = _lib.profile_report()
    _lib.profile_enable(False)
    n_mol = int(batch["n_mol"])
    algo = algorithmic_work(kind, E, N, n_mol, F, n_int, n_rbf)
    kernels = {}
    for tag, (cnt, ms) in prof.items():
        kernels[tag] = {"launches_per_step": cnt / psteps, "avg_us": 1e3 * ms / max(cnt, 1), "us_per_step": 1e3 * ms / psteps}
    # every kernel with an algorithmic-work model also carries its own fraction of the peak (the N-sized PaiNN kernels: SURVEY.md
    # section 8(d): mixing N * 360 kFLOP per interaction forward, 2x backward -- the backward launch leaves the channel-mix
    # transpose to a chain launch, so its own share is the two transposed context layers + the products: N * 2 * 2 (2F F + F 3F))
    mix_flop = N * 2.0 * (3 * F * 2 * F + 2 * F * F + F * 3 * F)
    algo_n = {"painn_mixing_fwd": mix_flop, "painn_mixing_bwd": N * 4.0 * (2 * F * F + F * 3 * F)}
    # the Dense share of the molecule-resident PaiNN launches (context net N * 2 (F F + F 3F) + mixing N * 360 k per interaction; backward
    # 2x) against the fp32 matrix peak, beside their HBM-convention fraction
    pm_dense = n_int * (N * 2.0 * (F * F + 3 * F * F) + mix_flop)
    algo_mfma_view = {"painn_mol_fwd": pm_dense, "painn_mol_bwd": 2 * pm_dense}
    FLAG = "exceeds 1: the algorithmic convention of SURVEY.md 8(d) books more work than this kernel issues (one filter per undirected pair, saved filters, cache-resident gathers); read executed_frac_of_peak / traffic instead"
    for tag, kd in kernels.items():
        if tag in algo:
            bound, work, executed, bmin = algo[tag]
            peak = MFMA_F32_PEAK_TFLOPS * 1e12 if bound == "mfma" else HBM_PEAK_GBS * 1e9
            kd["frac_of_peak"] = round(work / (kd["avg_us"] * 1e-6) / peak, 4)
            kd["executed_frac_of_peak"] = round(executed * kd["frac_of_peak"], 4)
            kd["bound"] = bound
            if kd["frac_of_peak"] > 1.0:
                kd["frac_flag"] = FLAG
            if tag in algo_mfma_view:
                kd["dense_flop_frac_of_mfma_peak"] = round(algo_mfma_view[tag] / (kd["avg_us"] * 1e-6) / (MFMA_F32_PEAK_TFLOPS * 1e12), 4)
        elif tag in algo_n:
            kd["frac_of_peak"] = round(algo_n[tag] / (kd["avg_us"] * 1e-6) / (MFMA_F32_PEAK_TFLOPS * 1e12), 4)
            kd["bound"] = "mfma"
    cand = [t for t in kernels if t in algo]
    roofline = None
    if cand:
        # dominant kernel: compile-time variants of one kernel (the first-interaction forms "_geom" / "_mu0") count as
        # one family when ranking; the family's main member is the one reported
        fam = lambda t: t.replace("_geom", "").replace("_mu0", "")
        fam_time = {}
        for t in cand:
            fam_time[fam(t)] = fam_time.get(fam(t), 0.0) + kernels[t]["us_per_step"]
        top = max(fam_time, key=fam_time.get)
        dom = max([t for t in cand if fam(t) == top], key=lambda t: kernels[t]["us_per_step"])
        bound, work, executed, bmin = algo[dom]
        sec = kernels[dom]["avg_us"] * 1e-6
        if bound == "mfma":
            ach, peak, unit = work / sec / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = work / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        roofline = {"kernel": dom, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "traffic": None,
                    "avg_launch_us": round(kernels[dom]["avg_us"], 2), "algorithmic_per_launch": work,
                    "executed_frac_of_peak": round(executed * ach / peak, 4),
                    "B_min_bytes_per_launch": bmin,
                    "B_min_note": "perfect-reuse lower bound of the HBM bytes of this launch (SURVEY.md 8(d)): every atom row once + 28 B per edge; "
                                  "time at the HBM peak = %.2f us" % (bmin / (HBM_PEAK_GBS * 1e9) * 1e6),
                    "note": "achieved = algorithmic work / HIP-event time of the launch; executed_frac_of_peak counts only the "
                            "work the kernel really issues (pair kernels evaluate one filter per undirected edge)"}
        if roofline["frac"] > 1.0:
            # a saved-filter / one-filter-per-pair kernel issues less work than the convention books: the headline `achieved` / `frac`
            # are then the EXECUTED work (what the matrix core really did) -- never a fraction above 1; the convention's figures stay
            # beside them
            roofline["achieved_by_convention"], roofline["frac_by_convention"] = roofline["achieved"], roofline["frac"]
            roofline["achieved"] = round(executed * ach, 3)
            roofline["frac"] = roofline["executed_frac_of_peak"]
            roofline["executed_per_launch"] = executed * work
            roofline["frac_flag"] = "frac_by_convention " + FLAG + " -- `achieved` and `frac` of this object are the executed work"

    # HBM traffic of the dominant kernel, measured in THIS run (collect_pmc: two `rocprofv3 --pmc` passes over a child of
    # this script); the committed record of an earlier run is only the fallback, and says which kernel revision it is from
    pmc = None
    if roofline is not None and world == 1 and with_pmc and not args.no_pmc:
        pmc = collect_pmc(args, kind, workload)
    # what every kernel DOES, from this run's counters (not from a table of instruction counts): executed matrix-core FLOP =
    # SQ_INSTS_VALU_MFMA_MOPS_F32 x 512, HBM-side bytes = FETCH_SIZE x 2 + WRITE_SIZE, both over the HIP-event time of the launch;
    # MFMA-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).  A kernel below half of BOTH roofs
    # is labelled latency / VALU bound.
    if pmc is not None:
        for tag, c in pmc.items():
            if tag not in kernels:
                continue
            kd, sec = kernels[tag], kernels[tag]["avg_us"] * 1e-6
            meas = {"source": "rocprofv3 --pmc passes of this run"}
            if "read_bytes" in c and "write_bytes" in c:
                meas["hbm_bytes"] = c["read_bytes"] + c["write_bytes"]
                meas["hbm_frac_of_peak"] = round(meas["hbm_bytes"] / sec / (HBM_PEAK_GBS * 1e9), 4)
            if c.get("mfma_mops_f16"):
                # split-precision path: three f16 products stand for one fp32 product, so the issued f16 FLOP / 3 are the fp32-equivalent
                # ("logical") FLOP of the split phases; their own roof is the dense f16 peak
                meas["mfma_f16_flop_issued"] = 512.0 * c["mfma_mops_f16"]
                meas["mfma_f16_frac_of_f16_peak"] = round(meas["mfma_f16_flop_issued"] / sec / (MFMA_F16_PEAK_TFLOPS * 1e12), 4)
            if "mfma_mops" in c:
                meas["mfma_f32_flop_issued"] = 512.0 * c["mfma_mops"]
                meas["mfma_flop_issued"] = 512.0 * c["mfma_mops"] + 512.0 * c.get("mfma_mops_f16", 0.0) / 3.0      # fp32-equivalent
                meas["mfma_frac_of_peak"] = round(meas["mfma_flop_issued"] / sec / (MFMA_F32_PEAK_TFLOPS * 1e12), 4)
                if tag in algo and algo[tag][0] == "mfma":
                    useful = algo[tag][1] * algo[tag][2]
                    meas["mfma_flop_useful_model"] = useful
                    meas["issued_over_useful"] = round(meas["mfma_flop_issued"] / useful, 3) if useful else None
            if "mfma_busy_frac_gui" in c or "mfma_busy_frac_time" in c:
                clk = c.get("implied_clock_ghz")
                gui_ok = "mfma_busy_frac_gui" in c and (clk is None or 1.2 <= clk <= 2.5)
                meas["mfma_busy_frac"] = round(c["mfma_busy_frac_gui"] if gui_ok else c["mfma_busy_frac_time"], 4)
                meas["mfma_busy_norm"] = "GRBM_GUI_ACTIVE" if gui_ok else "dispatch duration x 2.4 GHz"
                if clk is not None:
                    meas["implied_clock_ghz"] = round(clk, 3)
            hf, mf = meas.get("hbm_frac_of_peak"), meas.get("mfma_frac_of_peak")
            if meas.get("mfma_f16_flop_issued", 0.0) > meas.get("mfma_f32_flop_issued", 0.0):
                # the launch ran (mostly) on the f16 instruction: how busy the matrix pipe was is its share of the f16 peak plus the fp32 share
                meas["mfma_inputs"] = "f16x2"
                mf_pipe = meas["mfma_f16_frac_of_f16_peak"] + meas.get("mfma_f32_flop_issued", 0.0) / sec / (MFMA_F32_PEAK_TFLOPS * 1e12)
            else:
                meas["mfma_inputs"] = "f32" if "mfma_mops" in c else None
                mf_pipe = mf
            if hf is not None and mf_pipe is not None:
                meas["matrix_pipe_frac"] = round(mf_pipe, 4)
                meas["bound_measured"] = "latency/valu" if max(hf, mf_pipe) < 0.5 else ("mfma" if mf_pipe >= hf else "hbm")
            kd["measured"] = meas
            # executed fraction: HBM-bound kernels -- the measured bytes; MFMA-bound kernels -- the ISSUED matrix-core work of the counters
            # divided by the issued / useful ratio where a model of the useful work exists (rows of padding in the 32-row tiles are issued,
            # not useful), else the issued work itself
            if kd.get("bound") == "mfma" and mf is not None:
                kd["issued_frac_of_peak"] = mf
                # (never above the issued work: where the model of the useful work books MORE than the counters saw issued -- one filter per
                #  undirected pair against the convention's per-direction count -- the issued work is the executed work)
                kd["executed_frac_of_peak"] = round(mf / max(1.0, meas["issued_over_useful"]), 4) if meas.get("issued_over_useful") else mf
            elif kd.get("bound") == "hbm" and hf is not None:
                kd["executed_frac_of_peak"] = hf
        if roofline is not None and "measured" in kernels.get(roofline["kernel"], {}):
            m_ = kernels[roofline["kernel"]]["measured"]
            roofline["measured"] = m_
            ex = kernels[roofline["kernel"]].get("executed_frac_of_peak")
            if roofline["bound"] == "mfma" and "mfma_frac_of_peak" in m_:
                roofline["issued_frac_of_peak"], roofline["issued_over_useful"], roofline["mfma_busy_frac"] = m_["mfma_frac_of_peak"], m_.get("issued_over_useful"), m_.get("mfma_busy_frac")
            if ex is not None:
                roofline["executed_frac_of_peak"] = ex
                roofline["executed_frac_source"] = "counters of this run (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512, resp. FETCH_SIZE x 2 + WRITE_SIZE, over the HIP-event time)"
                if roofline["bound"] == "mfma" and "frac_by_convention" not in roofline and roofline["frac"] > m_.get("mfma_frac_of_peak", 1.0):
                    # the convention books more matrix-core work than the counters of this run saw ISSUED: same treatment as a fraction above 1
                    roofline["achieved_by_convention"], roofline["frac_by_convention"] = roofline["achieved"], roofline["frac"]
                    roofline["frac_flag"] = ("frac_by_convention exceeds the issued matrix-core work of this run's counters (the algorithmic convention of SURVEY.md 8(d) books "
                                             "more than the kernel issues: one filter per undirected pair); `achieved` and `frac` of this object are the executed work")
                if roofline["bound"] == "hbm" and "frac_by_convention" not in roofline and "hbm_frac_of_peak" in m_:
                    # HBM-bound legs get the treatment the MFMA legs have: `frac` / `achieved` are the bytes the counters of this
                    # run saw moved (FETCH_SIZE x 2 + WRITE_SIZE) over the launch time; the no-reuse gather convention of SURVEY.md 8(d) stays beside them
                    roofline["achieved_by_convention"], roofline["frac_by_convention"] = roofline["achieved"], roofline["frac"]
                    roofline["frac_flag"] = ("`achieved` and `frac` are the HBM-side bytes of this run's counters over the launch time; frac_by_convention books the "
                                             "no-reuse gather bytes of SURVEY.md 8(d), most of which the L2 / Infinity Cache serve")
                if "frac_by_convention" in roofline:      # the headline fraction is the executed one
                    roofline["frac"] = ex
                    roofline["achieved"] = round(ex * roofline["peak"], 3)
    if roofline is not None and pmc is not None and roofline["kernel"] in pmc and "read_bytes" in pmc[roofline["kernel"]] and "write_bytes" in pmc[roofline["kernel"]]:
        c = pmc[roofline["kernel"]]
        roofline["traffic"] = c["read_bytes"] + c["write_bytes"]
        roofline["traffic_over_B_min"] = round(roofline["traffic"] / bmin, 2) if bmin else None
        roofline["traffic_frac_of_hbm_peak"] = round(roofline["traffic"] / (kernels[roofline["kernel"]]["avg_us"] * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)
        roofline["traffic_detail"] = {"read_bytes": c["read_bytes"], "write_bytes": c["write_bytes"], "kernel_name": c["kernel_name"],
                                      "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) over 5 eager "
                                                "force calls of this workload; KiB per dispatch, FETCH_SIZE x 2 (gfx950); Infinity-Cache hits are counted",
                                      "csrc_digest": csrc_digest(),
                                      "all_kernels": {t: {"read_bytes": v.get("read_bytes"), "write_bytes": v.get("write_bytes")} for t, v in pmc.items()}}
    else:
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic_%s_%s.json" % (kind, workload))
        if roofline is not None and os.path.exists(pmc_file):
            try:
                rec = json.load(open(pmc_file))
                c = rec["counters"].get(roofline["kernel"])
                if c and "FETCH_SIZE_raw_per_launch" in c and "WRITE_SIZE_raw_per_launch" in c:
                    rd = 2.0 * 1024.0 * c["FETCH_SIZE_raw_per_launch"]
                    wr = 1024.0 * c["WRITE_SIZE_raw_per_launch"]
                    roofline["traffic"] = rd + wr
                    roofline["traffic_detail"] = {"read_bytes": rd, "write_bytes": wr, "source": os.path.relpath(pmc_file, ROOT) + " (committed record of an earlier run)",
                                                  "record_csrc_digest": rec.get("csrc_digest"), "csrc_digest": csrc_digest(),
                                                  "record_is_current": rec.get("csrc_digest") == csrc_digest()}
            except Exception as exc:  # pragma: no cover
                sys.stderr.write("[bench] could not read %s: %s\n" % (pmc_file, exc))

    # which matrix instruction did the work (round 6): with the split-precision path on, the filter / Dense products of the fused kernels are
    # three v_mfma_f32_32x32x16_f16 products of (high, low) fp16 operand pairs with fp32 accumulation per fp32 product -- fp32-quality
    # results (csrc/spk_split.h; profiles/r06_split_mfma.md).  `dtype` stays f32; `frac` stays against the fp32 matrix peak (157.3 TF: the
    # roof of the arithmetic the contract names); frac_of_split_path_roof puts the same executed work over the roof of the path that ran,
    # dense f16 peak / 3 products.
    if roofline is not None and roofline["bound"] == "mfma":
        split_on = bool(_lib.get_split())
        if isinstance(roofline.get("measured"), dict) and roofline["measured"].get("mfma_inputs"):
            split_on = roofline["measured"]["mfma_inputs"] == "f16x2"        # what the counters of this launch say, not what the switch says
        elif not any(t in roofline["kernel"] for t in ("schnet_mol", "painn_mol", "cfconv_fwd_pair", "cfconv_fwd_rowtile", "cfconv_bwd_pair_gs")):
            split_on = False        # (round 6: the pair kernels of the box regime have split forms too, n_filters = 128)
        roofline["mfma_inputs"] = ("f16x2 split: fp16 high + 2^-11-scaled fp16 low operand pairs, 3 x v_mfma_f32_32x32x16_f16 per fp32 product, fp32 accumulate"
                                   if split_on else "f32 (v_mfma_f32_32x32x2_f32)")
        if split_on:
            roof = MFMA_F16_PEAK_TFLOPS / 3.0
            useful = roofline.get("executed_frac_of_peak", roofline["frac"]) * MFMA_F32_PEAK_TFLOPS
            roofline["peak_split_path"] = round(roof, 1)
            roofline["frac_of_split_path_roof"] = round(useful / roof, 4)
            roofline["split_note"] = ("executed fp32-equivalent TFLOP/s (= frac x 157.3) over 2500 / 3 TFLOP/s; phases that stayed on the fp32 instruction "
                                      "(in2f beside the first pair tile, the energy head) are booked at the same roof")
            if roofline.get("executed_frac_of_peak", roofline["frac"]) > 1.0:
                # (a launch that issues all the work the convention books -- the row-tile forward: one filter per directed edge -- can exceed the
                #  fp32 matrix roof because its products run on the f16 matrix instructions)
                roofline["frac_flag"] = ("frac exceeds 1 against the fp32 matrix peak: this launch issues every product the convention books, on the f16 matrix "
                                         "instructions with split operands; the roof of the path that ran is peak_split_path -- read frac_of_split_path_roof")

    # context for `roofline` (which, per contract, is about the dominant launch): every launch of one force call with an algorithmic-work
    # model over the wall time of the call
    if roofline is not None:
        try:
            b = roofline["bound"]
            tot = sum(algo[t][1] * kernels[t]["launches_per_step"] for t in kernels if t in algo and algo[t][0] == b)
            sec = dt / steps
            if b == "mfma":
                roofline["force_call"] = {"algorithmic_flop": tot, "achieved": round(tot / sec / 1e12, 3), "unit": "TFLOP/s",
                                          "frac": round(tot / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                          "note": "algorithmic FLOP of every MFMA-bound launch of one force call / wall time of the call"}
                issued = sum(kernels[t]["measured"]["mfma_flop_issued"] * kernels[t]["launches_per_step"] for t in kernels if "mfma_flop_issued" in kernels[t].get("measured", {}))
                if issued:
                    roofline["force_call"]["issued_flop"] = issued
                    roofline["force_call"]["issued_frac"] = round(issued / sec / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
            else:
                whole = 3.0 * n_int * (E * 3100.0 + 2 * N * 4096.0)       # B_force = 3 sum_layers (message + mixing) forward bytes
                roofline["force_call"] = {"algorithmic_bytes": whole, "achieved": round(whole / sec / 1e9, 1), "unit": "GB/s",
                                          "frac": round(whole / sec / 1e9 / HBM_PEAK_GBS, 4),
                                          "note": "SURVEY.md 8(d) B_force = 3 x sum over interactions of (message + mixing) forward bytes / wall time of the call"}
            if roofline["force_call"]["frac"] > 1.0:
                roofline["force_call"]["frac_flag"] = FLAG
        except Exception:  # pragma: no cover
            pass

    # ---------------- CPU baseline: the reference's own modules on the host cores, same batch, same
    # weights; the oracle restatement ("port") only where the reference is not available
    cpu = None
    if world == 1 and with_cpu and not args.no_cpu_baseline:
        # torch's intra-op pool stops scaling (and then degrades) on these small per-op sizes well before
        # the 100+ cores of a GPU host; 16 threads is the bounded, stated sample configuration
        ncores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(ncores)
        ref_model = reference_model(kind, rep_p, head_p, F, n_int, n_rbf, cutoff)
        if ref_model is not None:
            def cpu_call():
                o = ref_model(reference_inputs(batch))     # fresh tensors per call: the model writes into the dict
                return {"energy": o["energy"].detach(), "forces": o["forces"].detach()}
            kind_ = "reference"
        else:
            from oracle import spk_oracle as O          # test infrastructure: the CPU restatement, timed as the baseline
            cpu_call = lambda: O.energy_and_forces(kind, rep_p, head_p, batch, n_int)
            kind_ = "port"
        c0 = time.perf_counter()
        oc = cpu_call()
        first = time.perf_counter() - c0
        reps = max(1, min(cpu_reps, int(20.0 / max(first, 1e-3)))) if cpu_reps > 0 else 0      # bounded sample: about 20 s of CPU work at most
        ts = []
        for _ in range(reps):
            c0 = time.perf_counter()
            oc = cpu_call()
            ts.append(time.perf_counter() - c0)
        ts.sort()
        med = ts[len(ts) // 2] if ts else first        # cpu_reps == 0 (the water box: ~20 s per call): the one call made is the sample
        reps = max(reps, 1)
        df = (f_ref.cpu() - oc["forces"]).double()
        cpu = {"value": round(E * n_int / med / 1e6, 4), "unit": "M edge-messages/s", "cores": ncores, "kind": kind_,
               "sample": "same %s, median of %d force calls (%.2f s each) of %s, torch %s fp32 CPU" % (
                   "%d-frame batch" % (hi - lo) if workload == "aspirin" else "%d-atom box" % N, reps, med,
                   "the reference's NeuralNetworkPotential (PairwiseDistances + representation + Atomwise + Forces) via oracle/refshim.py" if kind_ == "reference" else "the oracle restatement",
                   torch.__version__),
               "parity_rel_forces": float(df.abs().max() / oc["forces"].abs().max()),
               "parity_rms_forces": float(df.pow(2).mean().sqrt() / oc["forces"].double().pow(2).mean().sqrt()),
               "parity_rel_energy": float((e_ref.cpu() - oc["energy"]).abs().max() / oc["energy"].abs().max())}
    res.update({"kernels": kernels, "roofline": roofline, "cpu": cpu})
    return res


def drop_in_measure(args, dev, rep_p, head_p, batch, f_hip, steps=50):
    """north_star's "callers stay untouched": the REFERENCE's own NeuralNetworkPotential / Atomwise / Forces code (from
    oracle/_ref resp. /root/reference -- the callers, not a result oracle) around the HIP classes after
    schnetpack_amd.install.install(), eager and through the opt-in ``fused_potential`` route that hands the standard potential
    to the two-launch operator.  Returns edge-messages/s of both and the agreement with the mirror model's forces."""
    from oracle import refshim
    if not refshim.available():
        return None
    import schnetpack_amd.install as inst
    ns = refshim.load()
    import numpy as np
    sys.modules["ase.data"].atomic_masses = np.ones(119)
    E = int(batch["idx_i"].shape[0])
    out = {"what": "the reference's NeuralNetworkPotential(+ its Atomwise, Forces) on cuda:0 after schnetpack_amd.install.install(): the representation, "
                   "PairwiseDistances, Dense and scatter_add underneath are the HIP classes; same batch, same weights, eager calls (the reference "
                   "model allocates its outputs per call), median of 3 repetitions of %d calls; module_by_module = install(fused_head=False, fused_potential=False), "
                   "fused_potential = install() with its defaults" % steps}
    try:
        for label, kw in (("module_by_module", {"fused_head": False, "fused_potential": False}), ("fused_potential", {"fused_head": True, "fused_potential": True})):
            inst.install(sys.modules["schnetpack"], **kw)
            try:
                spk = sys.modules["schnetpack"]
                rb, cf = spk.nn.GaussianRBF(20, 5.0), spk.nn.CosineCutoff(5.0)
                rep = sys.modules["schnetpack.representation.schnet"].SchNet(128, 3, rb, cf)
                aw = sys.modules["schnetpack.atomistic.atomwise"].Atomwise(n_in=128, output_key="energy")
                pd = sys.modules["schnetpack.atomistic.distances"].PairwiseDistances()
                m = ns.model.NeuralNetworkPotential(rep, input_modules=[pd], output_modules=[aw, ns.response.Forces()])
                m.representation.load_state_dict(rep_p)
                m.output_modules[0].load_state_dict(head_p)
                m = m.to(dev).eval()
                ri = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in reference_inputs(batch).items()}

                def call():
                    d = dict(ri)
                    d["_positions"] = ri["_positions"].detach().clone()
                    o = m(d)
                    return o["forces"].detach()
                for _ in range(5):
                    f = call()
                reps_ms = []
                for _ in range(3):          # an eager, launch-bound leg: median of three repetitions (34 % run-to-run spread seen on single ones)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        f = call()
                    torch.cuda.synchronize()
                    reps_ms.append(1e3 * (time.perf_counter() - t0) / steps)
                ms = sorted(reps_ms)[1]
                out[label] = {"ms_per_call": round(ms, 4), "M_edge_messages_per_s": round(E * 3 / ms / 1e3, 1),
                              "repetitions_ms": [round(v, 4) for v in reps_ms],
                              "model_class": type(m).__module__ + "." + type(m).__name__,
                              "rel_diff_forces_vs_mirror_model": float((f - f_hip).abs().max() / f_hip.abs().max())}
            finally:
                inst.uninstall()
    except Exception as exc:  # pragma: no cover
        out["error"] = str(exc)[:300]
    return out


def main():
    import faulthandler
    faulthandler.enable()
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (there is no CPU fallback of the product path)")
    local = local % torch.cuda.device_count()      # (several ranks may share a device in the gloo self-test)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # SPK_BENCH_FORCE_DIST=1: build the process group even at world size 1 (under a launcher) -- the RCCL branches (barriers, max-over-ranks
    # reductions, the flat gradient all-reduce, the bead all-gather) then execute on a one-GPU box (tests/test_gpu_rccl.py)
    if world > 1 or (os.environ.get("SPK_BENCH_FORCE_DIST") == "1" and "MASTER_PORT" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  SPK_BENCH_BACKEND=gloo lets two ranks share ONE device so that the multi-rank
        # control flow (sharding, barriers, max-over-ranks timing) can be exercised on a single-GPU box.
        backend = os.environ.get("SPK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from schnetpack_amd import _lib, model as M, synthetic as S

    _lib.set_variant({"auto": _lib.VARIANT_AUTO, "simple": _lib.VARIANT_SIMPLE, "mfma": _lib.VARIANT_MFMA,
                      "directed": _lib.VARIANT_MFMA_DIRECTED, "pair": _lib.VARIANT_MFMA_PAIR, "mol": _lib.VARIANT_MFMA_MOL}[args.variant])
    if os.environ.get("SPK_CHAIN_ROWS"):      # tuning hook: force the row-tile height of the fused Dense chains
        _lib.lib().spk_chain_set_rows(int(os.environ["SPK_CHAIN_ROWS"]))
    n_int, F, n_rbf, cutoff = 3, 128, 20, 5.0

    def make_model(kind):
        # random-init weights of the named architecture (seeded; the package's own initialisation, which follows the
        # reference's: xavier_uniform Dense weights, zero biases, N(0, 1) embedding).  Host copies with the reference's
        # state_dict keys are kept for the cpu_baseline leg -- the only place the oracle is imported.
        torch.manual_seed(0)
        m = M.build_model(kind, F, n_int, n_rbf, cutoff)
        rp = {k: v.detach().clone() for k, v in m.representation.state_dict().items()}
        hp = {k: v.detach().clone() for k, v in m.output_modules[0].state_dict().items()}
        return m.to(dev).eval(), rp, hp

    model, rep_p, head_p = make_model(args.kind)

    if args.mode == "train":
        line = train_measure(args, args.kind, rank, world, dev, dist, model, rep_p, head_p, args.steps, args.warmup, with_pmc=True)
        if rank == 0 and line is not None:
            emit(line, args.detail)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.mode == "md":
        return md_main(args, rank, world, dev, dist, model)

    r = eval_leg(args, args.kind, args.workload, model, rep_p, head_p, rank, world, dev, dist, args.steps, args.warmup, cpu_reps=args.cpu_reps)
    if r is None:       # --pmc-child
        return
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    value, dt, E, N, batch, inp = r["value"], r["dt"], r["E"], r["N"], r["batch"], r["inp"]
    kernels, roofline, cpu = r["kernels"], r["roofline"], r["cpu"]
    default_line = world == 1 and args.kind == "schnet" and args.workload == "aspirin"

    # ---------------- scatter_add op alone (north_star: HBM roofline of the segmented sum) and the measured copy
    # bandwidth of this box beside it (fraction of nominal AND of measured copy bandwidth).
    # Both are timed as 50 launches inside ONE HIP graph (no host launch gaps), with events around the replay.
    idx = inp["_idx_i"]
    from schnetpack_amd import ops
    src = torch.empty(64 * 1024 * 1024, device=dev)     # 256 MB: beyond the Infinity Cache together with dst
    dst = torch.empty_like(src)

    def graph_time_us(fn, reps=50):
        fn(); torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg):
            for _ in range(reps):
                fn()
        gg.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gg.replay(); e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    def scatter_case(idx_, n_rows, chan, n_buf, reps, label):
        """`reps` launches of spk_scatter_add_f32 inside one HIP graph, cycling through `n_buf` input / output buffer pairs: with n_buf = 1 the
        operand stays in the 256 MB Infinity Cache between launches, with a working set far beyond it every launch streams from DRAM."""
        e_ = int(idx_.shape[0])
        rp_ = ops.segment_rowptr(idx_, n_rows)
        xb = [torch.randn(e_, chan, device=dev) for _ in range(n_buf)]
        yb = [torch.empty(n_rows, chan, device=dev) for _ in range(n_buf)]
        it = [0]

        def once():
            k = it[0] % n_buf
            it[0] += 1
            _lib.check(_lib.lib().spk_scatter_add_f32(_lib.fptr(xb[k]), _lib.iptr(idx_), _lib.iptr(rp_, torch.int32), 1, e_, chan, n_rows, _lib.fptr(yb[k]), _lib.stream()))
        us = graph_time_us(once, reps=reps)
        nbytes = 4.0 * e_ * chan + 8.0 * e_ + 4.0 * n_rows * chan
        gbs = nbytes / (us * 1e-6) / 1e9
        ws = n_buf * (4.0 * e_ * chan + 4.0 * n_rows * chan) / 1e6
        del xb, yb
        return {"what": label, "shape": [e_, chan, n_rows], "buffers": n_buf, "working_set_MB": round(ws, 1), "us": round(us, 2), "achieved": round(gbs, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}

    cp_us = graph_time_us(lambda: dst.copy_(src), reps=10)
    copy_gbs = 2.0 * src.numel() * 4 / (cp_us * 1e-6) / 1e9
    del src, dst
    scatter = {"measured_copy_GBs": round(copy_gbs, 1)}
    cases = [("cache_resident", idx, N, F, 1, 50, "configs[1] x_ij shape, ONE buffer replayed: the 43 MB operand stays in the Infinity Cache (an upper bound, not a DRAM figure)"),
             ("dram_rotating", idx, N, F, 16, 48, "configs[1] x_ij shape, 16 rotating buffer pairs (working set >> 256 MB MALL): every launch streams from DRAM")]
    if default_line and not args.no_md:
        wb_ = cached_water_box(args.water_side, 0)
        cases.append(("dram_water_dmu", wb_["idx_i"].to(dev), int(wb_["Z"].shape[0]), 3 * F, 2, 6,
                      "configs[4] per-bead PaiNN dmu shape (E x 3F floats = 2.6 GB per launch, far beyond the MALL)"))
    for name, idx_, n_rows, chan, n_buf, reps, label in cases:
        try:
            scatter[name] = scatter_case(idx_, n_rows, chan, n_buf, reps, label)
            scatter[name]["frac_of_measured_copy"] = round(scatter[name]["achieved"] / copy_gbs, 4)
        except Exception as exc:  # pragma: no cover
            scatter[name] = {"error": str(exc)[:200]}
        torch.cuda.empty_cache()
    dram = [v["frac"] for k, v in scatter.items() if k.startswith("dram_") and isinstance(v, dict) and "frac" in v]
    scatter["frac_dram"] = min(dram) if dram else None        # north_star's ">= 40 % of the HBM roofline on the scatter_add": the WORST of the DRAM-resident cases
    scatter["note"] = ("algorithmic bytes 4EC + 8E + 4NC per launch over the HIP-graph time of the launches (events around one replay); copy = 256 MB device-to-device "
                       "torch copy (read + write bytes) timed the same way; frac = achieved / 8 TB/s; frac_dram = the smaller of the DRAM-resident cases")

    # ---------------- neighbour-list rebuild on the device for this workload
    from schnetpack_amd import neighborlist as NL
    nl_cell = batch["cell"].reshape(1, 3, 3).to(dev) if args.workload == "water" else None
    nl_pbc = torch.tensor([True, True, True], device=dev) if args.workload == "water" else None
    nl_kw = dict(idx_m=inp["_idx_m"], cell=nl_cell, pbc=nl_pbc, n_systems=int(batch["n_mol"]))
    nl = NL.neighbor_list(inp["_positions"].detach(), cutoff, **nl_kw)
    t_nl = []
    for _ in range(5):
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        NL.neighbor_list(inp["_positions"].detach(), cutoff, **nl_kw)
        torch.cuda.synchronize()
        t_nl.append(time.perf_counter() - c0)
    t_nl.sort()
    # bit-exact comparison with the list the workload was built with (north_star: "neighbour indices bit-exact"): both lists in the
    # canonical order (idx_i, idx_j, image shift), indices and integer image shifts compared with torch.equal -- not a pair count
    def _canonical(ii, jj, off):
        if nl_cell is not None:
            sh = torch.round(off.double() @ torch.linalg.inv(nl_cell[0].double())).to(torch.int64)
        else:
            sh = torch.zeros(ii.shape[0], 3, dtype=torch.int64, device=ii.device)
        key = ((ii.to(torch.int64) * int(inp["_positions"].shape[0]) + jj.to(torch.int64)) * 27 + (sh[:, 0] + 1) * 9 + (sh[:, 1] + 1) * 3 + (sh[:, 2] + 1))
        order = torch.argsort(key, stable=True)
        return ii[order].to(torch.int64), jj[order].to(torch.int64), sh[order]
    try:
        nl_full = NL.neighbor_list(inp["_positions"].detach(), cutoff, **nl_kw)
        ca = _canonical(nl_full["_idx_i"], nl_full["_idx_j"], nl_full["_offsets"])
        cb = _canonical(inp["_idx_i"], inp["_idx_j"], inp["_offsets"])
        nl_exact = bool(ca[0].shape == cb[0].shape and all(torch.equal(x, y) for x, y in zip(ca, cb)))
    except Exception as exc:  # pragma: no cover
        nl_exact = "error: %s" % str(exc)[:120]
    nbl = {"pairs": int(nl["_idx_i"].shape[0]), "matches_input_list": nl_exact, "pair_count_matches": int(nl["_idx_i"].shape[0]) == E,
           "build_ms": round(1e3 * t_nl[len(t_nl) // 2], 4), "M_pairs_per_s": round(E / t_nl[len(t_nl) // 2] / 1e6, 1),
           "note": "count + fill incl. allocation and the one D2H of the pair count (wall clock, median of 5)"}
    if cpu is not None and args.workload == "aspirin":
        # the reference's per-molecule TorchNeighborList loop, restated (oracle/nbl_oracle.py), same batch
        from oracle import nbl_oracle as NB
        c0 = time.perf_counter()
        oi, _, _, _ = NB.batch_neighbor_list(batch["R"], batch["idx_m"], None, None, cutoff)
        nbl["cpu_oracle_ms"] = round(1e3 * (time.perf_counter() - c0), 2)
        nbl["cpu_oracle_pairs"] = int(oi.shape[0])

    # ---------------- configs[2]: the same 256-frame batch through PaiNN (default line only)
    painn = None
    painn_model = None
    if default_line and not args.no_painn:
        try:
            painn_model, p_rep, p_head = make_model("painn")
            pr = eval_leg(args, "painn", "aspirin", painn_model, p_rep, p_head, 0, 1, dev, None, min(args.steps, 100), min(args.warmup, 10), cpu_reps=3)
            painn = {"metric": "M edge-messages/s (eval force call, MD17-aspirin 256-frame batch, PaiNN)", "value": round(pr["value"], 2),
                     "unit": "M edge-messages/s", "ms_per_step": round(1e3 * pr["dt"] / pr["steps"], 4), "steps": pr["steps"], "hip_graph": pr["graph"],
                     "config": {"workload": "configs[2]: MD17 aspirin x %d frames, PaiNN(n_atom_basis=128, n_interactions=3, n_rbf=20, cutoff=5.0) + Atomwise + Forces; "
                                            "N=%d atoms, E=%d directed edges" % (pr["frames"], pr["N"], pr["E"])},
                     "roofline": pr["roofline"], "cpu_baseline": pr["cpu"], "kernels": pr["kernels"]}
        except Exception as exc:  # pragma: no cover
            painn = {"error": str(exc)[:300]}

    # ---------------- configs[4] per-GPU share: the eval force call on the 32k-atom bulk-water PBC box, both models (default line only):
    # roofline of the dominant kernel with this run's counters, traffic over the perfect-reuse bound, ONE force call of the
    # reference on the host cores as the baseline (PaiNN only -- a call takes ~20 s; SchNet's: profiles/)
    water = None
    if default_line and not args.no_md:
        water = {}
        for k in ("painn", "schnet"):
            try:
                if k == "painn":
                    if painn_model is None:
                        painn_model, p_rep, p_head = make_model("painn")
                    wm, w_rep, w_head = painn_model, p_rep, p_head
                else:
                    wm, w_rep, w_head = model, rep_p, head_p
                wr = eval_leg(args, k, "water", wm, w_rep, w_head, 0, 1, dev, None, 10, 2, with_pmc=True, with_cpu=True, cpu_reps=0)
                water[k] = {"metric": "M edge-messages/s (eval force call, 32k-atom bulk-water PBC box, %s)" % ("PaiNN" if k == "painn" else "SchNet"),
                            "value": round(wr["value"], 2), "unit": "M edge-messages/s", "ms_per_step": round(1e3 * wr["dt"] / wr["steps"], 4), "steps": wr["steps"],
                            "hip_graph": wr["graph"], "n_atoms": wr["N"], "n_edges": wr["E"], "ns_per_day_at_0.5fs_per_call": round(wr["steps"] / wr["dt"] * 0.5 * 86400e-6, 3),
                            "roofline": wr["roofline"], "cpu_baseline": wr["cpu"], "kernels": wr["kernels"]}
            except Exception as exc:  # pragma: no cover
                water[k] = {"error": str(exc)[:300]}
        torch.cuda.empty_cache()

    # ---------------- configs[3] per-GPU share: one AdamW step of the force-matching loss, PaiNN and SchNet
    train = None
    if default_line and not args.no_train:
        train = {}
        for k in ("painn", "schnet"):
            try:
                tm, t_rep, t_head = make_model(k)
                tl = train_measure(args, k, 0, 1, dev, None, tm, t_rep, t_head, steps=100, warmup=8, with_pmc=True)
                train[k] = {kk: tl[kk] for kk in ("metric", "value", "unit", "ms_per_step", "steps", "config", "cpu_baseline", "launches_per_step", "roofline")}
                del tm
            except Exception as exc:  # pragma: no cover
                train[k] = {"error": str(exc)[:300]}
        torch.cuda.empty_cache()

    # ---------------- the other half of BASELINE's metric: MD ns/day (on-device NVE loop, rows f1-f3), aspirin x 256 and the
    # 32k-atom water box, same model kind; configs[4] as stated: PaiNN, 8 beads, RPMD 0.2 fs + PILE-L NVT on the full box
    md = None
    if world == 1 and not args.no_md:
        md = {}
        keys = ("ns_per_day", "ms_per_step", "n_atoms", "pairs_in_list", "trajectories", "rebuilds", "fraction_in_rebuilds", "dt_fs", "steps", "hip_graph")
        for wl in ("aspirin", "water"):
            try:
                rr = md_run(args, model, dev, wl, 0, 1, None, steps=args.md_steps if wl == "aspirin" else max(args.md_steps // 4, 20), warmup=10)
                md[wl] = {k: rr[k] for k in keys}
            except Exception as exc:  # pragma: no cover
                md[wl] = {"error": str(exc)[:200]}
        md["note"] = ("NVE velocity Verlet, 0.5 fs, one HIP-graph replay per step; ns/day per trajectory.  Water box: device neighbour list with a "
                      "%.1f A skin.  Aspirin batch: isolated molecules of <= 28 atoms keep the COMPLETE intramolecular list (an infinite skin: no "
                      "rebuilds, no host synchronisation; pairs beyond the cutoff get no tile in the kernels)" % args.md_shell)
        if default_line and not args.no_pimd:
            try:
                if painn_model is None:
                    painn_model, _, _ = make_model("painn")
                torch.cuda.empty_cache()
                rr = md_run(args, painn_model, dev, "water", 0, 1, None, steps=30, warmup=6, beads=8, thermostat="pile")
                md["water_pimd"] = {k: rr[k] for k in keys + ("beads", "thermostat", "ms_per_rebuild", "kinetic_temperature_K", "collectives_per_step")}
                md["water_pimd"]["what"] = ("configs[4] on ONE GPU: 31 944-atom bulk-water PBC box, PaiNN(128, 3, 20, 5.0), ring-polymer MD with 8 beads folded into "
                                            "the batch (8 x 31 944 atoms per force call), 0.2 fs, PILE-L thermostat (300 K, tau = 100 fs) at step begin and end = NVT; "
                                            "units A / Dalton / ps; ns/day of the ring polymer")
            except Exception as exc:  # pragma: no cover
                md["water_pimd"] = {"error": str(exc)[:300]}
    sweep = None
    if world == 1 and not args.no_sweep:
        try:
            sweep = sweep_measure(model, dev, args.kind)
            if default_line:        # PaiNN rows beside the SchNet ones (symmetric k = 16 / 32 / 64, asymmetric k = 32)
                if painn_model is None:
                    painn_model, _, _ = make_model("painn")
                sweep["rows"] += sweep_measure(painn_model, dev, "painn")["rows"]
                sweep["kind"] = "schnet+painn"
        except Exception as exc:  # pragma: no cover
            sweep = {"error": str(exc)[:200]}
    # ---------------- experiment, default OFF in the product: tabulated SchNet filters (schnetpack_amd/tabulate.py) beside the fp32-MFMA
    # contract path on the 32k-atom water box -- time of the eval force call and agreement of the forces
    experiments = None
    if default_line and not args.no_md:
        try:
            from schnetpack_amd import tabulate
            wb = cached_water_box(args.water_side, 0)
            winp = M.batch_to_inputs(wb, dev)

            def wcall():
                return model(dict(winp))["forces"].detach()

            def wtime(reps=8):
                for _ in range(2):
                    wcall()
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                for _ in range(reps):
                    wcall()
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - c0) / reps
            f_c = wcall().clone()
            ms_c = wtime()
            tabulate.tabulate_filters(model.representation, 512)
            try:
                f_t = wcall().clone()
                ms_t = wtime()
            finally:
                tabulate.clear_filter_tables()
            Ew = int(wb["idx_i"].shape[0])
            experiments = {"tabulated_filters": {
                "what": "EXPERIMENT (opt-in, eval only): W_l(d) f_c(d) from 512-knot cubic-Hermite tables (value + slope, float64 build) instead of the filter "
                        "network; eval force call of SchNet(128, 3, 20, 5.0) on the %d-atom water box (E = %d), eager launches" % (int(wb["Z"].shape[0]), Ew),
                "contract_path_ms": round(ms_c, 4), "tabulated_ms": round(ms_t, 4),
                "contract_M_edge_messages_per_s": round(Ew * 3 / ms_c / 1e3, 1), "tabulated_M_edge_messages_per_s": round(Ew * 3 / ms_t / 1e3, 1),
                "forces_rel_diff_tabulated_vs_contract": float((f_t - f_c).abs().max() / f_c.abs().max()),
                "note": "the contract (fp32 MFMA) path is what every headline number of this line runs; table error: value ~4e-8, slope ~1.6e-6 of max |dW/dd| at 512 knots (2e-7 at 1024)"}}
        except Exception as exc:  # pragma: no cover
            experiments = {"tabulated_filters": {"error": str(exc)[:300]}}
    cliff = None
    if default_line and not args.no_sweep:
        try:
            if painn_model is None:
                painn_model, _, _ = make_model("painn")
            cliff = cliff_measure([("schnet", model), ("painn", painn_model)], dev)
        except Exception as exc:  # pragma: no cover
            cliff = {"error": str(exc)[:200]}
    drop_in = None
    if default_line and not args.no_drop_in and not args.no_cpu_baseline:
        drop_in = drop_in_measure(args, dev, rep_p, head_p, batch, r["f_ref"])

    info = _lib.device_info()
    hi_lo = r["frames"]
    line = {
        "metric": "M edge-messages/s (eval force call, %s, %s)" % ("MD17-aspirin 256-frame batch" if args.workload == "aspirin" else "32k-atom bulk-water PBC box", "SchNet" if args.kind == "schnet" else "PaiNN"),
        "value": round(value, 2), "value_without_ramp": round(r["value_without_ramp"], 2) if world == 1 else None,
        "unit": "M edge-messages/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("configs[1]: MD17 aspirin x %d frames/GPU, %s(128,3,20,5.0)+Atomwise+Forces eval force call" % (hi_lo, "SchNet" if args.kind == "schnet" else "PaiNN"))
                               if args.workload == "aspirin" else
                               ("configs[4] per-GPU share: 32k-atom bulk-water PBC box, %s(128,3,20,5.0)+Atomwise+Forces eval force call" % ("SchNet" if args.kind == "schnet" else "PaiNN")),
                   "workload_detail": ("n_atom_basis=128, n_interactions=3, n_rbf=20 (Gaussian), cosine cutoff 5.0 A; N=%d atoms, E=%d directed edges per GPU; one replica / batch per GPU; "
                                       "ns/day at 0.5 fs per force call = %.3f" % (N, E, args.steps / dt * 0.5 * 86400e-6)),
                   "n_atoms": N, "n_edges": E, "frames_per_s": round(hi_lo * world * args.steps / dt, 1),
                   "parallelism": "frames sharded over %d rank(s), no data-path collective" % world,
                   "world_size": world, "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist is not None else None,
                   "hip_graph": r["graph"], "variant": args.variant, "compute_units": info["compute_units"],
                   "preconditioning": "%d untimed replays (%.2f s clock ramp) before the %d warm-up steps; the timed region is exactly %d steps; "
                                      "value_without_ramp = the same %d steps timed once before the ramp (this rank)" % (r["n_ramp"], RAMP_S, args.warmup, r["steps"], r["steps"])},
        "roofline": roofline, "cpu_baseline": cpu, "painn": painn, "water": water, "train": train, "md": md, "sweep": sweep, "drop_in": drop_in, "experiments": experiments,
        "kernels": kernels, "scatter_add": scatter, "neighbor_list": nbl, "molecule_cliff": cliff,
    }
    line["config"]["ramp_s"] = RAMP_S
    line["config"]["split_precision_matrix_path"] = bool(_lib.get_split())
    if isinstance(water, dict):
        for k_ in ("painn", "schnet"):
            if isinstance(water.get(k_), dict) and "n_atoms" in water[k_]:
                line["config"]["water_atoms"], line["config"]["water_pairs"] = water[k_]["n_atoms"], water[k_]["n_edges"]      # 22^3 x 3 = 31 944 atoms (the survey's 32 001 is another lattice)
                break
    line["parity_ledger_worst"] = parity_ledger_worst()
    line["config"]["multi_gpu_measured"] = world > 1      # N > 1 has never run on hardware from this repo: the driver's SCALE run is the measurement
    emit(line, args.detail)
    if dist is not None:
        dist.destroy_process_group()


def md_run(args, model, dev, workload, rank, world, dist, steps, warmup, beads=None, thermostat=None, bead_parallel=None):
    """Molecular dynamics, the whole step on the device: fused kick + drift + skin check,
    device neighbour list with a skin (rebuilt when an atom moved more than half of it), HIP-graph force call, kick.
    ``beads`` > 1: ring-polymer MD (0.2 fs, md_configs/dynamics/integrator/rpmd.yaml) with the beads folded into the batch;
    ``thermostat`` "pile": PILE-L at 300 K, tau = 100 fs at step begin and end (NVT; md_configs/dynamics/thermostat/
    pile_local.yaml) in the unit system A / Dalton / ps (energy unit Da A^2 / ps^2 = 0.01 kJ/mol).  Without ``bead_parallel``
    every rank integrates its own batch / box (replicas only, no collective); with it ONE ring polymer is spread over the ranks
    (beads / world each; exchange scheme "state" or "forces", md.RPMDSimulation).  ns/day = steps/s x dt x 86400e-6 per
    trajectory; random-init weights give an arbitrary but smooth potential (NVE runs start from zero momenta).
    Returns the measurements of this rank (time = max over ranks)."""
    from schnetpack_amd import model as M, synthetic as S
    from schnetpack_amd import md as MD
    beads = args.beads if beads is None else beads
    thermostat = args.thermostat if thermostat is None else thermostat
    bead_parallel = args.bead_parallel if bead_parallel is None else bead_parallel
    if thermostat == "auto":
        thermostat = "pile" if beads > 1 else "none"
    if thermostat == "pile" and beads <= 1:
        raise SystemExit("bench.py: the PILE-L thermostat belongs to ring-polymer MD (--beads B > 1)")
    if bead_parallel and (beads <= 1 or dist is None or beads % world):
        raise SystemExit("bench.py --bead-parallel: needs --beads B > 1 divisible by --gpus N and a process group (N > 1, or SPK_BENCH_FORCE_DIST=1 under a launcher)")
    dt_fs = 0.5 if beads <= 1 else 0.2     # md.yaml resp. rpmd.yaml of the reference
    shared = bool(bead_parallel)            # bead-parallel: every rank starts from the SAME system
    if workload == "water":
        batch = cached_water_box(args.water_side, 0 if shared else rank)
        n_traj = 1
    else:
        batch = S.molecule_batch("aspirin", args.frames, seed=1000 + rank if (world > 1 and not shared) else 0)
        n_traj = args.frames
    inp = M.batch_to_inputs(batch, dev)
    N = int(batch["Z"].shape[0])
    inp["_n_atoms"] = torch.bincount(batch["idx_m"], minlength=int(batch["n_mol"])).to(dev)
    if workload == "water":
        inp["_cell"] = batch["cell"].reshape(1, 3, 3).to(dev)
        inp["_pbc"] = torch.tensor([True, True, True], device=dev)
    masses = torch.where(batch["Z"] == 1, 1.008, torch.where(batch["Z"] == 6, 12.011, 15.999)).to(dev)
    KB = 100.0 * MD.KB_MD                  # Boltzmann's constant in Da A^2 / ps^2 / K (the positions are in A)
    T_bath = 300.0
    th = None
    if beads > 1:
        if thermostat == "pile":
            th = MD.PILELocalThermostat(T_bath, 100.0, seed=1234, kb=KB)          # tau = 100 fs, like pile_local.yaml
            time_step = dt_fs * MD.FS_MD     # real units: 0.2 fs in ps; omega = kB n T / hbar (1 / ps)
            omega = None
        else:
            time_step, omega = 0.02, 3.0     # NVE ring polymer in model units (round-1/2 lines)
        sim = MD.RPMDSimulation(model, inp, masses, time_step, beads, cutoff=5.0, temperature=T_bath, omega=omega, cutoff_shell=args.md_shell,
                                use_graph=not args.no_graph, thermostat=th, group=(dist.group.WORLD if bead_parallel else None),
                                exchange=bead_parallel or "state")
    else:
        # model energy unit := eV-like; time step chosen so that atoms move ~1e-3 A per step
        sim = MD.NVESimulation(model, inp, masses, 0.02, cutoff=5.0, cutoff_shell=args.md_shell, use_graph=not args.no_graph)
    sim.step(max(warmup, 2))
    e0 = sim.total_energy()
    b0 = sim.nl.n_builds
    tr0 = sim.t_rebuild
    c0 = getattr(sim, "n_collectives", 0)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.step(steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else torch.device("cpu"), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    steps_s = steps / dt
    n_loc = getattr(sim, "n_local", 1)
    ke = float(sim.kinetic_energy())
    res = {"ns_per_day": round(steps_s * dt_fs * 86400e-6, 4), "ms_per_step": round(1e3 * dt / steps, 4), "steps_per_s": steps_s, "dt_fs": dt_fs,
           "n_atoms": N, "pairs_in_list": int(sim._lists["_idx_i"].shape[0]), "trajectories": n_traj, "steps": steps, "seconds": dt,
           "rebuilds": sim.nl.n_builds - b0, "ms_per_rebuild": round(1e3 * (sim.t_rebuild - tr0) / max(sim.nl.n_builds - b0, 1), 3),
           "fraction_in_rebuilds": round((sim.t_rebuild - tr0) / dt, 4), "graph_captures": sim.n_captures, "hip_graph": sim.graph is not None,
           "beads": beads, "beads_per_rank": n_loc if beads > 1 else 1, "thermostat": ("PILE-L 300 K tau=100 fs" if th is not None else None),
           "collectives_per_step": (getattr(sim, "n_collectives", 0) - c0) / steps,
           "energy_drift": abs(sim.total_energy() - e0) / max(ke, 1e-12) / steps}
    if th is not None:      # ring-polymer momenta equilibrate at n_beads x T (thermostats_rpmd.py:93-100)
        res["kinetic_temperature_K"] = round(2.0 * ke / (3 * N * n_loc * KB) / beads, 2)
    return res


def md_main(args, rank, world, dev, dist, model):
    """`--mode md`: one line for the on-device MD loop of md_run()."""
    n_int = 3
    r = md_run(args, model, dev, args.workload, rank, world, dist, args.steps, args.warmup)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    kind = "SchNet" if args.kind == "schnet" else "PaiNN"
    n_traj, N, E_list = r["trajectories"], r["n_atoms"], r["pairs_in_list"]
    bp = args.bead_parallel
    nve = r["thermostat"] is None
    if args.beads <= 1:
        metric = "MD ns/day per trajectory (NVE, 0.5 fs, %s, %s)"
    else:
        metric = "RPMD ns/day per ring polymer (" + str(args.beads) + " beads, 0.2 fs, " + ("NVE" if nve else "PILE-L NVT 300 K") + ", %s, %s)"
    if bp:
        par = ("bead-parallel: ONE ring polymer of %d beads over %d rank(s), %d bead(s) per rank, exchange scheme '%s': %.0f all-gather(s) per step"
               % (args.beads, world, r["beads_per_rank"], bp, r["collectives_per_step"]))
    else:
        par = "replicas only: %d independent rank(s), no collective" % world
    line = {
        "metric": metric % ("MD17-aspirin x %d replicas" % n_traj if args.workload == "aspirin" else "32k-atom bulk-water PBC box", kind),
        "value": r["ns_per_day"], "unit": "ns/day", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong" if bp else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("%s: %s(128, 3, 20, 5.0) + Atomwise + Forces, %s, device neighbour list with a %.1f A skin (%d pairs in the list of this rank), "
                                "%d trajectories per GPU advanced together; N=%d atoms per bead"
                                % ("configs[1]-style MD17 aspirin batch" if args.workload == "aspirin" else "configs[4] (32k-atom bulk water)",
                                   kind, "velocity Verlet" if args.beads <= 1 else "ring-polymer integrator, %d beads (%d folded into the batch of a rank)" % (args.beads, r["beads_per_rank"]),
                                   args.md_shell, E_list, n_traj, N)),
                   "trajectories_per_gpu": n_traj, "aggregate_ns_per_day": round(r["ns_per_day"] * n_traj * (1 if bp else world), 3),
                   "M_edge_messages_per_s_in_list": round(E_list * n_int * r["steps_per_s"] * world / 1e6, 1),
                   "neighbor_list_rebuilds_in_timed_region": r["rebuilds"],
                   "ms_per_rebuild_incl_recapture": r["ms_per_rebuild"],
                   "fraction_of_time_in_rebuilds": r["fraction_in_rebuilds"], "graph_captures": r["graph_captures"],
                   "hip_graph": r["hip_graph"], "beads": r["beads"], "beads_per_rank": r["beads_per_rank"], "thermostat": r["thermostat"],
                   "kinetic_temperature_K": r.get("kinetic_temperature_K"),
                   "collectives_per_step": r["collectives_per_step"],
                   "energy_drift_per_step_rel_to_kinetic": r["energy_drift"] if nve else None,
                   "world_size": world, "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist is not None else None,
                   "parallelism": par},
        "roofline": None, "cpu_baseline": None,
    }
    emit(line, args.detail)
    if dist is not None:
        dist.destroy_process_group()


def train_measure(args, kind, rank, world, dev, dist, model, rep_p, head_p, steps, warmup, with_pmc=False):
    """configs[3] (SURVEY.md section 8 cfg 4): rMD17-aspirin training step, PaiNN/SchNet in train() mode
    (Forces with create_graph=True -> double backward through the differentiable HIP primitives), loss
    0.01 MSE(E) + 0.99 MSE(F), AdamW(lr 1e-3), 8 frames per GPU, ONE all-reduce of one flat gradient
    bucket per step (RCCL; bucket views, no copy kernels).  Returns the JSON line (dict) on rank 0, None elsewhere."""
    from schnetpack_amd import _lib, synthetic as S
    from schnetpack_amd.train import GraphedTrainStep
    n_int, F, n_rbf, cutoff = 3, 128, 20, 5.0
    pool = []
    for k in range(8):                      # 8 different resident mini-batches, cycled
        b = S.molecule_batch("aspirin", args.train_frames, seed=5000 + 97 * rank + k)
        g = torch.Generator().manual_seed(k + 31 * rank)
        bd = {kk: (v.to(dev) if torch.is_tensor(v) else v) for kk, v in b.items()}
        pool.append((b, bd, torch.randn(args.train_frames, generator=g).to(dev),
                     torch.randn(b["Z"].shape[0], 3, generator=g).to(dev)))
    n_atoms = int(pool[0][0]["Z"].shape[0])
    emax = 64 * (max(int(p[0]["idx_i"].shape[0]) for p in pool) // 64 + 2)      # static capacity of the pair list
    tstep = GraphedTrainStep(model, n_atoms, args.train_frames, emax, 5.0, lr=1e-3,
                             group=(dist.group.WORLD if dist is not None else None), use_graph=not args.no_graph)
    reducer = tstep.reducer

    # resident batches in the packed layout of the step's two static buffers (what a collate worker hands over): two copies per load
    packed = [tstep.pack(p[0], p[2], p[3], device=dev) for p in pool]

    def step(i):
        tstep.load_packed(*packed[i % len(packed)])
        return tstep.step()

    if args.pmc_child:          # wrapped by `rocprofv3 --pmc` from collect_train_pmc(): PMC_TRAIN_STEPS eager steps, no output
        tstep.use_graph = False
        for i in range(PMC_TRAIN_STEPS):
            step(i)
        torch.cuda.synchronize()
        return None

    # launches of one step: the HIP-event profile scopes of the library see its own launches; the step's total (incl. the framework's
    # element-wise / concatenation kernels) is the kernel count of the captured graph -- counted by an eager step under the profiler hooks
    losses = [float(step(i).detach()) for i in range(max(warmup, 4))]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    coll0 = reducer.collectives
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    coll_timed = reducer.collectives - coll0          # all-reduces of the flat bucket issued inside the timed region (parallel.FlatGradAllReduce._reduce)
    coll_ptr_ok = reducer.last_reduced_ptr is not None and reducer.last_reduced_ptr == reducer.flat.data_ptr()
    tstep.check()
    if dist is not None:
        tt = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else torch.device("cpu"), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return None
    # launch count of one step (library + framework kernels): torch's kineto profile of one eager step
    launches = None
    try:
        from torch.profiler import profile, ProfilerActivity
        tstep_use_graph = tstep.use_graph
        tstep.use_graph = False
        step(0)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as pr:
            step(1)
            torch.cuda.synchronize()
        launches = sum(1 for ev in pr.events() if str(getattr(ev, "device_type", "")).endswith("CUDA") and not ev.name.lower().startswith(("memcpy", "memset")))
        tstep.use_graph = tstep_use_graph
    except Exception as exc:  # pragma: no cover
        sys.stderr.write("[bench] launch count unavailable (%s)\n" % exc)
    ms_step = 1e3 * dt / steps
    # training = forward + recorded first-order backward (B_force) + the backward of that pass (about 2x the force call):
    # booked as 3 x B_force of the per-GPU batch; the step is launch-latency bound at this size, which the fraction shows
    b0 = pool[0][0]
    E0, N0 = int(b0["idx_i"].shape[0]), n_atoms
    if kind == "painn":
        work, bound, peak, unit = 3.0 * 3.0 * n_int * (E0 * 3100.0 + 2 * N0 * 4096.0), "hbm", HBM_PEAK_GBS, "GB/s"
        ach = work / (ms_step * 1e-3) / 1e9
    else:
        work, bound, peak, unit = 3.0 * 3.0 * n_int * (2.0 * E0 * (n_rbf * F + F * F) + 2.0 * N0 * 3 * F * F), "mfma", MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
        ach = work / (ms_step * 1e-3) / 1e12
    roofline = {"kernel": "whole training step (%s launches, one HIP-graph replay)" % (launches if launches is not None else "n/a"), "bound": bound,
                "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 5), "traffic": None,
                "algorithmic_per_step": work,
                "note": "algorithmic work of a force-matching step booked as 3 x the force call of its batch (forward + recorded backward + the backward of "
                        "both, SURVEY.md 8(d) / Appendix B) over the wall time of the step; at 8 frames per GPU the step is launch-latency bound"}
    if world == 1 and with_pmc and not args.no_pmc:      # (the `--mode train` line; the default line's embedded training legs skip the two profiler passes)
        tp = collect_train_pmc(args, kind)
        if tp is not None and "read_bytes" in tp and "write_bytes" in tp:
            roofline["traffic"] = tp["read_bytes"] + tp["write_bytes"]
            roofline["traffic_detail"] = dict(tp, source="measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) summed over every "
                                              "dispatch of %d eager steps of this configuration / %d; KiB per dispatch, FETCH_SIZE x 2 (gfx950); includes the moments and "
                                              "parameters of the optimizer, the saved activations of all four passes and the framework's copies" % (PMC_TRAIN_STEPS, PMC_TRAIN_STEPS))
            roofline["traffic_frac_of_hbm_peak"] = round(roofline["traffic"] / (ms_step * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        ncores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(ncores)
        b, _, Et, Ft = pool[0]
        Et, Ft = Et.cpu(), Ft.cpu()
        ref_model = reference_model(kind, rep_p, head_p, F, n_int, n_rbf, cutoff)
        if ref_model is not None:
            # the REFERENCE's modules in train() mode: Forces(create_graph=True) -> double backward, torch AdamW (atomistic/response.py:59-68)
            ref_model.train()
            copt = torch.optim.AdamW(ref_model.parameters(), lr=1e-3)

            def cpu_step():
                copt.zero_grad()
                o = ref_model(reference_inputs(b))
                l = 0.01 * ((o["energy"] - Et) ** 2).mean() + 0.99 * ((o["forces"] - Ft) ** 2).mean()
                l.backward()
                copt.step()
                return float(l.detach())
            kind_ = "reference"
        else:
            from oracle import spk_oracle as O          # test infrastructure: the CPU restatement, timed as the baseline
            rp = {k: (v.clone().requires_grad_(True) if torch.is_tensor(v) and v.is_floating_point() and k.endswith(("weight", "bias")) else v) for k, v in rep_p.items()}
            hp = {k: v.clone().requires_grad_(True) for k, v in head_p.items()}
            leaves = [v for v in list(rp.values()) + list(hp.values()) if torch.is_tensor(v) and v.requires_grad]
            copt = torch.optim.AdamW(leaves, lr=1e-3)

            def cpu_step():
                copt.zero_grad()
                R = b["R"].clone().requires_grad_(True)
                r_ij = O.pairwise_vectors(R, b["idx_i"], b["idx_j"], b["offsets"])
                if kind == "schnet":
                    x = O.schnet_representation(b["Z"], r_ij, b["idx_i"], b["idx_j"], rp, n_int)
                else:
                    x, _ = O.painn_representation(b["Z"], r_ij, b["idx_i"], b["idx_j"], rp, n_int)
                E = O.atomwise_energy(x, b["idx_m"], args.train_frames, hp)
                (dEdR,) = torch.autograd.grad([E.sum()], [R], create_graph=True)
                l = 0.01 * ((E - Et) ** 2).mean() + 0.99 * ((-dEdR - Ft) ** 2).mean()
                l.backward()
                copt.step()
                return float(l.detach())
            kind_ = "port"
        first_cpu_loss = cpu_step()
        ts = []
        for _ in range(min(args.cpu_reps, 10)):
            c0 = time.perf_counter()
            cpu_step()
            ts.append(time.perf_counter() - c0)
        ts.sort()
        cpu = {"value": round(args.train_frames / ts[len(ts) // 2], 2), "unit": "samples/s", "cores": ncores, "kind": kind_,
               "sample": "the same %d-frame AdamW step of the force-matching loss on %s (torch CPU autograd, fp32), median of %d; its first loss %.6f vs %.6f here "
                         "(same weights, same first batch)" % (args.train_frames, "the reference's NeuralNetworkPotential in train() mode" if kind_ == "reference" else "the oracle restatement",
                                                             len(ts), first_cpu_loss, losses[0]),
               "first_loss_rel_diff": abs(first_cpu_loss - losses[0]) / max(abs(first_cpu_loss), 1e-30)}
    kname = "SchNet" if kind == "schnet" else "PaiNN"
    return {
        "metric": "training samples/s (rMD17-aspirin force-matching step, %s)" % kname,
        "value": round(args.train_frames * world * steps / dt, 2), "unit": "samples/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(ms_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[3]: rMD17 aspirin training, %d frames per GPU (global batch %d), %s(128, 3, 20, 5.0) + Atomwise + "
                               "Forces(create_graph), loss 0.01 MSE(E) + 0.99 MSE(F), AdamW lr 1e-3, one flat-bucket all-reduce of %d floats per step; "
                               "static shapes (pair list padded to %d) replayed as HIP graphs: %s"
                               % (args.train_frames, args.train_frames * world, kname, reducer.numel, emax, tstep.g_bwd is not None),
                   "parallelism": "dp%d" % world, "first_loss": losses[0], "last_loss": float(loss.detach()),
                   "world_size": world, "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist is not None else None,
                   "allreduce_between_graphs": tstep.g_opt is not None, "multi_gpu_measured": world > 1,
                   "allreduce_calls_timed": coll_timed, "allreduce_floats": reducer.last_reduced_numel, "allreduce_buffer_is_flat_bucket": bool(coll_ptr_ok)},
        "launches_per_step": launches,
        "roofline": roofline, "cpu_baseline": cpu,
    }


if __name__ == "__main__":
    main()
