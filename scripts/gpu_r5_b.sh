#!/bin/bash
# round 5, second GPU call: asymmetric-list PaiNN backward (TS + GEOM row passes), parity hardening tests, sweep rows
set -x
mkdir -p gpurun_out/r5b
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "asymmetric or painn_message" 2>&1 | tail -15 > gpurun_out/r5b/pytest_ops.log
cat gpurun_out/r5b/pytest_ops.log
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "asymmetric or trained or live_reference" 2>&1 | tail -15 > gpurun_out/r5b/pytest_models.log
cat gpurun_out/r5b/pytest_models.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-md --no-painn --no-train --no-drop-in --no-pmc --no-cpu-baseline --detail gpurun_out/r5b/sweep_detail.json > gpurun_out/r5b/sweep_line.json 2> gpurun_out/r5b/sweep_err.log
python - <<'P'
import json
d=json.load(open("gpurun_out/r5b/sweep_detail.json"))
for r in d["sweep"]["rows"]: print(r["model"], r["list"], r["k"], r["ms_fwd_bwd"], r["M_edge_messages_per_s"])
P
cp gpurun_out/parity_ledger.json gpurun_out/r5b/ 2>/dev/null
