#!/bin/bash
# chain kernel: tests for both tile heights + timing of both on aspirin and water
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -x -k "chain or golden" 2>&1 | tail -8
python - <<'PY'
import json, subprocess, sys, os
for rows in (32, 16):
    for wl, kind in (("aspirin","schnet"),("aspirin","painn"),("water","schnet"),("water","painn")):
        env = dict(os.environ, SPK_CHAIN_ROWS=str(rows))
        out = subprocess.run([sys.executable, "bench.py", "--workload", wl, "--kind", kind, "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True, env=env)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(rows, wl, kind, "FAILED", out.stderr[-400:]); continue
        ks = {k: round(v["avg_us"],1) for k,v in d["kernels"].items() if k.startswith("chain")}
        print("rows", rows, wl, kind, d["value"], "M edge-msg/s", d["ms_per_step"], "ms", ks)
PY
