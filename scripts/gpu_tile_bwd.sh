#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/tb
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -k "painn or PaiNN or pretrained or water or skin" 2>&1 | tail -3
for W in aspirin water; do
  timeout 400 python bench.py --kind painn --workload $W --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/tb/bench_$W.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/tb/bench_$W.json"))
print("$W", d["value"], d["ms_per_step"], {k: round(v["avg_us"],1) for k,v in d["kernels"].items() if "msg" in k})
PY
done
