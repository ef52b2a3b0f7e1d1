#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ss
timeout 400 python -m pytest tests/test_gpu_models.py tests/test_gpu_md.py -x -q -k "schnet or SchNet or golden or nve or NVE" 2>&1 | tail -2
for W in aspirin water; do
  timeout 400 python bench.py --kind schnet --workload $W --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ss/bench_$W.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ss/bench_$W.json"))
print("$W", d["value"], d["ms_per_step"], {k: round(v["avg_us"],1) for k,v in d["kernels"].items() if "cfconv" in k})
PY
done
