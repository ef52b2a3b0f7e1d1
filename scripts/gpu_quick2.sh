#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/q2
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_md.py tests/test_deploy.py -x -q 2>&1 | tail -4
for K in schnet painn; do
  timeout 300 python bench.py --kind $K --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/q2/bench_$K.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/q2/bench_$K.json"))
print("$K", d["value"], d["ms_per_step"], {k: round(v["avg_us"],1) for k,v in d["kernels"].items()})
PY
done
if [ "$1" = "water" ]; then
  for K in painn schnet; do
  timeout 400 python bench.py --kind $K --workload water --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/q2/bench_water_$K.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/q2/bench_water_$K.json"))
print("water $K", d["value"], d["ms_per_step"], {k: round(v["avg_us"],1) for k,v in d["kernels"].items()})
PY
  done
fi
