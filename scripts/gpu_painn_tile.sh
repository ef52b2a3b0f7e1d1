#!/bin/bash
# PaiNN message tile kernels: parity tests + timing against the row kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/tile
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -k "painn or PaiNN or pretrained or water or pbc or skin" 2>&1 | tail -15 | tee gpurun_out/tile/pytest.log
for W in aspirin water; do
  for ROW in 0 1; do
    if [ $ROW = 1 ]; then export SPK_PAINN_ROW=1; else unset SPK_PAINN_ROW; fi
    timeout 600 python bench.py --kind painn --workload $W --steps 30 --warmup 5 > gpurun_out/tile/bench_${W}_row$ROW.json 2> gpurun_out/tile/bench_${W}_row$ROW.err
    python - <<PY
import json
d=json.load(open("gpurun_out/tile/bench_${W}_row$ROW.json"))
print("$W row=$ROW", d["value"], d["ms_per_step"], {k: round(v["avg_us"],1) for k,v in d["kernels"].items() if "msg" in k}, d["cpu_baseline"].get("parity_rel_forces"))
PY
  done
done
