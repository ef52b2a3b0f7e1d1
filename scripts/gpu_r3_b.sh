#!/bin/bash
# round 3: molecule-resident PaiNN forward -- parity tests + cfg-3 bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r3b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_painn_mol.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -30 | tee $OUT/pytest_painn_mol.log
timeout 600 python bench.py --kind painn --steps 50 --warmup 5 --no-cpu-baseline --no-md --no-sweep --no-pmc > $OUT/bench_painn.json 2> $OUT/bench_painn.err
echo rc=$?; tail -3 $OUT/bench_painn.err | grep -v amdgpu
python - <<PY
import json
d=json.load(open("$OUT/bench_painn.json"))
print("painn", d["value"], "M edge-msg/s", d["ms_per_step"], "ms/step graph", d["config"]["hip_graph"])
for k,v in sorted(d["kernels"].items()):
    print("   %-26s x%.0f  %.1f us -> %.0f us/step" % (k, v["launches_per_step"], v["avg_us"], v["us_per_step"]))
PY
