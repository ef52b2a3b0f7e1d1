#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/${1:-water}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -k "water" 2>&1 | tail -3
for K in painn schnet; do
timeout 900 python bench.py --workload water --kind $K --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_water_$K.json 2> $OUT/bench_water_$K.err
echo rc=$?; tail -3 $OUT/bench_water_$K.err | grep -v amdgpu
python - <<PY
import json
d=json.load(open("$OUT/bench_water_$K.json"))
print("$K water", d["value"], "M edge-msg/s", d["ms_per_step"], "ms/step graph", d["config"]["hip_graph"])
print("   ", d["config"]["workload"][-60:])
for k,v in sorted(d["kernels"].items()): print("   %-24s x%.0f  %.1f us  -> %.0f us/step" % (k, v["launches_per_step"], v["avg_us"], v["us_per_step"]))
print("   roofline", {k: d["roofline"][k] for k in ("kernel","achieved","unit","frac")} if d["roofline"] else None)
PY
done
