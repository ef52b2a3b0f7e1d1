#!/bin/bash
# quick check: selected GPU tests + aspirin and water benches (no CPU baseline)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/${1:-quick}; mkdir -p $OUT
SEL=${2:-"atomwise or pairwise or golden"}
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -x -k "$SEL" 2>&1 | tail -15
for W in aspirin water; do for K in schnet painn; do
timeout 900 python bench.py --workload $W --kind $K --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_${W}_$K.json 2> $OUT/bench_${W}_$K.err
echo rc=$?; tail -3 $OUT/bench_${W}_$K.err | grep -v amdgpu
python - <<PY
import json
d=json.load(open("$OUT/bench_${W}_$K.json"))
print("$K $W", d["value"], "M edge-msg/s", d["ms_per_step"], "ms/step graph", d["config"]["hip_graph"])
tot=0
for k,v in sorted(d["kernels"].items()):
    print("   %-24s x%.0f  %.1f us  -> %.0f us/step" % (k, v["launches_per_step"], v["avg_us"], v["us_per_step"])); tot+=v["us_per_step"]
print("   profiled total %.0f us;" % tot, "roofline", {k: d["roofline"][k] for k in ("kernel","achieved","unit","frac")} if d["roofline"] else None)
PY
done; done
