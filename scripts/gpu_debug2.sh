#!/bin/bash
# debug round: full GPU tests, both eval benches, training benches
TAG=${1:-dbg}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -25 > $OUT/pytest.log
cat $OUT/pytest.log
for K in schnet painn; do
timeout 300 python bench.py --kind $K --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_$K.json 2> $OUT/bench_$K.err
echo rc=$?; tail -3 $OUT/bench_$K.err; python - <<PY
import json
d=json.load(open("$OUT/bench_$K.json"))
print("$K", d["value"], "M edge-msg/s", d["ms_per_step"], "ms/step graph", d["config"]["hip_graph"])
for k,v in sorted(d["kernels"].items()): print("   %-24s x%.0f  %.1f us  -> %.0f us/step" % (k, v["launches_per_step"], v["avg_us"], v["us_per_step"]))
print("   roofline", d["roofline"])
PY
timeout 300 python bench.py --mode train --kind $K --steps 30 --warmup 5 --cpu-reps 5 > $OUT/train_$K.json 2> $OUT/train_$K.err
echo train rc=$?; tail -3 $OUT/train_$K.err; cat $OUT/train_$K.json
done
