#!/bin/bash
TAG=${1:-r02c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== mol tests"; timeout 900 python -m pytest tests/test_gpu_mol.py -q -m gpu -p no:cacheprovider -x > $OUT/pytest_mol.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest_mol.log | cut -c1-300
echo "== bench schnet"; timeout 900 python bench.py --steps 100 --warmup 10 --no-pmc --no-sweep > $OUT/bench_schnet.json 2> $OUT/bench_schnet.err; echo "rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_schnet.json"))
print(d["value"], d["ms_per_step"], d["roofline"])
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["us_per_step"]): print("  %-28s %6.2f x %5.1f us = %7.1f" % (k, v["launches_per_step"], v["avg_us"], v["us_per_step"]))
print(d["md"])
print(d["cpu_baseline"])
PY
tail -3 $OUT/bench_schnet.err
echo "== suite (minus scale)"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_scale.py --deselect tests/test_gpu_mol.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest_gpu.log | cut -c1-300
