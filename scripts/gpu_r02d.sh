#!/bin/bash
TAG=${1:-r02d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== mol tests"; timeout 900 python -m pytest tests/test_gpu_mol.py -q -m gpu -p no:cacheprovider > $OUT/pytest_mol.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest_mol.log | cut -c1-300
echo "== timing"; timeout 300 python scripts/mol_timing.py 256 2>&1 | tail -50
echo "== golden tests"; timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -x > $OUT/pytest_models.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_models.log | cut -c1-300
echo "== bench schnet"; timeout 900 python bench.py --steps 100 --warmup 10 --no-pmc --no-sweep --no-md --cpu-reps 3 > $OUT/bench_schnet.json 2> $OUT/bench_schnet.err; echo "rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench_schnet.json"))
print(d["value"], d["ms_per_step"], d["roofline"])
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["us_per_step"]): print("  %-28s %6.2f x %5.1f us = %7.1f" % (k, v["launches_per_step"], v["avg_us"], v["us_per_step"]))
print(d["cpu_baseline"])
PY
tail -3 $OUT/bench_schnet.err
