#!/bin/bash
# A/B of env-var settings on ONE box with the cfg-3 bench line (no stamps): usage gpu_ab_bench.sh TAG "VAR=val VAR2=val" "..." ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/${1:-abb}; mkdir -p $OUT; shift
KIND=${KIND:-painn}
for rep in 1 2; do
for V in "$@"; do
  env $V timeout 300 python bench.py --kind $KIND --steps 100 --warmup 10 --no-cpu-baseline --no-md --no-sweep --no-pmc --no-train --no-drop-in --no-painn 2>/dev/null > $OUT/b.json
  python - <<PY
import json
d=json.load(open("$OUT/b.json"))
print("%-40s %7.2f M  %.4f ms  " % ("$V", d["value"], d["ms_per_step"]), {k: round(v["avg_us"],1) for k,v in d["kernels"].items() if "mol" in k})
PY
done; done
