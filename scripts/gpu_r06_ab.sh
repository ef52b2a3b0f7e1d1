#!/bin/bash
# Round 6: A/B of the split-precision matrix path (SPK_SPLIT=0 / 1) on ONE box.
#   usage: gpu_r06_ab.sh TAG KIND [extra bench args]      (KIND = schnet | painn; WORKLOAD=water for the box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
TAG=${1:-ab}; KIND=${2:-schnet}; shift; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
for V in 0 1; do
  env SPK_SPLIT=$V timeout 600 python bench.py --kind $KIND --steps 200 --warmup 20 --no-cpu-baseline --no-md --no-sweep --no-pmc --no-train --no-drop-in --no-painn --no-pimd \
      --detail $OUT/detail_${KIND}_split$V.json "$@" 2>$OUT/err_$V.txt > $OUT/line_${KIND}_split$V.json
  python - <<PY
import json
d = json.load(open("$OUT/detail_${KIND}_split$V.json"))
k = d.get("kernels") or {}
print("SPK_SPLIT=$V %-8s %8.2f M  %.4f ms  " % ("$KIND", d["value"], d["ms_per_step"]), {n: round(v["avg_us"], 1) for n, v in k.items() if isinstance(v, dict) and "avg_us" in v})
PY
done; done
