#!/bin/bash
# Round 6: box-regime PaiNN force call (water, 31 944 atoms) under the switches of the split / row-tile work, on ONE box.
#   usage: gpu_r06_painn_box.sh TAG "ENV1=.. ENV2=.." "ENV.." ...      (each argument = one configuration; "-" = defaults)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
TAG=${1:-box}; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
n=0
for CFG in "$@"; do
  n=$((n+1)); [ "$CFG" = "-" ] && CFG=""
  env $CFG timeout 600 python bench.py --kind painn --steps 100 --warmup 10 --no-cpu-baseline --no-md --no-sweep --no-pmc --no-train --no-drop-in --no-painn --no-pimd --workload water \
      --detail $OUT/detail_$n.json 2>$OUT/err_$n.txt > $OUT/line_$n.json
  python - <<PY
import json
d = json.load(open("$OUT/detail_$n.json"))
k = d.get("kernels") or {}
print("%-40s %.4f ms  " % ("$CFG" or "defaults", d["ms_per_step"]), {n: round(v["avg_us"], 1) for n, v in k.items() if isinstance(v, dict) and "avg_us" in v and "msg" in n})
PY
done
