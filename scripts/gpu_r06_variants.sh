#!/bin/bash
# Round 6 debug: rebuild spk_cfconv with each flag set ON THE BOX, count bad edges (r06_box_gr_diff2.py) and time the water SchNet force call.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/${TAG:-r06v}; mkdir -p $OUT
for V in "$@"; do
  touch schnetpack_amd/csrc/spk_cfconv.hip
  FL=$(echo $V | sed "s/+/ /g"); [ "$V" = none ] && FL=""
  SPK_EXTRA_FLAGS="$FL" python -m schnetpack_amd.csrc.build > $OUT/build_$V.txt 2>&1 || { echo "$V: build failed"; continue; }
  echo "=== $V ($FL)"
  python scripts/r06_box_gr_diff2.py 2>/dev/null | grep -v "^ *print\|UserWarning\|Consider\|amdgpu.ids"
  env SPK_SPLIT=1 timeout 600 python bench.py --kind schnet --steps 200 --warmup 20 --no-cpu-baseline --no-md --no-sweep --no-pmc --no-train --no-drop-in --no-painn --no-pimd --workload water \
      --detail $OUT/detail_$V.json 2>$OUT/err_$V.txt > $OUT/line_$V.json
  python - <<PY
import json
d = json.load(open("$OUT/detail_$V.json"))
k = d.get("kernels") or {}
print("   %.4f ms  " % d["ms_per_step"], {n: round(v["avg_us"], 1) for n, v in k.items() if isinstance(v, dict) and "avg_us" in v and "cfconv" in n})
PY
done
