#!/bin/bash
# round 4: XCD-contiguous walk of the persistent tile / row loops on (SPK_XCD_WALK=1, default) / off: water-box force call, both models
OUT=gpurun_out/${1:-r04u}; mkdir -p $OUT
for K in painn schnet; do for B in 0 1; do
  SPK_XCD_WALK=$B timeout 400 python bench.py --kind $K --workload water --steps 20 --warmup 5 --no-md --no-sweep --no-cpu-baseline --no-pmc > $OUT/water_${K}_xcd$B.json 2> $OUT/water_${K}_xcd$B.err
  python - <<PY
import json
d = json.loads(open("$OUT/water_${K}_xcd$B.json").read().strip().splitlines()[-1])
print("$K SPK_XCD_WALK=$B", d["ms_per_step"], "ms", d["value"], d["unit"])
for k, v in sorted(d.get("kernels", {}).items()): print("   %-28s %9.1f us/step" % (k, v["us_per_step"]))
PY
done; done 2>&1 | tee $OUT/xcd_ab.txt
