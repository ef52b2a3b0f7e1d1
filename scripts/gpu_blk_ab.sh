#!/bin/bash
# round 4: water-box PaiNN force call with the block kernels of the message off (SPK_BLOCKS=0) / on (default rule): per-tag kernel times
OUT=gpurun_out/${1:-r04p}; mkdir -p $OUT
for B in 0 1; do
  SPK_BLOCKS=$B timeout 400 python bench.py --kind painn --workload water --steps 20 --warmup 5 --no-md --no-sweep --no-cpu-baseline --no-pmc > $OUT/water_painn_blocks$B.json 2> $OUT/water_painn_blocks$B.err
  python - <<PY
import json
d = json.loads(open("$OUT/water_painn_blocks$B.json").read().strip().splitlines()[-1])
print("SPK_BLOCKS=$B", d["ms_per_step"], "ms", d["value"], d["unit"])
for k, v in sorted(d.get("kernels", {}).items()): print("   %-28s %9.1f us/step" % (k, v["us_per_step"]))
PY
done
