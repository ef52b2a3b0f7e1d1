#!/bin/bash
# round 4: which of the pair-Dense uses pay (SPK_FM_DUAL_MASK bits: 1 forward pair, 2 tangent alone, 4 reverse of the pair)
OUT=gpurun_out/${1:-r04mask}; mkdir -p $OUT
for k in painn schnet; do for MASK in 0 1 2 4 3 7 0 7; do
  export SPK_FM_DUAL_MASK=$MASK
  timeout 200 python bench.py --mode train --kind $k --no-cpu-baseline > $OUT/train_${k}_m$MASK.json 2> $OUT/train_${k}_m$MASK.err
  python - <<PY
import json
d = json.loads(open("$OUT/train_${k}_m$MASK.json").read().strip().splitlines()[-1])
print("$k mask $MASK", d["ms_per_step"], "ms/step", d.get("launches_per_step"), "launches")
PY
done; done 2>&1 | tee $OUT/ab.txt
