#!/bin/bash
# round 4: training step at larger batches (frames per step) -- the 8-frame line is latency bound; how the engine scales with the batch
OUT=gpurun_out/${1:-r04frames}; mkdir -p $OUT
for k in schnet painn; do for FR in 8 32 128; do
  timeout 300 python bench.py --mode train --kind $k --train-frames $FR --no-cpu-baseline > $OUT/train_${k}_$FR.json 2> $OUT/train_${k}_$FR.err
  python - <<PY
import json
d = json.loads(open("$OUT/train_${k}_$FR.json").read().strip().splitlines()[-1])
print("$k frames $FR", d["ms_per_step"], "ms/step", d["value"], "samples/s", d.get("launches_per_step"), "launches")
PY
done; done 2>&1 | tee $OUT/frames.txt
