#!/bin/bash
# round 4: quick check of an engine change -- engine / training tests, then the training bench lines of both models twice
OUT=gpurun_out/${1:-r04quick}; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_fm.py tests/test_gpu_train.py -q 2>&1 | tail -3) | tee $OUT/pytest.log
for k in schnet painn schnet painn; do
  timeout 200 python bench.py --mode train --kind $k --no-cpu-baseline > $OUT/train_$k.json 2> $OUT/train_$k.err
  python - <<PY
import json
d = json.loads(open("$OUT/train_$k.json").read().strip().splitlines()[-1])
print("$k", d["ms_per_step"], "ms/step", d["value"], "samples/s", d.get("launches_per_step"), "launches", "loss", d["config"].get("first_loss"), d["config"].get("last_loss"))
PY
done 2>&1 | tee $OUT/ab.txt
