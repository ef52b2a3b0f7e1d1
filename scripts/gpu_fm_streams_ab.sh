#!/bin/bash
# round 4: force-matching engine with (SPK_FM_STREAMS=1) / without (default) its side streams: tests, then the training bench lines A/B on one box
OUT=gpurun_out/${1:-r04z}; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_fm.py tests/test_gpu_train.py -q 2>&1 | tail -5) | tee $OUT/pytest.log
for k in schnet painn; do for NS in 1 0 1 0; do
  if [ $NS = 1 ]; then unset SPK_FM_STREAMS; else export SPK_FM_STREAMS=1; fi
  timeout 200 python bench.py --mode train --kind $k --no-cpu-baseline > $OUT/train_${k}_ns$NS.json 2> $OUT/train_${k}_ns$NS.err
  python - <<PY
import json
d = json.loads(open("$OUT/train_${k}_ns$NS.json").read().strip().splitlines()[-1])
print("$k", "no side streams" if "$NS" == "1" else "side streams   ", d["ms_per_step"], "ms/step", d["value"], "samples/s", d.get("launches_per_step"), "launches", "loss", d["config"].get("first_loss"), d["config"].get("last_loss"))
PY
done; done 2>&1 | tee $OUT/ab.txt
