#!/bin/bash
# round 4: Dense layers of the force-matching engine on (value, tangent) pairs in one launch (default) against the separate launches
# they replace (SPK_FM_NO_DUAL=1): kernel parity, engine / training tests on both routes, then the training bench lines A/B on one box
OUT=gpurun_out/${1:-r04dual}; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "value_tangent_pairs" 2>&1 | tail -5) | tee $OUT/pytest_kernel.log
(timeout 900 python -m pytest tests/test_gpu_fm.py tests/test_gpu_train.py -q 2>&1 | tail -5) | tee $OUT/pytest.log
(SPK_FM_NO_DUAL=1 timeout 900 python -m pytest tests/test_gpu_fm.py -q 2>&1 | tail -3) | tee $OUT/pytest_nodual.log
for k in schnet painn; do for ND in 0 1 0 1; do
  if [ $ND = 0 ]; then unset SPK_FM_NO_DUAL; else export SPK_FM_NO_DUAL=1; fi
  timeout 200 python bench.py --mode train --kind $k --no-cpu-baseline > $OUT/train_${k}_nd$ND.json 2> $OUT/train_${k}_nd$ND.err
  python - <<PY
import json
d = json.loads(open("$OUT/train_${k}_nd$ND.json").read().strip().splitlines()[-1])
print("$k", "pair launches    " if "$ND" == "0" else "separate launches", d["ms_per_step"], "ms/step", d["value"], "samples/s", d.get("launches_per_step"), "launches", "loss", d["config"].get("first_loss"), d["config"].get("last_loss"))
PY
done; done 2>&1 | tee $OUT/ab.txt
