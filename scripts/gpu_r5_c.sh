#!/bin/bash
# round 5, third GPU call: row chains of the force-matching engine -- parity, launch counts, step time A/B
set -x
mkdir -p gpurun_out/r5c
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_fm.py -x -q 2>&1 | tail -25 > gpurun_out/r5c/pytest_fm.log
cat gpurun_out/r5c/pytest_fm.log
for kind in painn schnet; do
  for mode in 0 1; do
    SPK_FM_CHAIN=$mode timeout 300 python bench.py --mode train --kind $kind --steps 200 --warmup 8 --no-pmc --no-cpu-baseline --detail gpurun_out/r5c/train_${kind}_chain$mode.json > gpurun_out/r5c/train_${kind}_chain$mode.line 2>> gpurun_out/r5c/err.log
    python -c "import json;d=json.load(open('gpurun_out/r5c/train_${kind}_chain$mode.json'));print('$kind chain=$mode', d['ms_per_step'], d['launches_per_step'], d['value'], d['config']['last_loss'])"
  done
done
tail -5 gpurun_out/r5c/err.log
