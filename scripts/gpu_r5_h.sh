#!/bin/bash
# round 5: index jobs of a training step in one launch -- parity of the training tests, step time
set -x
mkdir -p gpurun_out/r5h
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fm.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -4
for kind in schnet painn; do
  timeout 300 python bench.py --mode train --kind $kind --steps 300 --warmup 8 --no-pmc --no-cpu-baseline --detail gpurun_out/r5h/t_$kind.json > /dev/null 2>> gpurun_out/r5h/err.log
  python -c "import json;d=json.load(open('gpurun_out/r5h/t_$kind.json'));print('$kind', d['ms_per_step'], d['launches_per_step'], d['value'], d['config']['last_loss'])"
done
