#!/bin/bash
# round 5: row chains -- parity + step-time A/B + per-launch durations in one call
set -x
mkdir -p gpurun_out/r5e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fm.py -x -q 2>&1 | tail -8
for kind in painn schnet; do
  for mode in 0 1; do
    SPK_FM_CHAIN=$mode timeout 300 python bench.py --mode train --kind $kind --steps 200 --warmup 8 --no-pmc --no-cpu-baseline --detail gpurun_out/r5e/train_${kind}_chain$mode.json > /dev/null 2>> gpurun_out/r5e/err.log
    python -c "import json;d=json.load(open('gpurun_out/r5e/train_${kind}_chain$mode.json'));print('$kind chain=$mode', d['ms_per_step'], d['launches_per_step'], d['value'], d['config']['last_loss'])"
  done
  SPK_FM_CHAIN=1 SPK_FM_CHAIN_DEBUG=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5e/prof_$kind -o t -- python bench.py --mode train --kind $kind --steps 2 --warmup 3 --no-graph --no-pmc --no-cpu-baseline --detail /tmp/d.json > /dev/null 2> gpurun_out/r5e/stages_$kind.log
done
