#!/bin/bash
# round 5: row chains after the descriptor copy to LDS -- parity, step times (chains off / on / on without warm-up), timing with everything off
set -x
mkdir -p gpurun_out/r5g
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_fm.py -x -q 2>&1 | tail -4
for kind in schnet painn; do
  for cfg in "0 0" "1 0" "1 16" "1 31"; do
    set -- $cfg
    SPK_FM_CHAIN=$1 SPK_FM_CHAIN_DRY=$2 timeout 300 python bench.py --mode train --kind $kind --steps 200 --warmup 8 --no-pmc --no-cpu-baseline --detail gpurun_out/r5g/t_${kind}_$1_$2.json > /dev/null 2>> gpurun_out/r5g/err.log
    python -c "import json;d=json.load(open('gpurun_out/r5g/t_${kind}_$1_$2.json'));print('$kind chain=$1 dry=$2', d['ms_per_step'], d['launches_per_step'], d['config']['last_loss'])"
  done
done
