#!/bin/bash
# round 5: where a stage of the row-chain kernel spends its time -- step time of the chained SchNet / PaiNN training step with parts of the
# stages switched off (SPK_FM_CHAIN_DRY bits: 1 no epilogue loads / stores, 2 no weight loads, 4 no X loads, 8 no element-wise stages,
# 16 no warm-up; the results of such a run are WRONG, only its time means something)
set -x
mkdir -p gpurun_out/r5f
cd "$GRAFT_REPO_ROOT"
for kind in schnet painn; do
  for dry in 0 1 2 3 7 15 31; do
    SPK_FM_CHAIN_DRY=$dry SPK_FM_CHAIN=1 timeout 300 python bench.py --mode train --kind $kind --steps 200 --warmup 8 --no-pmc --no-cpu-baseline --detail gpurun_out/r5f/t_${kind}_$dry.json > /dev/null 2>> gpurun_out/r5f/err.log
    python -c "import json;d=json.load(open('gpurun_out/r5f/t_${kind}_$dry.json'));print('$kind dry=$dry', d['ms_per_step'], d['launches_per_step'])"
  done
done
