#!/bin/bash
# round 5: segmented-sum unroll A/B (4 / 8 / 16 entries in flight per thread), scatter_add cases of bench.py
set -x
mkdir -p gpurun_out/r5s
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "scatter" 2>&1 | tail -3
for u in 4 8 16; do
  SPK_SEGSUM_UNROLL=$u timeout 600 python bench.py --steps 20 --warmup 5 --no-sweep --no-painn --no-train --no-drop-in --no-pmc --no-cpu-baseline --no-pimd --md-steps 20 --detail gpurun_out/r5s/d$u.json > /dev/null 2>> gpurun_out/r5s/err.log
  python -c "
import json
d=json.load(open('gpurun_out/r5s/d$u.json'))['scatter_add']
print('unroll $u', {k:(v['us'], v['frac'], v['frac_of_measured_copy']) for k,v in d.items() if isinstance(v, dict) and 'us' in v}, d['measured_copy_GBs'])"
done
