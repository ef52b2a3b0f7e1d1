#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r02f; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_scale.py -x 2>&1 | tail -8 | cut -c1-300
for KIND in schnet painn; do
  timeout 600 python bench.py --mode train --kind $KIND --steps 50 --warmup 5 --cpu-reps 3 > $OUT/train_$KIND.json 2>$OUT/train_$KIND.err; python -c "
import json; d=json.load(open('$OUT/train_$KIND.json')); print('$KIND', d['value'], d['ms_per_step'], d['cpu_baseline'])"
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_train -o t -- python $OLDPWD/bench.py --mode train --kind painn --steps 20 --warmup 5 --no-cpu-baseline > /tmp/rp.log 2>&1; echo rc=$?)
f=$(find /tmp/rp_train -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/train_painn_kernel_stats.csv; head -25 "$f" | cut -c1-150
