#!/bin/bash
# training-regime kernels: primitive tests, model-level training tests, bench + launch profile
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r02g; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_train_ops.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_models.py tests/test_gpu_ops.py -x -q -k "train or dense or Dense or weight or gather or static or malformed" 2>&1 | tail -8
for KIND in schnet painn; do
  timeout 600 python bench.py --mode train --kind $KIND --steps 50 --warmup 5 --cpu-reps 2 > $OUT/bench_train_$KIND.json 2> $OUT/bench_train_$KIND.err; echo "rc=$?"; cut -c1-400 $OUT/bench_train_$KIND.json
done
cd /tmp
for KIND in schnet painn; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$KIND -o t -- python $GRAFT_REPO_ROOT/bench.py --mode train --kind $KIND --steps 20 --warmup 5 --no-graph --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/prof_$KIND.err
  F=$(find /tmp/prof_$KIND -name "*kernel_stats.csv" | head -1); cp $F $GRAFT_REPO_ROOT/$OUT/train_${KIND}_kernel_stats.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print("$KIND total ms %.2f calls %d = %.0f per step" % (tot/1e6, calls, calls/25))
for r in rows[:22]:
    print('  %-80s %6s %8.1f %5.1f%%'%(r['Name'][:80],r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY
done
