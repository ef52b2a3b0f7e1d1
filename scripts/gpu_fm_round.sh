#!/bin/bash
# round 4: one box -- kernel tests of the Dense / weight-gradient GEMMs, engine and training tests, training bench lines A/B
# (pair launches | SPK_FM_NO_DUAL=1), then per-kernel durations of the SchNet and PaiNN steps (rocprofv3 --kernel-trace --stats)
OUT=gpurun_out/${1:-r04round}; mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_ops.py -q -k "dense or gemm or tn" 2>&1 | tail -3) | tee $OUT/pytest_kernel.log
(timeout 900 python -m pytest tests/test_gpu_fm.py tests/test_gpu_train.py tests/test_gpu_reference_callers.py -q 2>&1 | tail -3) | tee $OUT/pytest.log
for k in schnet painn; do for ND in 0 1 0 1; do
  if [ $ND = 0 ]; then unset SPK_FM_NO_DUAL; else export SPK_FM_NO_DUAL=1; fi
  timeout 200 python bench.py --mode train --kind $k --no-cpu-baseline > $OUT/train_${k}_nd$ND.json 2> $OUT/train_${k}_nd$ND.err
  python - <<PY
import json
d = json.loads(open("$OUT/train_${k}_nd$ND.json").read().strip().splitlines()[-1])
print("$k", "pair launches    " if "$ND" == "0" else "separate launches", d["ms_per_step"], "ms/step", d["value"], "samples/s", d.get("launches_per_step"), "launches", "loss", d["config"].get("first_loss"), d["config"].get("last_loss"))
PY
done; done 2>&1 | tee $OUT/ab.txt
unset SPK_FM_NO_DUAL
cd /tmp && export TMPDIR=/tmp
for k in schnet painn; do
  rm -rf /tmp/prof_$k
  (cd $ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$k -o tr -- python bench.py --mode train --kind $k --no-cpu-baseline > $ROOT/$OUT/prof_bench_$k.json 2> $ROOT/$OUT/prof_bench_$k.err)
  f=$(find /tmp/prof_$k -name "*kernel_stats.csv" | head -1)
  cp "$f" $ROOT/$OUT/train_${k}_kernel_stats.csv
done
