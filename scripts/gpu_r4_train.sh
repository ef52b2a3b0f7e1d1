#!/bin/bash
# round 4: force-matching engine -- GPU tests of the engine + the training tests, the training bench lines and their rocprofv3 kernel traces
# usage (on the GPU box, from the repo root): bash scripts/gpu_r4_train.sh <tag> [notest]
tag=${1:-r4}
out=gpurun_out/$tag
mkdir -p $out
if [ "$2" != "notest" ]; then
  (timeout 900 python -m pytest tests/test_gpu_fm.py tests/test_gpu_train.py -q 2>&1 | tail -40) > $out/pytest.log 2>&1
  tail -6 $out/pytest.log
fi
for k in schnet painn; do
  timeout 200 python bench.py --mode train --kind $k --no-cpu-baseline > $out/train_$k.json 2> $out/train_$k.err
  python - <<PY
import json
d = json.loads(open("$out/train_$k.json").read().strip().splitlines()[-1])
print("$k", d["ms_per_step"], "ms/step", d["value"], "samples/s", d.get("launches_per_step"), "launches")
PY
done
export TMPDIR=/tmp
for k in schnet painn; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$k -o t -- python $GRAFT_REPO_ROOT/bench.py --mode train --kind $k --no-cpu-baseline --steps 50 --warmup 5 > /tmp/prof_$k.log 2>&1)
  f=$(find /tmp/prof_$k -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/train_${k}_kernel_stats.csv
  tail -2 /tmp/prof_$k.log > $out/train_${k}_under_rocprof.json
done
head -30 $out/train_painn_kernel_stats.csv
