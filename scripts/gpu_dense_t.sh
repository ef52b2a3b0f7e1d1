#!/bin/bash
# transposed Dense forms with 4-block chunks: training benches + the tests that exercise them
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for KIND in schnet painn; do
timeout 100 python bench.py --mode train --kind $KIND --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/tt_$KIND.json 2> gpurun_out/tt_$KIND.err; echo "$KIND rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/tt_$KIND.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/tt_$KIND.json)"
done
timeout 150 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_ops.py tests/test_gpu_train.py -x -q -p no:cacheprovider 2>&1 | tail -2
