#!/bin/bash
# round 5, first GPU call: the new tests (ADVICE fixes, RCCL at world size 1, bench line contract) + the driver's bench command
set -x
mkdir -p gpurun_out/r5a
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fm.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -25 > gpurun_out/r5a/pytest_new.log
cat gpurun_out/r5a/pytest_new.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r5a/bench_detail.json > gpurun_out/r5a/bench_line.json 2> gpurun_out/r5a/bench_err.log
echo "bench rc=$?"
wc -c gpurun_out/r5a/bench_line.json
tail -c 9000 gpurun_out/r5a/bench_line.json
tail -5 gpurun_out/r5a/bench_err.log
timeout 600 python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -15 > gpurun_out/r5a/pytest_bench.log
cat gpurun_out/r5a/pytest_bench.log
