#!/bin/bash
# Round-2 GPU session: smoke, whole GPU suite, bench lines.
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; grep smoke $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== torchscript + reference callers + bench tests"; timeout 1200 python -m pytest tests/test_gpu_torchscript.py tests/test_gpu_reference_callers.py tests/test_gpu_bench.py -q -m gpu -p no:cacheprovider --durations=8 > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_new.log; tail -60 $OUT/pytest_new.log | cut -c1-400
echo "== rest of the suite"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 --deselect tests/test_gpu_torchscript.py --deselect tests/test_gpu_bench.py --deselect tests/test_gpu_reference_callers.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -40 $OUT/pytest_gpu.log | cut -c1-400
echo "== bench schnet"; timeout 900 python bench.py --steps 100 --warmup 10 --no-pmc > $OUT/bench_schnet.json 2> $OUT/bench_schnet.err; echo "rc=$?"; cut -c1-1500 $OUT/bench_schnet.json; tail -5 $OUT/bench_schnet.err
echo "== bench painn"; timeout 900 python bench.py --kind painn --steps 100 --warmup 10 --no-pmc --no-md --no-sweep > $OUT/bench_painn.json 2> $OUT/bench_painn.err; echo "rc=$?"; cut -c1-1500 $OUT/bench_painn.json; tail -5 $OUT/bench_painn.err
du -sh $OUT
