#!/bin/bash
# Round-2 first GPU session: smoke, the whole GPU suite (incl. the new scale / reference-caller / bench tests), default bench lines.
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
(rocm-smi --showproductname 2>&1 | grep -E "Card|GFX" | head -4; echo "host cores: $(nproc)"; free -g | head -2) > $OUT/env.log
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; grep smoke $OUT/smoke.log
echo "== new tests first"; timeout 1500 python -m pytest tests/test_gpu_reference_callers.py tests/test_gpu_scale.py tests/test_gpu_bench.py -q -m gpu -p no:cacheprovider --durations=12 -x > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_new.log; tail -40 $OUT/pytest_new.log
echo "== bench schnet (default line)"; timeout 900 python bench.py --steps 100 --warmup 10 > $OUT/bench_schnet.json 2> $OUT/bench_schnet.err; echo "rc=$?"; cut -c1-3000 $OUT/bench_schnet.json; tail -5 $OUT/bench_schnet.err
echo "== bench painn"; timeout 900 python bench.py --kind painn --steps 100 --warmup 10 > $OUT/bench_painn.json 2> $OUT/bench_painn.err; echo "rc=$?"; cut -c1-3000 $OUT/bench_painn.json; tail -5 $OUT/bench_painn.err
echo "== rest of the suite"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 --deselect tests/test_gpu_scale.py --deselect tests/test_gpu_bench.py --deselect tests/test_gpu_reference_callers.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -14 $OUT/pytest_gpu.log
du -sh $OUT
