#!/bin/bash
TAG=${1:-dbg}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=line 2>&1 | tail -25 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_schnet.json 2> $OUT/bench.err
echo rc=$?; cat $OUT/bench_schnet.json; grep -v amdgpu.ids $OUT/bench.err | grep -v Warning | tail -5
timeout 300 python bench.py --kind painn --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_painn.json 2> $OUT/bench2.err
echo rc=$?; cat $OUT/bench_painn.json; grep -v amdgpu.ids $OUT/bench2.err | grep -v Warning | tail -5
