#!/bin/bash
TAG=${1:-r01i}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
echo "== bench schnet"; timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench_schnet.json 2> $OUT/bench_schnet.err; cut -c1-230 $OUT/bench_schnet.json
echo "== rocprof schnet"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_schnet -o schnet -- python $ROOT/bench.py --kind schnet --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $OUT/rp_schnet.log 2>&1; echo "rocprof rc=$?")
f=$(find $OUT/rp_schnet -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/schnet_kernel_stats.csv && head -6 "$f" | cut -c1-150
grep -o '{"metric.*' $OUT/rp_schnet.log > $OUT/schnet_bench_under_rocprof.json
rm -rf $OUT/rp_schnet $OUT/rp_schnet.log
