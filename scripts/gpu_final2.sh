#!/bin/bash
TAG=${1:-r01h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
echo "== bench painn"; timeout 600 python bench.py --kind painn --steps 100 --warmup 10 > $OUT/bench_painn.json 2> $OUT/bench_painn.err; cut -c1-230 $OUT/bench_painn.json
echo "== rocprof painn"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_painn -o painn -- python $ROOT/bench.py --kind painn --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $OUT/rp_painn.log 2>&1; echo "rocprof rc=$?")
f=$(find $OUT/rp_painn -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/painn_kernel_stats.csv && head -8 "$f" | cut -c1-150
grep -o '{"metric.*' $OUT/rp_painn.log > $OUT/painn_bench_under_rocprof.json
rm -rf $OUT/rp_painn $OUT/rp_painn.log
echo "== water box PaiNN"; timeout 600 python bench.py --workload water --kind painn --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_water_painn.json 2>/dev/null; cut -c1-230 $OUT/bench_water_painn.json
echo "== MD PaiNN"
for W in aspirin water; do
  timeout 600 python bench.py --mode md --workload $W --kind painn --steps 200 --warmup 10 > $OUT/bench_md_${W}_painn.json 2>/dev/null; cut -c1-200 $OUT/bench_md_${W}_painn.json
done
