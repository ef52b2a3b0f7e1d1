#!/bin/bash
# Second closing session of round 2 (after the eight-wave PaiNN mixing kernels): the whole GPU suite, smoke, the PaiNN lines.
TAG=${1:-r02n}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log | cut -c1-200
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; grep "smoke " $OUT/smoke.log
echo "== bench painn"; timeout 600 python bench.py --kind painn --steps 100 --warmup 10 > $OUT/bench_painn.json 2> $OUT/bench_painn.err; echo "rc=$?"; cut -c1-300 $OUT/bench_painn.json
echo "== water painn"; timeout 600 python bench.py --workload water --kind painn --steps 30 --warmup 5 --no-md --no-sweep --cpu-reps 1 > $OUT/bench_water_painn.json 2> $OUT/bench_water_painn.err; echo "rc=$?"; cut -c1-300 $OUT/bench_water_painn.json
echo "== rocprof painn"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_painn -o painn -- python $ROOT/bench.py --kind painn --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-md --no-sweep --no-pmc > $OUT/rp_painn.log 2>&1; echo "rocprof rc=$?")
f=$(find $OUT/rp_painn -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/painn_kernel_stats.csv && head -6 "$f" | cut -c1-160
grep -o '{"metric.*' $OUT/rp_painn.log > $OUT/painn_bench_under_rocprof.json
rm -rf $OUT/rp_painn $OUT/rp_painn.log
du -sh $OUT
