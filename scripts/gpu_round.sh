#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench lines, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box):  bash scripts/gpu_round.sh [tag]
# Everything that should come back is written under gpurun_out/<tag>/.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== rocm-smi" > $OUT/env.log; (rocm-smi --showproductname 2>&1 | head -20; nproc; free -g | head -2) >> $OUT/env.log
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -5 $OUT/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -40 $OUT/pytest_gpu.log
echo "== bench schnet"; timeout 900 python bench.py --steps 50 --warmup 10 > $OUT/bench_schnet.json 2> $OUT/bench_schnet.err; echo "rc=$?"; cat $OUT/bench_schnet.json; tail -5 $OUT/bench_schnet.err
echo "== bench schnet eager"; timeout 600 python bench.py --steps 50 --warmup 10 --no-graph --no-cpu-baseline > $OUT/bench_schnet_eager.json 2> $OUT/bench_schnet_eager.err; echo "rc=$?"; cat $OUT/bench_schnet_eager.json
echo "== bench painn"; timeout 900 python bench.py --kind painn --steps 50 --warmup 10 > $OUT/bench_painn.json 2> $OUT/bench_painn.err; echo "rc=$?"; cat $OUT/bench_painn.json; tail -5 $OUT/bench_painn.err
echo "== rocprof schnet"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_schnet -o schnet -- python $ROOT/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $OUT/prof_schnet.log 2>&1; echo "rocprof rc=$?")
find $OUT/prof_schnet -name "*stats*" | head; f=$(find $OUT/prof_schnet -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
echo "== rocprof painn"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_painn -o painn -- python $ROOT/bench.py --kind painn --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $OUT/prof_painn.log 2>&1; echo "rocprof rc=$?")
f=$(find $OUT/prof_painn -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep only the small summaries (the raw traces can be large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
