#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command (eager, so every kernel is a separate dispatch)
TAG=${1:-prof}
KIND=${2:-schnet}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_$KIND -o $KIND -- python $ROOT/bench.py --kind $KIND --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $OUT/rp_$KIND.log 2>&1
echo "rocprof rc=$?"
f=$(find $OUT/rp_$KIND -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/${KIND}_kernel_stats.csv && head -40 "$f" | cut -c1-200
find $OUT/rp_$KIND -name "*kernel_trace.csv" -delete
tail -3 $OUT/rp_$KIND.log | cut -c1-600
