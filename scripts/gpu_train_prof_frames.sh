#!/bin/bash
# round 4: per-kernel durations of the training step at a larger batch (default 128 frames)
OUT=gpurun_out/${1:-r04prof128}; mkdir -p $OUT
FR=${2:-128}
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for k in schnet painn; do
  rm -rf /tmp/prof_$k
  (cd $ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$k -o tr -- python bench.py --mode train --kind $k --train-frames $FR --no-cpu-baseline --steps 50 --warmup 5 > $ROOT/$OUT/prof_bench_$k.json 2> $ROOT/$OUT/prof_bench_$k.err)
  f=$(find /tmp/prof_$k -name "*kernel_stats.csv" | head -1)
  cp "$f" $ROOT/$OUT/train_${k}_${FR}frames_kernel_stats.csv
done
