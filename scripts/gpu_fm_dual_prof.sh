#!/bin/bash
# round 4: per-kernel durations of the training step with the pair Dense launches (default) and without (SPK_FM_NO_DUAL=1)
OUT=gpurun_out/${1:-r04dualprof}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ND in 0 1; do
  if [ $ND = 0 ]; then unset SPK_FM_NO_DUAL; else export SPK_FM_NO_DUAL=1; fi
  rm -rf /tmp/prof_$ND
  (cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$ND -o tr -- python bench.py --mode train --kind schnet --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_nd$ND.json 2> $GRAFT_REPO_ROOT/$OUT/bench_nd$ND.err)
  f=$(find /tmp/prof_$ND -name "*kernel_stats.csv" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/$OUT/train_schnet_nd${ND}_kernel_stats.csv
done
