#!/bin/bash
# round 5: kernel trace of the padded-neighbour sweep for PaiNN (symmetric k = 16 / 32 / 64 and the asymmetric k = 32 list: TS + GEOM row passes)
set -x
mkdir -p gpurun_out/r5b2
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5b2/prof -o p -- python bench.py --kind painn --steps 5 --warmup 2 --no-md --no-pmc --no-cpu-baseline --detail gpurun_out/r5b2/detail.json > /dev/null 2> gpurun_out/r5b2/err.log
cp gpurun_out/r5b2/prof/p_kernel_stats.csv gpurun_out/r5b2/painn_sweep_kernel_stats.csv
rm -rf gpurun_out/r5b2/prof
head -12 gpurun_out/r5b2/painn_sweep_kernel_stats.csv | cut -c1-220
