#!/bin/bash
# Round 6 measurement session: the driver's command, rocprofv3 kernel stats of the headline / PaiNN / water-box lines, the GPU test-suite.
#   usage: bash scripts/gpu_final_r06.sh <tag>     (SKIP_TESTS=1 leaves the test-suite out)
TAG=${1:-r06final}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== driver command"; SECONDS=0
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_driver_command_detail.json > $OUT/bench_driver_command_line.json 2> $OUT/bench_driver.err; echo "rc=$? wall=${SECONDS}s" | tee $OUT/bench_driver.wall; cut -c1-400 $OUT/bench_driver_command_line.json
for KIND in schnet painn; do
  echo "== rocprof $KIND (aspirin x 256)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_$KIND -o $KIND -- python $ROOT/bench.py --kind $KIND --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-md --no-sweep --no-pmc --no-painn --no-train --no-drop-in --no-pimd > $OUT/rp_$KIND.log 2>&1; echo "rocprof rc=$?")
  f=$(find $OUT/rp_$KIND -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${KIND}_kernel_stats.csv && head -4 "$f" | cut -c1-160
  rm -rf $OUT/rp_$KIND $OUT/rp_$KIND.log
  echo "== rocprof $KIND (water box)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rpw_$KIND -o $KIND -- python $ROOT/bench.py --kind $KIND --workload water --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-md --no-sweep --no-pmc --no-painn --no-train --no-drop-in --no-pimd > $OUT/rpw_$KIND.log 2>&1; echo "rocprof rc=$?")
  f=$(find $OUT/rpw_$KIND -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/water_${KIND}_kernel_stats.csv && head -8 "$f" | cut -c1-160
  rm -rf $OUT/rpw_$KIND $OUT/rpw_$KIND.log
done
if [ -z "$SKIP_TESTS" ]; then
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
  cp gpurun_out/parity_ledger.json $OUT/parity_ledger.json 2>/dev/null
fi
du -sh $OUT
