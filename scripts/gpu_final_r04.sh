#!/bin/bash
# Round 4 measurement session: the default bench line (driver command), PaiNN / water lines, training lines, rocprofv3 kernel stats of
# the SchNet, PaiNN and training lines, kernel resource table.  usage: bash scripts/gpu_final_r04.sh <tag>
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== default bench"; SECONDS=0
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$? wall=${SECONDS}s" | tee $OUT/bench_default.wall; cut -c1-300 $OUT/bench_default.json
if [ -z "$SKIP_DRIVER_LINE" ]; then
echo "== driver command"; SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver.err; echo "rc=$? wall=${SECONDS}s" | tee $OUT/bench_driver.wall; cut -c1-200 $OUT/bench_driver_command.json
fi
for KIND in schnet painn; do
  echo "== bench water $KIND"
  timeout 600 python bench.py --kind $KIND --workload water --steps 30 --warmup 5 --no-md --no-sweep --cpu-reps 1 > $OUT/bench_water_$KIND.json 2> $OUT/bench_water_$KIND.err; echo "rc=$?"; cut -c1-200 $OUT/bench_water_$KIND.json
done
echo "== bench painn aspirin (own line)"; timeout 600 python bench.py --kind painn --steps 100 --warmup 10 > $OUT/bench_painn.json 2> $OUT/bench_painn.err; echo "rc=$?"; cut -c1-200 $OUT/bench_painn.json
for k in schnet painn; do
  timeout 200 python bench.py --mode train --kind $k > $OUT/bench_train_$k.json 2> $OUT/train_$k.err; cut -c1-200 $OUT/bench_train_$k.json
done
for KIND in schnet painn; do
  echo "== rocprof $KIND"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_$KIND -o $KIND -- python $ROOT/bench.py --kind $KIND --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-md --no-sweep --no-pmc --no-painn --no-train --no-drop-in > $OUT/rp_$KIND.log 2>&1; echo "rocprof rc=$?")
  f=$(find $OUT/rp_$KIND -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${KIND}_kernel_stats.csv && head -4 "$f" | cut -c1-160
  grep -o '{"metric.*' $OUT/rp_$KIND.log > $OUT/${KIND}_bench_under_rocprof.json
  rm -rf $OUT/rp_$KIND $OUT/rp_$KIND.log
done
for k in schnet painn; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$k -o t -- python $ROOT/bench.py --mode train --kind $k --no-cpu-baseline --steps 50 --warmup 5 > /tmp/prof_$k.log 2>&1)
  f=$(find /tmp/prof_$k -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/train_${k}_kernel_stats.csv
done
if [ -z "$SKIP_RESOURCES" ]; then echo "== kernel resources"; timeout 600 python scripts/kernel_resources.py > $OUT/kernel_resources.md 2>/dev/null; tail -3 $OUT/kernel_resources.md; fi
du -sh $OUT
