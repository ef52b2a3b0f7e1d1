#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench lines, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box):  bash scripts/gpu_round.sh [tag]
# Everything that should come back is written under gpurun_out/<tag>/ (small files only).
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
(rocm-smi --showproductname 2>&1 | grep -E "Card|GFX" | head -4; echo "host cores: $(nproc)"; free -g | head -2) > $OUT/env.log
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; grep smoke $OUT/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -14 $OUT/pytest_gpu.log
echo "== bench schnet"; timeout 900 python bench.py --steps 100 --warmup 10 > $OUT/bench_schnet.json 2> $OUT/bench_schnet.err; echo "rc=$?"; cat $OUT/bench_schnet.json
echo "== bench painn"; timeout 900 python bench.py --kind painn --steps 100 --warmup 10 > $OUT/bench_painn.json 2> $OUT/bench_painn.err; echo "rc=$?"; cat $OUT/bench_painn.json
for KIND in schnet painn; do
  echo "== rocprof $KIND"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_$KIND -o $KIND -- python $ROOT/bench.py --kind $KIND --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $OUT/rp_$KIND.log 2>&1; echo "rocprof rc=$?")
  f=$(find $OUT/rp_$KIND -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${KIND}_kernel_stats.csv && head -12 "$f" | cut -c1-160
  grep -o '{"metric.*' $OUT/rp_$KIND.log > $OUT/${KIND}_bench_under_rocprof.json
  rm -rf $OUT/rp_$KIND $OUT/rp_$KIND.log
done
echo "== water box (configs[4] per-GPU share)"
for KIND in schnet painn; do
  timeout 900 python bench.py --workload water --kind $KIND --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_water_$KIND.json 2> $OUT/bench_water_$KIND.err; echo "rc=$?"; cut -c1-400 $OUT/bench_water_$KIND.json
done
echo "== training step (configs[3])"
for KIND in schnet painn; do
  timeout 600 python bench.py --mode train --kind $KIND --steps 50 --warmup 5 --cpu-reps 5 > $OUT/bench_train_$KIND.json 2> $OUT/bench_train_$KIND.err; echo "rc=$?"; cut -c1-300 $OUT/bench_train_$KIND.json
done
echo "== MD loop (NVE, device neighbour list with skin, graph replay)"
for W in aspirin water; do for KIND in schnet painn; do
  timeout 600 python bench.py --mode md --workload $W --kind $KIND --steps 200 --warmup 10 > $OUT/bench_md_${W}_$KIND.json 2> $OUT/bench_md_${W}_$KIND.err; echo "rc=$?"; cut -c1-200 $OUT/bench_md_${W}_$KIND.json
done; done
echo "== PMC traffic"
bash $ROOT/scripts/gpu_pmc_traffic.sh $TAG schnet aspirin
bash $ROOT/scripts/gpu_pmc_traffic.sh $TAG painn aspirin
bash $ROOT/scripts/gpu_pmc_traffic.sh $TAG painn water
bash $ROOT/scripts/gpu_pmc_traffic.sh $TAG schnet water
du -sh $OUT
