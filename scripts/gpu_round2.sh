#!/bin/bash
# Round-2 measurement session (one GPU box): smoke, the whole GPU suite, every bench line quoted in DESIGN.md / profiles/README.md,
# rocprofv3 kernel stats of the eval bench commands.  Small files only, under gpurun_out/<tag>/.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
(rocm-smi --showproductname 2>&1 | grep -E "Card|GFX" | head -4; echo "host cores: $(nproc)"; free -g | head -2) > $OUT/env.log
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; grep "smoke " $OUT/smoke.log
echo "== pytest gpu"; timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -14 $OUT/pytest_gpu.log | cut -c1-200
for KIND in schnet painn; do
  echo "== bench $KIND (default line: roofline + in-run PMC + cpu_baseline(reference) + md + sweep)"
  timeout 900 python bench.py --kind $KIND --steps 100 --warmup 10 > $OUT/bench_$KIND.json 2> $OUT/bench_$KIND.err; echo "rc=$?"; cut -c1-600 $OUT/bench_$KIND.json
done
echo "== water box (configs[4] per-GPU share), with the reference on the host as parity check"
for KIND in schnet painn; do
  timeout 1200 python bench.py --workload water --kind $KIND --steps 30 --warmup 5 --no-md --no-sweep --cpu-reps 1 > $OUT/bench_water_$KIND.json 2> $OUT/bench_water_$KIND.err; echo "rc=$?"; cut -c1-400 $OUT/bench_water_$KIND.json
done
echo "== training step (configs[3])"
for KIND in schnet painn; do
  timeout 600 python bench.py --mode train --kind $KIND --steps 50 --warmup 5 --cpu-reps 5 > $OUT/bench_train_$KIND.json 2> $OUT/bench_train_$KIND.err; echo "rc=$?"; cut -c1-300 $OUT/bench_train_$KIND.json
done
echo "== training step at larger per-GPU batches"
for FR in 32 128; do for KIND in schnet painn; do
  timeout 600 python bench.py --mode train --kind $KIND --train-frames $FR --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_train_${KIND}_$FR.json 2> $OUT/bench_train_${KIND}_$FR.err; echo "rc=$? $(cut -c1-200 $OUT/bench_train_${KIND}_$FR.json | grep -o '"value": [0-9.]*, "unit": "[a-z/]*"') $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_train_${KIND}_$FR.json)"
done; done
echo "== RPMD NVE (4 beads x 64 aspirin)"; timeout 600 python bench.py --mode md --beads 4 --frames 64 --steps 200 --warmup 10 > $OUT/bench_rpmd_aspirin_schnet.json 2> $OUT/bench_rpmd.err; echo "rc=$?"; cut -c1-250 $OUT/bench_rpmd_aspirin_schnet.json
for KIND in schnet painn; do
  echo "== rocprof $KIND"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_$KIND -o $KIND -- python $ROOT/bench.py --kind $KIND --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-md --no-sweep --no-pmc > $OUT/rp_$KIND.log 2>&1; echo "rocprof rc=$?")
  f=$(find $OUT/rp_$KIND -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${KIND}_kernel_stats.csv && head -8 "$f" | cut -c1-160
  grep -o '{"metric.*' $OUT/rp_$KIND.log > $OUT/${KIND}_bench_under_rocprof.json
  rm -rf $OUT/rp_$KIND $OUT/rp_$KIND.log
done
du -sh $OUT
