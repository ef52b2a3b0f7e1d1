#!/bin/bash
# trimmed end-of-round session: smoke, full GPU tests, headline bench lines, rocprofv3 kernel stats, PaiNN PMC traffic
TAG=${1:-r01g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; grep smoke $OUT/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
echo "== bench schnet"; timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench_schnet.json 2> $OUT/bench_schnet.err; cut -c1-330 $OUT/bench_schnet.json
echo "== bench painn"; timeout 600 python bench.py --kind painn --steps 100 --warmup 10 > $OUT/bench_painn.json 2> $OUT/bench_painn.err; cut -c1-330 $OUT/bench_painn.json
for KIND in schnet painn; do
  echo "== rocprof $KIND"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_$KIND -o $KIND -- python $ROOT/bench.py --kind $KIND --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $OUT/rp_$KIND.log 2>&1; echo "rocprof rc=$?")
  f=$(find $OUT/rp_$KIND -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${KIND}_kernel_stats.csv && head -6 "$f" | cut -c1-150
  grep -o '{"metric.*' $OUT/rp_$KIND.log > $OUT/${KIND}_bench_under_rocprof.json
  rm -rf $OUT/rp_$KIND $OUT/rp_$KIND.log
done
echo "== water box PaiNN"; timeout 600 python bench.py --workload water --kind painn --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_water_painn.json 2>/dev/null; cut -c1-300 $OUT/bench_water_painn.json
echo "== MD PaiNN"
for W in aspirin water; do
  timeout 600 python bench.py --mode md --workload $W --kind painn --steps 200 --warmup 10 > $OUT/bench_md_${W}_painn.json 2>/dev/null; cut -c1-200 $OUT/bench_md_${W}_painn.json
done
echo "== PMC traffic PaiNN aspirin"
bash $ROOT/scripts/gpu_pmc_traffic.sh $TAG painn aspirin 2>&1 | tail -8
du -sh $OUT
