#!/bin/bash
# Closing session of round 2 after the second pass over the molecule-resident kernels: smoke, the whole GPU suite, the default bench
# lines, RPMD, rocprofv3 kernel stats + MFMA counters of the SchNet line, cycle stamps.  (Water-box and training lines: unchanged
# kernels, see gpu_round2.sh.)
TAG=${1:-r02m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; grep "smoke " $OUT/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -10 $OUT/pytest_gpu.log | cut -c1-200
for KIND in schnet painn; do
  echo "== bench $KIND"
  timeout 600 python bench.py --kind $KIND --steps 100 --warmup 10 > $OUT/bench_$KIND.json 2> $OUT/bench_$KIND.err; echo "rc=$?"; cut -c1-500 $OUT/bench_$KIND.json
done
echo "== RPMD"; timeout 300 python bench.py --mode md --beads 4 --frames 64 --steps 200 --warmup 10 > $OUT/bench_rpmd_aspirin_schnet.json 2> $OUT/bench_rpmd.err; echo "rc=$?"; cut -c1-250 $OUT/bench_rpmd_aspirin_schnet.json
echo "== cycle stamps"; timeout 300 python scripts/mol_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/mol_cycle_stamps.txt; tail -3 $OUT/mol_cycle_stamps.txt
echo "== rocprof schnet"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_schnet -o schnet -- python $ROOT/bench.py --kind schnet --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-md --no-sweep --no-pmc > $OUT/rp_schnet.log 2>&1; echo "rocprof rc=$?")
f=$(find $OUT/rp_schnet -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/schnet_kernel_stats.csv && head -5 "$f" | cut -c1-160
grep -o '{"metric.*' $OUT/rp_schnet.log > $OUT/schnet_bench_under_rocprof.json
rm -rf $OUT/rp_schnet $OUT/rp_schnet.log
echo "== MFMA counters"
cd /tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmc_$C -o p -- python $ROOT/bench.py --pmc-child --kind schnet --workload aspirin --frames 256 --water-side 22 --variant auto > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("/tmp/pmc_$C/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name")!="$C": continue
        n=r["Kernel_Name"][:60]; a=acc[n]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("$OUT/pmc_mfma_schnet.txt","a") as fh:
    for n,(v,c) in sorted(acc.items(), key=lambda x:-x[1][0])[:2]:
        line="%-28s %-62s per-dispatch %.4g  (%d dispatches)" % ("$C",n,v/c,c); print(line); fh.write(line+"\n")
PY
done
du -sh $OUT
