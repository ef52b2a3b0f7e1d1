#!/bin/bash
# Round 3 measurement session: the default bench line (driver command), PaiNN / water lines, bead-parallel lines on two gloo ranks,
# rocprofv3 kernel stats of the SchNet and PaiNN lines, MFMA counters, PaiNN cycle stamps, kernel resource table.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== default bench"; SECONDS=0
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$? wall=${SECONDS}s" | tee $OUT/bench_default.wall; cut -c1-300 $OUT/bench_default.json
for KIND in schnet painn; do
  echo "== bench water $KIND"
  timeout 600 python bench.py --kind $KIND --workload water --steps 30 --warmup 5 --no-md --no-sweep --cpu-reps 2 > $OUT/bench_water_$KIND.json 2> $OUT/bench_water_$KIND.err; echo "rc=$?"; cut -c1-200 $OUT/bench_water_$KIND.json
done
echo "== bench painn aspirin (own line)"; timeout 600 python bench.py --kind painn --steps 100 --warmup 10 > $OUT/bench_painn.json 2> $OUT/bench_painn.err; echo "rc=$?"; cut -c1-200 $OUT/bench_painn.json
echo "== PIMD line (configs[4] on one GPU)"; timeout 600 python bench.py --mode md --kind painn --workload water --beads 8 --steps 30 --warmup 6 > $OUT/bench_pimd_water_painn.json 2> $OUT/bench_pimd.err; echo "rc=$?"; cut -c1-300 $OUT/bench_pimd_water_painn.json
for EX in state forces; do
  echo "== bead-parallel ($EX), two gloo ranks on the one device"
  SPK_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --mode md --beads 8 --bead-parallel $EX --kind painn --workload water --water-side 10 --steps 20 --warmup 4 2>/dev/null | grep '^{' > $OUT/bench_bead_parallel_${EX}_gloo2.json; cut -c1-200 $OUT/bench_bead_parallel_${EX}_gloo2.json
done
echo "== cycle stamps"; timeout 300 python scripts/painn_mol_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/painn_mol_cycle_stamps.txt; tail -3 $OUT/painn_mol_cycle_stamps.txt
timeout 300 python scripts/mol_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/mol_cycle_stamps.txt; tail -2 $OUT/mol_cycle_stamps.txt
for KIND in schnet painn; do
  echo "== rocprof $KIND"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_$KIND -o $KIND -- python $ROOT/bench.py --kind $KIND --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-md --no-sweep --no-pmc --no-painn --no-train --no-drop-in > $OUT/rp_$KIND.log 2>&1; echo "rocprof rc=$?")
  f=$(find $OUT/rp_$KIND -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${KIND}_kernel_stats.csv && head -4 "$f" | cut -c1-160
  grep -o '{"metric.*' $OUT/rp_$KIND.log > $OUT/${KIND}_bench_under_rocprof.json
  rm -rf $OUT/rp_$KIND $OUT/rp_$KIND.log
done
echo "== MFMA counters"
cd /tmp
for KIND in schnet painn; do
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmc_$C -o p -- python $ROOT/bench.py --pmc-child --kind $KIND --workload aspirin --frames 256 --water-side 22 --variant auto > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("/tmp/pmc_$C/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name")!="$C": continue
        n=r["Kernel_Name"][:60]; a=acc[n]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("$OUT/pmc_mfma.txt","a") as fh:
    for n,(v,c) in sorted(acc.items(), key=lambda x:-x[1][0])[:2]:
        line="%-7s %-28s %-62s per-dispatch %.4g  (%d dispatches)" % ("$KIND","$C",n,v/c,c); print(line); fh.write(line+"\n")
PY
done; done
du -sh $OUT
