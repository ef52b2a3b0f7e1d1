#!/bin/bash
# MFMA-pipe busy cycles of the hot kernels (separate --pmc passes, kernel-trace only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r02k; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for KIND in schnet painn; do
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmc_$C -o p -- python $ROOT/bench.py --pmc-child --kind $KIND --workload aspirin --frames 256 --water-side 22 --variant auto > /dev/null 2>&1
  echo "$KIND $C rc=$?"
  python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("/tmp/pmc_$C/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name")!="$C": continue
        n=r["Kernel_Name"][:60]; a=acc[n]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("$OUT/pmc_mfma_$KIND.txt","a") as fh:
    for n,(v,c) in sorted(acc.items(), key=lambda x:-x[1][0])[:8]:
        line="%-14s %-28s %-62s per-dispatch %.4g  (%d dispatches)" % ("$KIND","$C",n,v/c,c); print(line); fh.write(line+"\n")
PY
done; done
