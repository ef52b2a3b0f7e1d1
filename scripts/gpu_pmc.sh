#!/bin/bash
# PMC counters of selected kernels (separate passes; kernel-trace only, as gpurun requires)
TAG=${1:-pmc}
KIND=${2:-schnet}
FILTER=${3:-chain}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run_pmc () {
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $2 -d $OUT/pmc_$1 -o p -- python $ROOT/bench.py --kind $KIND --steps 4 --warmup 2 --no-graph --no-cpu-baseline > $OUT/pmc_$1.log 2>&1
  echo "pmc $1 rc=$?"
}
run_pmc a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
run_pmc b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE"
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen=set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:46]
        if "$FILTER" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key=(k,row["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k]+=1
    for k,v in acc.items():
        print(" ", k, "dispatches", cnt[k])
        for c,val in sorted(v.items()): print("      %-28s %.4g per dispatch" % (c, val/cnt[k]))
PY
rm -rf $OUT/pmc_a $OUT/pmc_b
