#!/bin/bash
# round 4: what bounds the box-regime PaiNN message kernels?  SQ instruction / wait counters of the row, tile and block kernels on the
# water box (separate --pmc passes, kernel-trace only).  usage: bash scripts/gpu_pmc_msg.sh <tag> [counter groups ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r04c}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export EXP_ORDER_ONLY_LATTICE=1
shift
if [ $# -eq 0 ]; then set -- "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "FETCH_SIZE WRITE_SIZE"; fi
for C in "$@"; do
  T=$(echo $C | tr ' ' '_'); rm -rf /tmp/pmc_$T
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $C -d /tmp/pmc_$T -o p -- python $ROOT/scripts/exp_order.py > /tmp/pmc_$T.log 2>&1
  echo "$C rc=$?"
  python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("/tmp/pmc_$T/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "painn" not in n: continue
        n=n.replace("void (anonymous namespace)::","").replace("void ","")
        a=acc[(n[:48], r["Counter_Name"])]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("$OUT/pmc_msg.txt","a") as fh:
    for (n,c),(v,k) in sorted(acc.items()):
        line="%-50s %-28s per-dispatch %.5g  (%d dispatches)" % (n,c,v/k,k); print(line); fh.write(line+"\n")
PY
done
