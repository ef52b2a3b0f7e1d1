#!/bin/bash
# L2 (TCC) hit / miss counts of the box-regime PaiNN kernels: rocprofv3 --pmc over the bench's own few-call child (kernel-trace only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/${1:-r06l2}; mkdir -p $OUT
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  d=$OUT/$(echo $C | tr ' ' '_'); rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $d -o p -- python bench.py --pmc-child --kind painn --workload water > $OUT/log.txt 2>&1
done
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/*/**/p_counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if "painn_msg" in k: print(k, {c: "%.3e" % (x / n[(k, c)]) for c, x in v.items()})
PY
