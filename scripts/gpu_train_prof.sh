#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-trainprof}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for K in schnet painn; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_$K -o $K -- python $ROOT/bench.py --mode train --kind $K --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $OUT/rp_$K.log 2>&1
f=$(find $OUT/rp_$K -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/train_${K}_kernel_stats.csv
rm -rf $OUT/rp_$K
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/train_${K}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("$K total kernel time %.2f ms over 25 steps = %.2f ms/step; %d launches = %.0f per step" % (tot/1e6, tot/1e6/25, calls, calls/25))
for r in rows[:18]: print("  %-90s calls %5s avg %6.1f us %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
