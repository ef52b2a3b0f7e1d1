#!/bin/bash
# round 4: the `--mode train` lines with the HBM traffic of one step from this run's counters
OUT=gpurun_out/${1:-r04trainpmc}; mkdir -p $OUT
for k in schnet painn; do
  timeout 400 python bench.py --mode train --kind $k > $OUT/bench_train_$k.json 2> $OUT/bench_train_$k.err
  python - <<PY
import json
d = json.loads(open("$OUT/bench_train_$k.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$k", d["ms_per_step"], "ms/step", d["value"], "samples/s; traffic per step", r.get("traffic"), r.get("traffic_frac_of_hbm_peak"), r.get("traffic_detail", {}).get("dispatches_per_step"))
PY
done 2>&1 | tee $OUT/lines.txt
