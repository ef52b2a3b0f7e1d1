#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 600 python bench.py --steps 50 --warmup 5 --no-sweep --no-md --cpu-reps 2 > gpurun_out/t2.json 2> gpurun_out/t2.err; echo rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/t2.json"))
print(d["value"], d["ms_per_step"])
r=d["roofline"]; print({k:r[k] for k in r if k!="traffic_detail"})
print(json.dumps(r.get("traffic_detail",{}).get("all_kernels"), indent=0))
PY
tail -3 gpurun_out/t2.err
