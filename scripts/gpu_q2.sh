#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p gpurun_out/r02k
timeout 900 python bench.py --kind painn --steps 100 --warmup 10 > gpurun_out/r02k/bench_painn.json 2> gpurun_out/r02k/bench_painn.err; echo rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/r02k/bench_painn.json"))
print(d["value"], d["ms_per_step"])
print({a:(round(v["avg_us"],1), v.get("frac_of_peak"), v.get("bound")) for a,v in d["kernels"].items()})
PY
timeout 900 python bench.py --steps 100 --warmup 10 > gpurun_out/r02k/bench_schnet.json 2> gpurun_out/r02k/bench_schnet.err; echo rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/r02k/bench_schnet.json"))
print(d["value"], d["ms_per_step"], {a:(round(v["avg_us"],1), v.get("frac_of_peak")) for a,v in d["kernels"].items()})
PY
