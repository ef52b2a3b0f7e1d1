#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p gpurun_out/r02k
( time python bench.py > gpurun_out/r02k/bench_default.json 2> gpurun_out/r02k/bench_default.err ) 2>&1 | grep real
python - <<PY
import json
d=json.load(open("gpurun_out/r02k/bench_default.json"))
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data")})
print(d["config"]["workload"][:80]); r=d["roofline"]; print({k:r[k] for k in r if k not in ("traffic_detail","note")}); print(d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print({a:(v.get("ns_per_day")) for a,v in d["md"].items() if isinstance(v,dict)})
PY
tail -2 gpurun_out/r02k/bench_default.err
