#!/bin/bash
# PaiNN mixing kernels on the water box: eight waves per tile (SPK_MIX_OCC=8) against the four-wave form (default there)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for OCC in 8 auto; do
SPK_MIX_OCC=$OCC timeout 300 python bench.py --workload water --kind painn --steps 10 --warmup 3 --no-sweep --no-md --no-pmc --no-cpu-baseline > gpurun_out/tp.json 2> gpurun_out/tp.err; echo "OCC=$OCC rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/tp.json"))
print(d["value"], d["ms_per_step"], {a:round(v["avg_us"],1) for a,v in d["kernels"].items() if "mixing" in a or "chain" in a})
PY
done
