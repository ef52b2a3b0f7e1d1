#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for OCC in 2 1; do
echo "== SPK_MIX_OCC=$OCC"
SPK_MIX_OCC=$OCC timeout 600 python bench.py --kind painn --steps 100 --warmup 10 --no-pmc --no-sweep --no-md --no-cpu-baseline > gpurun_out/p_$OCC.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/p_$OCC.json"))
print(d["value"], d["ms_per_step"])
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["us_per_step"])[:9]: print("  %-28s %6.2f x %6.1f us = %7.1f" % (k, v["launches_per_step"], v["avg_us"], v["us_per_step"]))
PY
done
echo "== water painn OCC 2 vs 1"
for OCC in 2 1; do SPK_MIX_OCC=$OCC timeout 600 python bench.py --kind painn --workload water --steps 20 --warmup 3 --no-pmc --no-sweep --no-md --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print($OCC, d['value'], d['ms_per_step'], {k: round(v['avg_us'],1) for k,v in d['kernels'].items() if 'mixing' in k})"; done
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -k "training_mode" 2>&1 | tail -5 | cut -c1-300
