#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
for V in auto pair directed; do
timeout 300 python bench.py --variant $V --steps 50 --warmup 10 --no-cpu-baseline > /tmp/b_$V.json 2>/dev/null
python - <<PY
import json
d=json.load(open("/tmp/b_$V.json"))
print("$V", d["value"], "M edge-msg/s", d["ms_per_step"], "ms/step")
for k,v in sorted(d["kernels"].items()):
    if "cfconv" in k: print("   %-24s x%.0f  %.1f us" % (k, v["launches_per_step"], v["avg_us"]))
PY
done
