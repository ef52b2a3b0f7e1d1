#!/bin/bash
# quick PaiNN molecule-kernel check: parity tests, cfg-3 bench line, cycle stamps
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_painn_mol.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -5 | tee $OUT/pytest_painn_mol.log
timeout 300 python bench.py --kind painn --steps 100 --warmup 10 --no-cpu-baseline --no-md --no-sweep --no-pmc --no-train --no-drop-in > $OUT/bench_painn.json 2> $OUT/bench_painn.err
python - <<PY
import json
d=json.load(open("$OUT/bench_painn.json"))
print("painn", d["value"], "M edge-msg/s", d["ms_per_step"], "ms/step")
for k,v in sorted(d["kernels"].items()):
    print("   %-26s x%.0f  %.1f us" % (k, v["launches_per_step"], v["avg_us"]))
PY
timeout 300 python scripts/painn_mol_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/painn_mol_cycle_stamps.txt; grep -A24 "bwd 0" $OUT/painn_mol_cycle_stamps.txt | head -30
