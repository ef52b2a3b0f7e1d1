#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mol.py tests/test_gpu_models.py -x -q 2>&1 | tail -4
timeout 300 python scripts/mol_timing.py 2>&1 | grep -v amdgpu.ids | tail -48
timeout 600 python bench.py --steps 50 --warmup 5 --no-sweep --no-md --no-pmc --cpu-reps 2 > gpurun_out/t2.json 2> gpurun_out/t2.err; echo rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/t2.json"))
print(d["value"], d["ms_per_step"], d["cpu_baseline"].get("parity_rel_forces"), {a:round(v["avg_us"],1) for a,v in d["kernels"].items()})
PY
