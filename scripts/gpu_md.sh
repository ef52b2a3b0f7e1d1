#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mol.py tests/test_gpu_models.py tests/test_gpu_md.py tests/test_gpu_torchscript.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 50 --warmup 5 --no-sweep --no-pmc --cpu-reps 2 --md-steps 400 > gpurun_out/t3.json 2> gpurun_out/t3.err; echo rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/t3.json"))
print(d["value"], d["ms_per_step"], d["cpu_baseline"].get("parity_rel_forces"))
print(json.dumps(d["md"], indent=0))
PY
tail -3 gpurun_out/t3.err
