#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_ops.py tests/test_gpu_models.py -x -q -k "train or matmul or gemm" 2>&1 | grep -v Warning | tail -4
for KIND in schnet painn; do timeout 600 python bench.py --mode train --kind $KIND --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done
