#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mol.py tests/test_gpu_models.py tests/test_gpu_md.py tests/test_torch_ops.py tests/test_gpu_bench.py tests/test_gpu_reference_callers.py tests/test_gpu_torchscript.py -x -q 2>&1 | grep -v Warning | tail -12
