#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p gpurun_out/r02k
timeout 300 python scripts/mol_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02k/mol_timing.txt; tail -45 gpurun_out/r02k/mol_timing.txt
