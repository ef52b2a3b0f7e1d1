#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 300 python scripts/mol_timing.py 2>&1 | grep -v amdgpu.ids | grep -E "^block|schnet_mol"
