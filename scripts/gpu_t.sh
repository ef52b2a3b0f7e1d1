#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_gpu_mol.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5
timeout 300 python scripts/mol_timing.py 256 2>&1 | tail -48
