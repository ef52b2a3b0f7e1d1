#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_gpu_mol.py tests/test_data_wire.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5
timeout 300 python scripts/mol_timing.py 256 2>&1 | grep -v "bwd L" | tail -24
timeout 300 python bench.py --steps 100 --warmup 10 --no-pmc --no-sweep --no-md --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
