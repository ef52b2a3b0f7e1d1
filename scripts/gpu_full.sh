#!/bin/bash
# whole GPU suite (what the driver runs at round end) + smoke
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/${1:-full}; mkdir -p $OUT
SECONDS=0
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -30 | tee $OUT/pytest_gpu_tail.log
echo "pytest wall ${SECONDS}s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
