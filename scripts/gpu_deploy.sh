#!/bin/bash
# deployment runtime: tests + PCIe-inclusive timing of the plain-C driver
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/deploy
timeout 900 python -m pytest tests/test_deploy.py -x -q 2>&1 | tail -25 | tee gpurun_out/deploy/pytest.log
timeout 600 python scripts/deploy_timing.py > gpurun_out/deploy/timing.json 2> gpurun_out/deploy/timing.err
tail -5 gpurun_out/deploy/timing.err
cat gpurun_out/deploy/timing.json
