#!/bin/bash
# round 5: cycle stamps of the row-chain kernel (workgroup 0), eager steps
set -x
mkdir -p gpurun_out/r5f
cd "$GRAFT_REPO_ROOT"
for kind in schnet painn; do
  SPK_FM_CHAIN=1 SPK_FM_CHAIN_DEBUG=1 SPK_FM_CHAIN_STAMPS=1 timeout 300 python bench.py --mode train --kind $kind --steps 1 --warmup 3 --no-graph --no-pmc --no-cpu-baseline --detail /tmp/d.json > /dev/null 2> gpurun_out/r5f/stamps_$kind.log
  grep "fm_chain" gpurun_out/r5f/stamps_$kind.log | tail -24
done
