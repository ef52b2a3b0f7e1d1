#!/bin/bash
# A/B of a tuning switch (env var) of the PaiNN molecule kernels on ONE box: cycle stamps + event timings per setting
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/${1:-ab}; mkdir -p $OUT
VAR=${2:-SPK_PM_ASSIGN}; shift; shift
for V in "$@"; do
  echo "=== $VAR=$V"
  env $VAR=$V timeout 300 python scripts/painn_mol_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/stamps_$V.txt
  grep -E "bwd 7 |bwd 8 |L2 P3|painn_mol" $OUT/stamps_$V.txt | cut -c1-200
  grep -A9 "bwd message" $OUT/stamps_$V.txt | cut -c1-200 | tail -8 | awk '{print $1,$2,$NF}'
done
