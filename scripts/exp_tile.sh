#!/bin/bash
# round 4: tuning experiments on the existing box kernels of the PaiNN message (water box): XCD-contiguous walks (tile forward, row
# kernels), 3 / 4 waves per SIMD for the tile forward
OUT=gpurun_out/${1:-r04s}; mkdir -p $OUT
for cfg in "SPK_ROW_LT=0" "SPK_ROW_LT=1"; do
  echo "== $cfg" | tee -a $OUT/exp_tile.txt
  env $cfg EXP_ORDER_ONLY_LATTICE=1 EXP_NO_BLOCKS=1 timeout 300 python scripts/exp_order.py 2>&1 | grep "^lattice" | cut -c1-200 | tee -a $OUT/exp_tile.txt
done
