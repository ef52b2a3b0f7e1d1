#!/bin/bash
# Round 6: hollow-out timing of the row-tile backward (what is the time: fixed per-chunk cost, gathers, GEMMs?) -- rebuilds spk_painn_tile on the box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
for H in ${HOLLOWS:-0 1 2 3}; do
  touch schnetpack_amd/csrc/spk_painn_tile.hip
  SPK_EXTRA_FLAGS="-DSPK_RT_HOLLOW=$H" python -m schnetpack_amd.csrc.build > /tmp/build_$H.txt 2>&1 || { echo "build $H failed"; tail -5 /tmp/build_$H.txt; continue; }
  echo "SPK_RT_HOLLOW=$H"; bash scripts/gpu_r06_painn_box.sh r06hollow$H - 2>&1
done
