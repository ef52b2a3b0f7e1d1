#!/bin/bash
# round 5: the driver's protocol -- full GPU suite, smoke, the driver's bench command, kernel traces of the headline legs
set -x
mkdir -p gpurun_out/r5full
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5full/pytest_gpu_tail.log
cat gpurun_out/r5full/pytest_gpu_tail.log
cp gpurun_out/parity_ledger.json gpurun_out/r5full/parity_ledger.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r5full/bench_driver_command_detail.json > gpurun_out/r5full/bench_driver_command_line.json 2> gpurun_out/r5full/bench_err.log
echo "bench rc=$? bytes=$(wc -c < gpurun_out/r5full/bench_driver_command_line.json)"
timeout 900 python bench.py --detail gpurun_out/r5full/bench_default_detail.json > gpurun_out/r5full/bench_default_line.json 2>> gpurun_out/r5full/bench_err.log
echo "bench default rc=$? bytes=$(wc -c < gpurun_out/r5full/bench_default_line.json)"
for kind in schnet painn; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5full/prof_$kind -o p -- python bench.py --kind $kind --steps 50 --warmup 5 --no-md --no-sweep --no-painn --no-train --no-drop-in --no-pmc --no-cpu-baseline --detail /tmp/x.json > /dev/null 2>&1
  cp gpurun_out/r5full/prof_$kind/p_kernel_stats.csv gpurun_out/r5full/${kind}_kernel_stats.csv 2>/dev/null
  rm -rf gpurun_out/r5full/prof_$kind
done
