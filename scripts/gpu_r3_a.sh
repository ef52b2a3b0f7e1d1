#!/bin/bash
# round 3, first GPU pass: the new tests (PIMD / bead-parallel / ADVICE regressions), the default bench line (timed), the
# bead-parallel bench mode with two gloo ranks on the one device
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r3a; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_pimd.py tests/test_gpu_train.py tests/test_gpu_reference_callers.py tests/test_gpu_md.py -q -m gpu -p no:cacheprovider --tb=short -x 2>&1 | tail -25 | tee $OUT/pytest_new.log
SECONDS=0
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench rc=$? wall=${SECONDS}s"; tail -5 $OUT/bench_default.err | grep -v amdgpu
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print("schnet", d["value"], d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("kernel","frac","executed_frac_of_peak","traffic","B_min_bytes_per_launch","traffic_over_B_min")})
print("cpu", d["cpu_baseline"])
p=d["painn"]; print("painn", {k: p.get(k) for k in ("value","ms_per_step","error")}); 
if p.get("roofline"): print("   roofline", {k: p["roofline"].get(k) for k in ("kernel","frac","traffic","frac_flag","force_call")}); print("   cpu", p["cpu_baseline"])
if p.get("kernels"):
    for k,v in sorted(p["kernels"].items()): print("   %-26s x%.0f %.1f us frac %s" % (k, v["launches_per_step"], v["avg_us"], v.get("frac_of_peak")))
print("train", json.dumps(d["train"])[:1500])
print("md", json.dumps(d["md"])[:2000])
print("drop_in", d["drop_in"])
PY
SPK_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --mode md --beads 4 --bead-parallel state --kind painn --workload water --water-side 8 --steps 10 --warmup 3 2>&1 | grep -v "^\[Gloo\]" | tail -3 | tee $OUT/bench_bp_state.json
SPK_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --mode md --beads 4 --bead-parallel forces --kind painn --workload water --water-side 8 --steps 10 --warmup 3 2>&1 | grep -v "^\[Gloo\]" | tail -3 | tee $OUT/bench_bp_forces.json
