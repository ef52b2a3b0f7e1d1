#!/bin/bash
# round 5: per-launch durations of the row-chain kernel inside one eager training step (rocprofv3 kernel trace) + the recorded stage lists
set -x
mkdir -p gpurun_out/r5d
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for kind in painn schnet; do
  SPK_FM_CHAIN=1 SPK_FM_CHAIN_DEBUG=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5d/prof_$kind -o t -- python bench.py --mode train --kind $kind --steps 2 --warmup 3 --no-graph --no-pmc --no-cpu-baseline --detail /tmp/d.json > /dev/null 2> gpurun_out/r5d/stages_$kind.log
  python - <<P
import csv, glob
f = glob.glob("gpurun_out/r5d/prof_$kind/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ch = [r for r in rows if "k_fm_chain" in r["Kernel_Name"]]
per = len(ch) // 6 if len(ch) >= 6 else len(ch)
print("$kind", "chain launches total", len(ch))
last = ch[-(len(ch) // 6 * 1 if len(ch) >= 6 else len(ch)):]
for r in ch[-30:]:
    print("  grid", r["Grid_Size"], "lds", r.get("LDS_Block_Size"), "us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
P
  grep "fm_chain" gpurun_out/r5d/stages_$kind.log | tail -30
done
