#!/bin/bash
# HBM traffic of the hot kernels from the L2 memory-side counters, one --pmc pass per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; kernel-trace only, as gpurun requires).
# Usage: bash scripts/gpu_pmc_traffic.sh <tag> <kind> [workload]
TAG=${1:-pmc}; KIND=${2:-schnet}; WL=${3:-aspirin}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$C -o p -- \
    python $ROOT/bench.py --kind $KIND --workload $WL --steps 3 --warmup 2 --no-graph --no-cpu-baseline > $OUT/pmc_${KIND}_${WL}_$C.log 2>&1
  echo "pmc $C rc=$?"
done
python - <<PY
import csv, glob, collections, json, re
TAGS = [("cfconv_fwd_pair", r"k_cfconv_pair<.*false, false, false>|k_cfconv_pair<[^>]*false, false>"),
        ("cfconv_bwd_pair_gs", r"k_cfconv_pair_t<"), ("cfconv_bwd_pair", r"k_cfconv_pair<.*true"),
        ("cfconv_fwd_mfma", r"k_cfconv_mfma<[^>]*false"), ("cfconv_bwd_mfma", r"k_cfconv_mfma<[^>]*true"),
        ("painn_msg_fwd_row", r"k_painn_msg_row<\d+, \d+, false, false, false>"), ("painn_msg_bwd_row", r"k_painn_msg_row<\d+, \d+, true, false, false>"),
        ("painn_msg_fwd_row_mu0", r"k_painn_msg_row<\d+, \d+, false, false, true>"), ("painn_msg_bwd_row_geom", r"k_painn_msg_row<\d+, \d+, true, true"),
        ("painn_msg_fwd_tile", r"k_painn_msg_tile<"), ("painn_msg_bwd_tile", r"k_painn_msg_tile_bwd<\d+, \d+, false"),
        ("painn_msg_bwd_tile_geom", r"k_painn_msg_tile_bwd<\d+, \d+, true"),
        ("dense_chain", r"k_dense_chain"), ("scatter_add_segsum", r"k_segsum<4>")]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != c: continue
            name = row["Kernel_Name"]
            for tag, pat in TAGS:
                if re.search(pat, name):
                    acc[tag] += float(row["Counter_Value"]); cnt[tag] += 1
                    res[tag]["kernel_name"] = name[:90]
                    break
    for tag in acc:
        res[tag][c + "_raw_per_launch"] = acc[tag] / cnt[tag]
        res[tag]["launches_" + c] = cnt[tag]
out = {"kind": "$KIND", "workload": "$WL", "counters": res,
       "units": "rocprofv3 derived metrics FETCH_SIZE / WRITE_SIZE are in KiB per dispatch (summed over XCDs)",
       "corrections": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> x2 for wide coalesced streaming reads "
                      "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated -> reported as is; the scatter_add_segsum "
                      "row (known 39.9 MB read + 2.75 MB written per launch at cfg 2) is the in-run calibration"}
json.dump(out, open("$OUT/pmc_traffic_${KIND}_${WL}.json", "w"), indent=1)
for tag, v in res.items(): print(tag, {k: (round(x, 1) if isinstance(x, float) else x) for k, x in v.items() if k != "kernel_name"})
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
