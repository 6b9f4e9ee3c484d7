#!/bin/bash
# usage: prof_counters.sh TAG -- program args...   : separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ set) + kernel trace
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- "$@" > $OUT/kt.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- "$@" > $OUT/fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- "$@" > $OUT/write.log 2>&1 || exit 3
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- "$@" > $OUT/sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- "$@" > $OUT/sq2.log 2>&1 || echo "sq2 pass failed"
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, os, json, sys, statistics
out = sys.argv[1]
per = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "mpc_" not in k: continue
        kn = k.split("(")[0][:40]
        d = per.setdefault(kn, {}).setdefault(row["Counter_Name"], {})
        key = (f, row.get("Dispatch_Id", row.get("Correlation_Id", "")))
        d[key] = d.get(key, 0.0) + float(row["Counter_Value"])      # rows of one dispatch are summed ...
res = {kn: {c: statistics.median(v.values()) for c, v in d.items()} for kn, d in per.items()}   # ... the per-launch figure is the median over dispatches
for kn, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0      # wide streaming reads: x2 (MI355X_MICROARCH.md)
        d["hbm_bytes_per_launch_raw"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "mpc_" in row.get("Name", ""):
            res.setdefault(row["Name"].split("(")[0][:40], {})["kernel_trace"] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
