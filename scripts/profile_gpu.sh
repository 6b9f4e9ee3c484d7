#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then two separate
# PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950). Summaries land in gpurun_out/prof/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
COMMIT=${COMMIT:-unknown}
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $ARGS > $OUT/bench_kt.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.log 2>&1 || exit 3
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/bench_sq.log 2>&1 || echo "sq pass failed (non fatal)"
cd $R
COMMIT=$COMMIT python3 - <<'PY'
import csv, glob, os, json, statistics
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "prof")
# one value per (dispatch, counter): rocprofv3 may emit several rows for one dispatch of one counter (one per counter
# dimension / instance) -- they are SUMMED inside the dispatch; the per-launch figure is then the MEDIAN over the dispatches
# (round 3's summary averaged rows, which turned SQ_WAVES 2048 into 2730.67 when one dispatch had two rows)
per = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "mpc_rollout_kernel" not in row.get("Kernel_Name", ""):
                continue
            d = per.setdefault(row["Counter_Name"], {})
            key = (f, row.get("Dispatch_Id", row.get("Correlation_Id", "")))
            d[key] = d.get(key, 0.0) + float(row["Counter_Value"])
res = {k: {"per_launch": statistics.median(v.values()), "launches": len(v), "per_launch_values": sorted(v.values())} for k, v in per.items()}
# workload of the profiled command (bench.py defaults) -- bench.py matches on these before quoting the traffic
res.update({"batch": 256, "N": 100, "Nsim": 600, "solver": "SQP_RTI", "commit": os.environ.get("COMMIT", "unknown"),
            "command": "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"})
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    # MI355X_MICROARCH.md (HBM): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of WIDE (16 B/lane)
    # streaming reads at 64 B -> x2 for those; the 8-byte item loads of the residual / NLP / fast-path passes are counted as they
    # are.  This kernel mixes both (wide: the chunk copies of the factorisation sweep and the segment loads), so the truth lies
    # between the raw and the doubled figure: both are recorded, bench.py quotes the doubled one (upper bound) as roofline.traffic.
    f_, w_ = res["FETCH_SIZE"]["per_launch"], res["WRITE_SIZE"]["per_launch"]
    res["hbm_bytes_per_launch_raw"] = (f_ + w_) * 1024.0
    res["hbm_bytes_per_launch"] = (2.0 * f_ + w_) * 1024.0
    res["correction"] = "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (upper bound, wide-read correction applied to ALL reads); _raw = (FETCH_SIZE + WRITE_SIZE) * 1024; separate --pmc passes"
# kernel-trace average of the same command
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "mpc_rollout_kernel" in row.get("Name", ""):
            res["kernel_trace"] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "name": row["Name"][:60]}
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res))
PY
