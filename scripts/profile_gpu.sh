#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then two separate
# PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950). Summaries land in gpurun_out/prof/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $ARGS > $OUT/bench_kt.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.log 2>&1 || exit 3
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/bench_sq.log 2>&1 || echo "sq pass failed (non fatal)"
cd $R
python3 - <<'PY'
import csv, glob, os, json
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "prof")
summary = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "mpc_rollout_kernel" not in row.get("Kernel_Name", ""):
                continue
            summary.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
res = {k: {"per_launch_mean": sum(v) / len(v), "launches": len(v)} for k, v in summary.items()}
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res))
PY
