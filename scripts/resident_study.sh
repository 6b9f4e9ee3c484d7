#!/bin/bash
# "LDS-resident vs HBM-spilled stage factors" (BASELINE configs[4] / SURVEY 8d config 5, third leg), latency engine, one MI355X.
# The fp64 factor the solve sweeps read (K, R~^-1 h_u, e, p: 102 doubles per stage) + the chunk transition matrices fit the
# 152 KiB pool up to N ~ 125: fully resident at N = 100 (the headline horizon).  At N = 300 not even an all-fp32 copy fits
# (301 x 102 x 4 B = 123 KB + 18 KB Phi + >= 32 KB of chunk buffers > 155 KB): there the factor is resident ONE SEGMENT of 112
# transitions at a time (it goes through HBM once, the sweeps load it back per segment and run chunk-parallel inside it).
# Legs: N = 100, batch {64, 256, 1024} and N = 300, batch {64, 256}, one simulation per CU:
#   resident = libmpcbatch.so,  spilled = the same source built with -DMPCB_NO_RESIDENT (streaming sweeps).
# rocprofv3 counters per leg (separate --pmc passes): FETCH / WRITE, SQ_WAVES, wait / active, VALU, LDS bank conflicts.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/resident
T=${RES_T:-3.0}
export MPCB_SIMS_PER_CU=1
for leg in resident spilled; do
  if [ $leg = spilled ]; then export MPCB_LIB=$R/robotic_mpc_amd/libmpcbatch_nores.so; else unset MPCB_LIB; fi
  for spec in 64:100:$T 256:100:$T 1024:100:$T 64:300:1.5 256:300:1.5; do
    B=${spec//:/_}
    timeout -k 10 400 bash $R/scripts/prof_counters.sh res_${leg}_$B -- python3 $R/scripts/gpu_stream_sweep.py latency fp64 $spec > $R/gpurun_out/resident/${leg}_$B.json 2> $R/gpurun_out/resident/${leg}_$B.err
    grep "steps/s" $R/gpurun_out/prof_res_${leg}_$B/kt.log | grep -v amdgpu > $R/gpurun_out/resident/${leg}_$B.rate
    echo "done $leg $B"
  done
done
unset MPCB_SIMS_PER_CU
cd $R
for leg in resident spilled; do
  if [ $leg = spilled ]; then export MPCB_LIB=$R/robotic_mpc_amd/libmpcbatch_nores.so; else unset MPCB_LIB; fi
  python3 scripts/gpu_stream_sweep.py latency fp64 256:200:2 1024:300:1.5 1024:100:$T 2>&1 | grep "steps/s" > gpurun_out/resident/latency_other_$leg.rate
done
python3 - <<'PY'
import json, os, re
R = os.environ.get("GRAFT_REPO_ROOT", ".")
d = os.path.join(R, "gpurun_out", "resident")
rows = []
for leg in ("resident", "spilled"):
    for B in ("64_100_" + os.environ.get("RES_T", "3.0"), "256_100_" + os.environ.get("RES_T", "3.0"), "1024_100_" + os.environ.get("RES_T", "3.0"), "64_300_1.5", "256_300_1.5"):
        s = json.load(open(os.path.join(R, "gpurun_out", f"prof_res_{leg}_{B}", "summary.json")))
        k = [v for n, v in s.items() if "mpc_rollout_kernel" in n][0]
        rate = open(os.path.join(d, f"{leg}_{B}.rate")).read()
        m = re.search(r"([\d.]+) ms\s+(\d+) steps/s qp_it ([\d.]+) fail (\d+)", rate)
        ms, sps, qp, fail = float(m.group(1)), int(m.group(2)), float(m.group(3)), int(m.group(4))
        hbm = k["hbm_bytes_per_launch"]
        rows.append(dict(leg=leg, B=B, ms=ms, steps_per_s=sps, qp_it=qp, fail=fail, hbm_GB=hbm / 1e9,
                         hbm_GBps=hbm / (k["kernel_trace"]["avg_ns"] * 1e-9) / 1e9, fetch_GB=2 * k["FETCH_SIZE"] * 1024 / 1e9,
                         write_GB=k["WRITE_SIZE"] * 1024 / 1e9, waves=k["SQ_WAVES"],
                         active=k["SQ_ACTIVE_INST_ANY"] / k["SQ_WAVE_CYCLES"], wait=k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"],
                         valu=k["SQ_ACTIVE_INST_VALU"] / k["SQ_WAVE_CYCLES"], insts_valu=k["SQ_INSTS_VALU"], insts_lds=k["SQ_INSTS_LDS"],
                         bank_conf=k["SQ_LDS_BANK_CONFLICT"] / max(k["SQ_LDS_IDX_ACTIVE"], 1)))
with open(os.path.join(d, "table.txt"), "w") as f:
    f.write("LDS-resident vs HBM-spilled stage factors: UR10, SQP_RTI, dt=0.01, latency engine (one simulation per CU), one MI355X; workload = batch_N_seconds\n")
    f.write("(N=100: the whole factor resident; N=300: resident one segment of 112 transitions at a time)\n")
    f.write("leg       workload     kernel_ms  steps/s   qp_it  fail  FETCHx2_GB  WRITE_GB  HBM_GB/s  SQ_WAVES  active  wait   valu   LDS_conflict  VALU_insts  LDS_insts\n")
    for r in rows:
        f.write(f"{r['leg']:9s} {r['B']:12s}  {r['ms']:9.1f}  {r['steps_per_s']:8d}  {r['qp_it']:.2f}  {r['fail']:4d}  {r['fetch_GB']:10.1f}  {r['write_GB']:8.1f}  "
                f"{r['hbm_GBps']:8.0f}  {int(r['waves']):8d}  {r['active']:.2f}    {r['wait']:.2f}   {r['valu']:.2f}   {r['bank_conf']:.3f}         {r['insts_valu']:.3g}  {r['insts_lds']:.3g}\n")
    for leg in ("resident", "spilled"):
        f.write(f"\nother workloads, default launch geometry (beyond 256 simulations: two per CU, streaming sweeps in both builds), {leg} build:\n")
        f.write(open(os.path.join(d, f"latency_other_{leg}.rate")).read())
json.dump(rows, open(os.path.join(d, "table.json"), "w"), indent=1)
print(open(os.path.join(d, "table.txt")).read())
PY
