#!/bin/bash
# BASELINE configs[4]: N=300 long horizon, SQP_RTI, Riccati in fp64 vs fp32, batch {64,256,1024}, one MI355X.
# Throughput engine for both precisions (same code, FT = double | float) with rocprofv3 counters; the latency engine's
# fp64 rate at the same sizes beside it.  Output: gpurun_out/config5/*.json + table.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/config5
T=${CONFIG5_T:-1.5}
for prec in fp64 fp32; do
  for B in 64 256 1024; do
    timeout -k 10 400 bash $R/scripts/prof_counters.sh c5_${prec}_$B -- python3 $R/scripts/gpu_stream_sweep.py stream $prec $B:300:$T > $R/gpurun_out/config5/${prec}_$B.json 2> $R/gpurun_out/config5/${prec}_$B.err
    grep "steps/s" $R/gpurun_out/prof_c5_${prec}_$B/kt.log | grep -v amdgpu > $R/gpurun_out/config5/${prec}_$B.rate
    echo "done $prec $B"
  done
done
cd $R && python3 scripts/gpu_stream_sweep.py latency fp64 64:300:$T 256:300:$T 1024:300:$T 2>&1 | grep "steps/s" > gpurun_out/config5/latency_fp64.rate
python3 - <<'PY'
import json, os, re
R = os.environ.get("GRAFT_REPO_ROOT", ".")
d = os.path.join(R, "gpurun_out", "config5")
rows = []
for prec in ("fp64", "fp32"):
    for B in (64, 256, 1024):
        s = json.load(open(os.path.join(R, "gpurun_out", f"prof_c5_{prec}_{B}", "summary.json")))
        k = [v for n, v in s.items() if "mpc_stream_kernel" in n][0]
        rate = open(os.path.join(d, f"{prec}_{B}.rate")).read()
        m = re.search(r"([\d.]+) ms\s+(\d+) steps/s qp_it ([\d.]+) fail (\d+)", rate)
        ms, sps, qp, fail = float(m.group(1)), int(m.group(2)), float(m.group(3)), int(m.group(4))
        hbm = k["hbm_bytes_per_launch"]
        rows.append(dict(prec=prec, B=B, ms=ms, steps_per_s=sps, qp_it=qp, fail=fail, hbm_GB=hbm / 1e9,
                         hbm_GBps=hbm / (k["kernel_trace"]["avg_ns"] * 1e-9) / 1e9, fetch_GB=2 * k["FETCH_SIZE"] * 1024 / 1e9,
                         write_GB=k["WRITE_SIZE"] * 1024 / 1e9, waves=k["SQ_WAVES"],
                         active=k["SQ_ACTIVE_INST_ANY"] / k["SQ_WAVE_CYCLES"], wait=k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"],
                         valu=k["SQ_ACTIVE_INST_VALU"] / k["SQ_WAVE_CYCLES"], insts_valu=k["SQ_INSTS_VALU"], insts_lds=k["SQ_INSTS_LDS"],
                         bank_conf=k["SQ_LDS_BANK_CONFLICT"] / max(k["SQ_LDS_IDX_ACTIVE"], 1)))
with open(os.path.join(d, "table.txt"), "w") as f:
    f.write("BASELINE configs[4]: UR10, N=300, SQP_RTI, dt=0.01, throughput engine (mpc_stream_kernel<double|float>), one MI355X\n")
    f.write("prec  batch  kernel_ms  steps/s   qp_it  fail  FETCHx2_GB  WRITE_GB  HBM_GB/s  frac_of_8TB/s  SQ_WAVES  active  wait   valu   LDS_conflict\n")
    for r in rows:
        f.write(f"{r['prec']}  {r['B']:5d}  {r['ms']:9.1f}  {r['steps_per_s']:8d}  {r['qp_it']:.2f}  {r['fail']:4d}  {r['fetch_GB']:10.1f}  {r['write_GB']:8.1f}  "
                f"{r['hbm_GBps']:8.0f}  {r['hbm_GBps']/8000:13.3f}  {int(r['waves']):8d}  {r['active']:.2f}    {r['wait']:.2f}   {r['valu']:.2f}   {r['bank_conf']:.3f}\n")
    f.write("\nlatency engine (mpc_rollout_kernel), fp64, same workloads:\n")
    f.write(open(os.path.join(d, "latency_fp64.rate")).read())
json.dump(rows, open(os.path.join(d, "table.json"), "w"), indent=1)
print(open(os.path.join(d, "table.txt")).read())
PY
