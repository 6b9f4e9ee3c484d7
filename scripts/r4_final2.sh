# Round 4, second session: evidence at the session's last kernel commit (run by gpurun; outputs under gpurun_out/, copied to profiles/r04b_*)
mkdir -p gpurun_out
python bench.py > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err; echo "bench rc=$?"
COMMIT=$(cat .commit_id 2>/dev/null || echo unknown) bash scripts/profile_gpu.sh > gpurun_out/r4b_profile_gpu.log 2>&1; echo "profile_gpu rc=$?"
bash scripts/prof_counters.sh stream4096 -- python3 /root/repo/scripts/gpu_stream_sweep.py stream fp64 4096:100:6 > gpurun_out/r4b_prof_stream.log 2>&1; echo "prof_counters rc=$?"
python scripts/gpu_prof.py 256 100 6.0 > gpurun_out/r4b_device_breakdown.txt 2>&1
python scripts/gpu_pass_breakdown.py > gpurun_out/r4b_pass_breakdown.txt 2>&1
TSIM=6.0 python scripts/gpu_stream_prof.py 2048 4096 > gpurun_out/r4b_stream_pass_breakdown.txt 2>&1
tail -c 300 gpurun_out/r4b_bench.json
