mkdir -p gpurun_out
bash scripts/prof_counters.sh stream4096 -- python3 /root/repo/scripts/gpu_stream_sweep.py stream fp64 4096:100:6 > gpurun_out/r4_prof_stream.log 2>&1; echo "prof_counters rc=$?" >> gpurun_out/r4_prof_stream.log
