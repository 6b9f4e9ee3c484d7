mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r4_gpu_tests_final.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gpu_tests_final.log
python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err
COMMIT=$(cat .commit_id 2>/dev/null || echo unknown) bash scripts/profile_gpu.sh > gpurun_out/r4_profile_gpu.log 2>&1; echo "profile_gpu rc=$?" >> gpurun_out/r4_profile_gpu.log
bash scripts/prof_counters.sh stream4096 -- python3 /root/repo/scripts/gpu_stream_sweep.py stream fp64 4096:100:6 > gpurun_out/r4_prof_stream.log 2>&1; echo "prof_counters rc=$?" >> gpurun_out/r4_prof_stream.log
python scripts/gpu_prof.py 256 100 6.0 > gpurun_out/r4_device_breakdown_final.txt 2>&1
python scripts/gpu_pass_breakdown.py > gpurun_out/r4_pass_breakdown.txt 2>&1
