mkdir -p gpurun_out
python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_stream.py -q -m gpu > gpurun_out/r4_gpu_tests_c.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gpu_tests_c.log
python scripts/gpu_prof.py 256 100 6.0 > gpurun_out/r4_device_breakdown.txt 2>&1
