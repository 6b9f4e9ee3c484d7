mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_configs.py > gpurun_out/r4_gpu_tests_a.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gpu_tests_a.log
python -m pytest tests/test_gpu_configs.py -q -m gpu > gpurun_out/r4_gpu_tests_b.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gpu_tests_b.log
