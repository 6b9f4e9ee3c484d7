mkdir -p gpurun_out
python -m pytest tests/test_gpu_reference_regime.py tests/test_gpu_fast_path.py tests/test_gpu_parity.py -q -m gpu --durations=8 > gpurun_out/r4_gpu_tests_d.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gpu_tests_d.log
