mkdir -p gpurun_out
python -m pytest tests/test_gpu_stream.py tests/test_gpu_fast_path.py tests/test_gpu_random.py tests/test_gpu_reference_regime.py -q -m gpu -k "not 16000" > gpurun_out/r4_gpu_tests_f.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gpu_tests_f.log
python scripts/gpu_stream_sweep.py stream fp64 2048:100:6 4096:100:6 4096:100:6 4096:20:6 4096:200:2 > gpurun_out/r4_stream_after_cuts.txt 2>&1
