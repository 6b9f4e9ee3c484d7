mkdir -p gpurun_out
python scripts/gpu_config2_shares.py > gpurun_out/r4_config2_shares.txt 2>&1
python -m pytest tests/test_gpu_reference_regime.py -q -m gpu -k 16000 > gpurun_out/r4_gpu_tests_e.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gpu_tests_e.log
python scripts/gpu_prof.py 256 100 6.0 > gpurun_out/r4_device_breakdown2.txt 2>&1
