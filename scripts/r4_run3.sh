mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r4_gpu_tests_full.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gpu_tests_full.log
python tests/tools/gpu_fast_path_ab.py > gpurun_out/r4_fast_ab2.log 2>&1
python bench.py > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err
