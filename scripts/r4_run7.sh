mkdir -p gpurun_out
python -m pytest tests -q -m gpu --durations=6 > gpurun_out/r4_gpu_tests_full2.log 2>&1; echo "rc=$?" >> gpurun_out/r4_gpu_tests_full2.log
python bench.py > gpurun_out/r4_bench3.json 2> gpurun_out/r4_bench3.err
python bench.py --workload config2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4_bench_config2.json 2> gpurun_out/r4_bench_config2.err
python bench.py --workload config2 --results summary --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4_bench_config2_summary.json 2> gpurun_out/r4_bench_config2_summary.err
python bench.py --gpus 4 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4_bench_4rank_gloo.json 2> gpurun_out/r4_bench_4rank_gloo.err
python bench.py --workload config2 --gpus 4 --backend gloo --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r4_bench_config2_4rank_gloo.json 2> gpurun_out/r4_bench_config2_4rank_gloo.err
