# Round 4, second session: parity evidence and end-to-end lines at the session's last kernel commit
mkdir -p gpurun_out
python tests/tools/gpu_full_parity.py > gpurun_out/r4b_full_parity.txt 2>&1; echo "full parity rc=$?"
python tests/tools/gpu_random_soak.py > gpurun_out/r4b_random_soak.txt 2>&1; echo "soak rc=$?"
python scripts/gpu_config2_shares.py > gpurun_out/r4b_config2_shares.txt 2>&1; echo "shares rc=$?"
python bench.py --workload config2 --no-cpu-baseline > gpurun_out/r4b_bench_config2_one_gpu.json 2> gpurun_out/r4b_bench_config2.err; echo "config2 rc=$?"
python bench.py --workload config2 --results summary --no-cpu-baseline > gpurun_out/r4b_bench_config2_one_gpu_summary_only.json 2>> gpurun_out/r4b_bench_config2.err; echo "config2 summary rc=$?"
python bench.py --gpus 4 --backend gloo --no-cpu-baseline --no-secondary > gpurun_out/r4b_bench_4rank_gloo.json 2> gpurun_out/r4b_bench_4rank.err; echo "4rank rc=$?"
tail -3 gpurun_out/r4b_full_parity.txt | cut -c1-250
