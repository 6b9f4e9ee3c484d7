mkdir -p gpurun_out
python tests/tools/gpu_full_parity.py > gpurun_out/r4_full_parity.txt 2>&1
