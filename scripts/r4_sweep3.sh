# engine crossovers once more at the round's last kernels (joint-angle sincos, merit pass without spills): full SQP 2048..4096 and SQP_RTI 1024..1536
mkdir -p gpurun_out
O=gpurun_out/r4_engine_sweep3_raw.txt
echo "# full SQP, 600 steps, latency engine" > $O
SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py latency fp64 2048:100:6 2560:100:6 3072:100:6 3584:100:6 4096:100:6 2>/dev/null | cut -c1-175 >> $O
echo "# full SQP, 600 steps, throughput engine" >> $O
SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py stream fp64 2048:100:6 2560:100:6 3072:100:6 3584:100:6 4096:100:6 2>/dev/null | cut -c1-175 >> $O
echo "# full SQP, 200 steps" >> $O
SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py latency fp64 3072:100:2 4096:100:2 2>/dev/null | cut -c1-175 >> $O
SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py stream fp64 3072:100:2 4096:100:2 2>/dev/null | cut -c1-175 >> $O
echo "# SQP_RTI, latency / throughput engine" >> $O
python scripts/gpu_stream_sweep.py latency fp64 1024:100:6 1280:100:6 1536:100:6 2>/dev/null | cut -c1-175 >> $O
python scripts/gpu_stream_sweep.py stream fp64 1024:100:6 1280:100:6 1536:100:6 4096:100:6 2>/dev/null | cut -c1-175 >> $O
cat $O
