mkdir -p gpurun_out
O=gpurun_out/r4_engine_sweep_raw.txt
echo "# latency engine, default geometry per batch" > $O
python scripts/gpu_stream_sweep.py latency fp64 128:100:6 256:100:6 320:100:6 384:100:6 448:100:6 512:100:6 768:100:6 1024:100:6 1280:100:6 1536:100:6 2048:100:6 4096:100:6 >> $O 2>&1
echo "# latency engine forced to ONE simulation per CU (MPCB_SIMS_PER_CU=1, grid in rounds)" >> $O
MPCB_SIMS_PER_CU=1 python scripts/gpu_stream_sweep.py latency fp64 320:100:6 384:100:6 512:100:6 1024:100:6 >> $O 2>&1
echo "# throughput engine" >> $O
python scripts/gpu_stream_sweep.py stream fp64 256:100:6 512:100:6 768:100:6 1024:100:6 1280:100:6 1536:100:6 2048:100:6 3072:100:6 4096:100:6 8192:100:3 >> $O 2>&1
echo "# other horizons: latency (default geometry) / throughput" >> $O
python scripts/gpu_stream_sweep.py latency fp64 256:20:6 256:50:6 256:200:3 256:300:2 128:200:6 512:200:3 1024:20:6 1024:50:6 >> $O 2>&1
python scripts/gpu_stream_sweep.py stream fp64 1024:20:6 1024:50:6 4096:20:6 4096:50:6 4096:200:2 1024:300:1.2 >> $O 2>&1
echo "# full SQP" >> $O
SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py latency fp64 256:100:3 512:100:3 2560:100:6 4096:100:6 >> $O 2>&1
SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py stream fp64 2560:100:6 4096:100:6 >> $O 2>&1
