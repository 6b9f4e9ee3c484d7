mkdir -p gpurun_out
O=gpurun_out/r4_geometry.txt
echo "# configs[1]-like, fast path on: wavefronts per simulation (one simulation per CU)" > $O
for W in 8 4; do
  echo "## MPCB_WAVES_PER_SIM=$W" >> $O
  MPCB_WAVES_PER_SIM=$W python scripts/gpu_stream_sweep.py latency fp64 256:100:6 256:100:6 256:200:3 256:300:2 128:200:6 256:80:6 256:125:4 >> $O 2>&1
done
echo "## default" >> $O
python scripts/gpu_stream_sweep.py latency fp64 256:100:6 256:50:6 256:20:6 >> $O 2>&1
MPCB_WAVES_PER_SIM=8 python scripts/gpu_stream_sweep.py latency fp64 256:50:6 256:20:6 >> $O 2>&1
