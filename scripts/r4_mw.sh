for rep in 1 2 3 4; do
  echo "--- blocked blend 4096:100:6"; SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py stream fp64 4096:100:6 2>/dev/null | tail -1 | sed "s/{.waves.*}//" | cut -c1-120
  echo "--- per-entry     4096:100:6"; MPCB_LIB=robotic_mpc_amd/libmpcbatch_prev.so SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py stream fp64 4096:100:6 2>/dev/null | tail -1 | sed "s/{.waves.*}//" | cut -c1-120
done
