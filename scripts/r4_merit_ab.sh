# A/B of the SQP merit pass without / with its parameter block hoisted out of the stage loop and spilled (-DMPCB_NO_LICM_BLOCK), full SQP, both engines
mkdir -p gpurun_out
OUT=gpurun_out/r4_merit_ab.txt
: > $OUT
run() { echo "--- $1 $3 $4" >> $OUT; MPCB_LIB=$2 SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py $3 fp64 $4 2>/dev/null | tail -1 | cut -c1-175 >> $OUT; }
for rep in 1 2; do
  for spec in "latency 512:100:3" "latency 256:100:3" "stream 4096:100:3" "latency 2048:100:3"; do
    set -- $spec
    run "loads in the loop" robotic_mpc_amd/libmpcbatch.so $1 $2
    run "hoisted + spilled" robotic_mpc_amd/libmpcbatch_nolicm.so $1 $2
  done
done
cat $OUT
python -m pytest tests -x -q -m gpu -k "sqp or SQP or config3 or random or parity" > gpurun_out/r4_merit_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_merit_tests.log; tail -3 gpurun_out/r4_merit_tests.log
