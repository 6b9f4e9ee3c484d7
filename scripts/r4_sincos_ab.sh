# A/B of the joint-angle sincos (csrc/mpc_kin.h sincos_joint) against the library's (-DMPCB_LIBM_SINCOS), alternating, one box: both engines, SQP_RTI and full SQP
mkdir -p gpurun_out
OUT=gpurun_out/r4_sincos_ab.txt
: > $OUT
run() { # label lib engine solver cfg
  echo "--- $1 $3 $4 $5" >> $OUT
  MPCB_LIB=$2 SWEEP_SOLVER=$4 python scripts/gpu_stream_sweep.py $3 fp64 $5 2>/dev/null | tail -1 | cut -c1-175 >> $OUT
}
for rep in 1 2; do
  for spec in "latency SQP_RTI 256:100:6" "stream SQP_RTI 4096:100:6" "stream SQP_RTI 2048:100:6" "latency SQP_RTI 512:100:6" "latency SQP_RTI 256:200:3" "latency SQP 512:100:3" "stream SQP 4096:100:3"; do
    set -- $spec
    run "joint" robotic_mpc_amd/libmpcbatch.so $1 $2 $3
    run "libm " robotic_mpc_amd/libmpcbatch_libmsc.so $1 $2 $3
  done
done
cat $OUT
python -m pytest tests -x -q -m gpu > gpurun_out/r4_sincos_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_sincos_tests.log; tail -4 gpurun_out/r4_sincos_tests.log
