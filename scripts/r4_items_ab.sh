# A/B of the item-parallel first pass of an SQP_RTI step in the throughput engine (rti_items, csrc/mpc_stream.h) against the sequential
# passes it replaces (-DMPCB_STREAM_SEQ_RES), alternating on one box; then the stream-engine parity tests
mkdir -p gpurun_out
OUT=gpurun_out/r4_items_ab.txt
: > $OUT
for rep in 1 2; do
  for cfg in "4096:100:6" "2048:100:6" "4096:50:6" "4096:200:3"; do
    echo "--- items      $cfg" >> $OUT
    python scripts/gpu_stream_sweep.py stream fp64 $cfg 2>/dev/null | tail -1 >> $OUT
    echo "--- sequential $cfg" >> $OUT
    MPCB_LIB=robotic_mpc_amd/libmpcbatch_seqres.so python scripts/gpu_stream_sweep.py stream fp64 $cfg 2>/dev/null | tail -1 >> $OUT
  done
done
cat $OUT
python -m pytest tests/test_gpu_stream.py tests/test_gpu_random.py tests/test_gpu_fast_path.py -x -q -m gpu > gpurun_out/r4_items_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_items_tests.log; tail -5 gpurun_out/r4_items_tests.log
