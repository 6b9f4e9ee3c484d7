# A/B of the slot flip (an accepted fast-path candidate becomes the QP iterate by an index flip) against the commit copy, throughput engine
mkdir -p gpurun_out
OUT=gpurun_out/r4_flip_ab.txt
: > $OUT
python -m pytest tests/test_gpu_stream.py tests/test_gpu_random.py tests/test_gpu_fast_path.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r4_flip_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4_flip_tests.log; tail -3 gpurun_out/r4_flip_tests.log
for rep in 1 2 3; do
  for cfg in "4096:100:6" "2048:100:6" "1280:100:6"; do
    echo "--- flip $cfg" >> $OUT; python scripts/gpu_stream_sweep.py stream fp64 $cfg 2>/dev/null | tail -1 | sed "s/{.waves.*}//" | cut -c1-110 >> $OUT
    echo "--- copy $cfg" >> $OUT; MPCB_LIB=robotic_mpc_amd/libmpcbatch_copycommit.so python scripts/gpu_stream_sweep.py stream fp64 $cfg 2>/dev/null | tail -1 | sed "s/{.waves.*}//" | cut -c1-110 >> $OUT
  done
done
cat $OUT
