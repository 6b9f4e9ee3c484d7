# Engine crossover re-measured after the item-parallel first pass of the throughput engine (rti_items) and the folded passes of the
# latency engine: SQP_RTI 1024..2560 simulations and full SQP 2048..4096 on both engines, one box
mkdir -p gpurun_out
O=gpurun_out/r4_engine_sweep2_raw.txt
echo "# SQP_RTI, latency engine" > $O
python scripts/gpu_stream_sweep.py latency fp64 1024:100:6 1280:100:6 1536:100:6 1792:100:6 2048:100:6 2560:100:6 >> $O 2>&1
echo "# SQP_RTI, throughput engine" >> $O
python scripts/gpu_stream_sweep.py stream fp64 1024:100:6 1280:100:6 1536:100:6 1792:100:6 2048:100:6 2560:100:6 3072:100:6 4096:100:6 8192:100:3 >> $O 2>&1
echo "# full SQP, latency engine" >> $O
SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py latency fp64 2048:100:6 2560:100:6 3072:100:6 4096:100:6 4096:100:1 >> $O 2>&1
echo "# full SQP, throughput engine (item-parallel NLP residual norms)" >> $O
SWEEP_SOLVER=SQP python scripts/gpu_stream_sweep.py stream fp64 2048:100:6 2560:100:6 3072:100:6 4096:100:6 4096:100:1 >> $O 2>&1
echo "# full SQP, throughput engine, sequential passes (-DMPCB_STREAM_SEQ_RES)" >> $O
SWEEP_SOLVER=SQP MPCB_LIB=robotic_mpc_amd/libmpcbatch_seqres.so python scripts/gpu_stream_sweep.py stream fp64 2560:100:6 4096:100:6 >> $O 2>&1
cat $O
python -m pytest tests -x -q -m gpu > gpurun_out/r4_items_tests2.log 2>&1; echo "rc=$?" >> gpurun_out/r4_items_tests2.log; tail -5 gpurun_out/r4_items_tests2.log
