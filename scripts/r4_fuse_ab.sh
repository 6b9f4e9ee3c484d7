# A/B of the folded commit / right-hand-side passes (MPCB_FUSE, csrc/mpc_core.h): default build against -DMPCB_FUSE=0, alternating, one box
mkdir -p gpurun_out
OUT=gpurun_out/r4_fuse_ab.txt
: > $OUT
for rep in 1 2; do
  for cfg in "256 100 6.0" "512 100 6.0" "256 125 6.0" "256 200 3.0" "256 50 6.0" "1024 100 6.0"; do
    echo "--- fused    $cfg" >> $OUT
    python scripts/gpu_quick.py $cfg 2>/dev/null | tail -1 >> $OUT
    echo "--- separate $cfg" >> $OUT
    MPCB_LIB=robotic_mpc_amd/libmpcbatch_nofuse.so python scripts/gpu_quick.py $cfg 2>/dev/null | tail -1 >> $OUT
  done
done
cat $OUT
