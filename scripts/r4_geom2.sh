# four vs eight wavefronts per simulation again (one simulation per CU), after the joint-angle sincos removed the 256-register builds' spills
mkdir -p gpurun_out
OUT=gpurun_out/r4_geometry2.txt
: > $OUT
for rep in 1 2; do
  for cfg in "256 100 6.0" "256 80 6.0" "256 50 6.0" "256 112 6.0"; do
    for w in 4 8; do
      echo "--- waves $w: $cfg" >> $OUT
      MPCB_WAVES_PER_SIM=$w python scripts/gpu_quick.py $cfg 2>/dev/null | tail -1 | cut -c1-200 >> $OUT
    done
  done
done
cat $OUT
