OUT=gpurun_out/c18; mkdir -p $OUT
for v in FULL EMPTY EMPTY_NOEPI EMPTY_NOLOOP LAUNCH_ONLY; do
  L=""; if [ $v != FULL ]; then L=$PWD/tools/abl/libcutie_hip_$v.so; fi
  CUTIE_AMD_LIB=$L timeout 900 python tools/conv_sweep.py --objects 3 --out $OUT/sweep_$v > $OUT/sweep_$v.log 2>&1; tail -1 $OUT/sweep_$v.log
done
