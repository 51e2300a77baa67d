OUT=gpurun_out/c17; mkdir -p $OUT
for v in FULL EMPTY NO_DMA DMA_ONLY DMA_ONLY_NOBAR; do
  L=""; if [ $v != FULL ]; then L=$PWD/tools/abl/libcutie_hip_$v.so; fi
  CUTIE_AMD_LIB=$L timeout 900 python tools/conv_sweep.py --objects 3 --out $OUT/sweep_$v > $OUT/sweep_$v.log 2>&1; tail -1 $OUT/sweep_$v.log
done
timeout 120 tools/abl/dmabench > $OUT/dmabench.log 2>&1; tail -30 $OUT/dmabench.log
