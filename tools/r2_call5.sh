OUT=gpurun_out/c5; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "dma" > $OUT/1_kernel_tests.log 2>&1; tail -3 $OUT/1_kernel_tests.log
for t in 60 62 70 61 63 71 66 72; do
  for cfg in "1 360 216 128 128 1" "1 360 216 128 128 3" "1 360 216 512 128 3"; do
    echo "tile $t cfg $cfg: $(python tools/one_conv.py $cfg $t 5 2>&1 | tail -1)" >> $OUT/2_ksweep.txt
  done
done
for t in 67 66 72 63 71; do
  for cfg in "3 30 54 256 256 1" "3 30 54 256 256 3" "3 30 54 1024 256 3" "1 30 54 1024 256 1"; do
    echo "tile $t cfg $cfg: $(python tools/one_conv.py $cfg $t 5 2>&1 | tail -1)" >> $OUT/2_ksweep.txt
  done
done
cat $OUT/2_ksweep.txt
CUTIE_AMD_EXPERIMENTAL_TILES=1 timeout 900 python tools/conv_sweep.py --objects 3 --out $OUT/conv_sweep > $OUT/3_sweep.log 2>&1; tail -2 $OUT/3_sweep.log
