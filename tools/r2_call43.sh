OUT=gpurun_out/c43; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv_auto_tile or cout1 or conv_every" > $OUT/1_tests.log 2>&1; tail -2 $OUT/1_tests.log
timeout 900 python tools/conv_sweep.py --objects 3 --out $OUT/conv_sweep > $OUT/2_sweep.log 2>&1; grep -E "^\s+(77760|4860|1620)\s+1\s" $OUT/2_sweep.log | head; tail -1 $OUT/2_sweep.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > $OUT/3_parity.log 2>&1; tail -2 $OUT/3_parity.log
