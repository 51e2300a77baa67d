OUT=gpurun_out/c35; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "strip" > $OUT/1_strip_tests.log 2>&1; tail -25 $OUT/1_strip_tests.log
