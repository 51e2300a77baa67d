OUT=gpurun_out/c6; mkdir -p $OUT
timeout 1200 python tools/conv_sweep.py --objects 3 1 2 --out $OUT/conv_sweep > $OUT/1_sweep.log 2>&1; tail -2 $OUT/1_sweep.log
cp $OUT/conv_sweep_tiles.json cutie_amd/tiles_gfx950.json
timeout 1500 python -m pytest tests -q -m gpu -s > $OUT/2_gpu_suite.log 2>&1; tail -12 $OUT/2_gpu_suite.log | cut -c1-300
timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 > $OUT/3_bench.json 2> $OUT/3_bench.err; cut -c1-1500 $OUT/3_bench.json
timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --no-lookahead --no-roofline > $OUT/3_bench_nola.json 2> $OUT/3_bench_nola.err; cut -c1-300 $OUT/3_bench_nola.json
