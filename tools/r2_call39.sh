OUT=gpurun_out/c42; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/1_gpu_suite.log 2>&1; tail -3 $OUT/1_gpu_suite.log
timeout 600 python bench.py > $OUT/2_bench.json 2> $OUT/2_bench.err; tail -c 600 $OUT/2_bench.json
bash tools/profile_round.sh r02 > $OUT/3_profile.log 2>&1; tail -3 $OUT/3_profile.log
