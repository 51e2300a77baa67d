# the last validation of round 2: full GPU suite, default bench line, rocprofv3 evidence (-> profiles/r02_*)
OUT=gpurun_out/final; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/1_gpu_suite.log 2>&1; tail -3 $OUT/1_gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/2_smoke.log 2>&1; tail -1 $OUT/2_smoke.log
timeout 600 python bench.py > $OUT/3_bench.json 2> $OUT/3_bench.err; tail -c 300 $OUT/3_bench.json
bash tools/profile_round.sh r02 > $OUT/4_profile.log 2>&1; tail -2 $OUT/4_profile.log
