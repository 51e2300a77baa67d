OUT=gpurun_out/c22; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/1_gpu_suite.log 2>&1; tail -3 $OUT/1_gpu_suite.log
timeout 600 python bench.py > $OUT/2_bench.json 2> $OUT/2_bench.err; python -c "
import json; d=json.loads(open('$OUT/2_bench.json').read().strip().split('\n')[-1]); print({k:d[k] for k in ('value','ms_per_step','value_no_lookahead')}, d['full_bank'], d['roofline']['frac'], d['roofline_affinity'].get('matmul',{}).get('stage_us'), d.get('multi_clip'))"
