OUT=gpurun_out/c41; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "query_init or linear" > $OUT/1_tests.log 2>&1; tail -3 $OUT/1_tests.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher.py -q -x > $OUT/2_parity.log 2>&1; tail -3 $OUT/2_parity.log
for v in a b; do
timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-roofline > $OUT/3_bench_$v.json 2> $OUT/3_bench_$v.err
python -c "
import json; d=json.loads(open('$OUT/3_bench_$v.json').read().strip().split('\n')[-1]); print('$v:', d['value'], 'fps', d['ms_per_step'], 'no-lookahead', d.get('value_no_lookahead'))"
done
