OUT=gpurun_out/c44; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv_auto_tile" > $OUT/1_tests.log 2>&1; tail -4 $OUT/1_tests.log
timeout 900 python tools/conv_sweep.py --objects 3 --out $OUT/conv_sweep > $OUT/2_sweep.log 2>&1; grep -E "^\s+(77760|4860|1620)\s+1\s" $OUT/2_sweep.log | head
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher.py -q -x > $OUT/3_parity.log 2>&1; tail -2 $OUT/3_parity.log
for v in a b; do
timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-roofline > $OUT/3_bench_$v.json 2> $OUT/3_bench_$v.err
python -c "
import json; d=json.loads(open('$OUT/3_bench_$v.json').read().strip().split('\n')[-1]); print('$v:', d['value'], 'fps', d['ms_per_step'], 'no-lookahead', d.get('value_no_lookahead'))"
done
