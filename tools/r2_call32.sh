OUT=gpurun_out/c32; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention_with_fused or attn or aux_mask" > $OUT/1_kernel_tests.log 2>&1; tail -15 $OUT/1_kernel_tests.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher.py tests/test_gpu_small_model.py -q -x > $OUT/2_parity.log 2>&1; tail -3 $OUT/2_parity.log
for v in unfused fused unfused fused; do
U=0; if [ $v = unfused ]; then U=1; fi
CUTIE_AMD_UNFUSED=$U timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-roofline > $OUT/3_bench_$v.json 2> $OUT/3_bench_$v.err
python -c "
import json; d=json.loads(open('$OUT/3_bench_$v.json').read().strip().split('\n')[-1]); print('$v:', d['value'], 'fps', d['ms_per_step'], 'no-lookahead', d.get('value_no_lookahead'))"
done
