OUT=gpurun_out/c7; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "aff" > $OUT/1_aff_tests.log 2>&1; tail -5 $OUT/1_aff_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher.py -q -x -k "trajectory or 480p or teacher_forced_scenarios" > $OUT/2_parity.log 2>&1; tail -5 $OUT/2_parity.log | cut -c1-300
timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 > $OUT/3_bench.json 2> $OUT/3_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c7/3_bench.json').read().strip().split('\n')[-1])
print(d['value'], 'fps', d['ms_per_step'], 'ms; conv', d['roofline']['ms_per_frame'], 'frac', d['roofline']['frac'])
print(d['roofline_affinity']['ms_per_frame'], d['roofline_affinity']['matmul'])
PY
