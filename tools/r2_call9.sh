OUT=gpurun_out/c9; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -s > $OUT/1_gpu_suite.log 2>&1; tail -8 $OUT/1_gpu_suite.log | cut -c1-300
timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 > $OUT/2_bench.json 2> $OUT/2_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c9/2_bench.json').read().strip().split('\n')[-1])
print(d['value'], 'fps', d['ms_per_step'], 'ms; conv', d['roofline']['ms_per_frame'], 'frac', d['roofline']['frac'])
print(d['roofline_affinity']['ms_per_frame'], d['roofline_affinity']['matmul'])
PY
