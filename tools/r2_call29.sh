OUT=gpurun_out/c29; mkdir -p $OUT
for v in none -1 none -1; do
if [ $v = none ]; then unset BENCH_MAIN_PRIORITY; else export BENCH_MAIN_PRIORITY=$v; fi
timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-roofline > $OUT/3_bench_$v.json 2> $OUT/3_bench_$v.err
python -c "
import json; d=json.loads(open('$OUT/3_bench_$v.json').read().strip().split('\n')[-1]); print('main priority $v:', d['value'], 'fps', d['ms_per_step'])"
done
