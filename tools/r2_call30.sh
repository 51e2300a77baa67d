OUT=gpurun_out/c30; mkdir -p $OUT
for v in none 128 192 128/2 64/4 none 160; do
if [ $v = none ]; then unset CUTIE_AMD_SIDE_CUS; else export CUTIE_AMD_SIDE_CUS=$v; fi
n=$(echo $v | tr '/' '_')
timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-roofline > $OUT/3_bench_$n.json 2> $OUT/3_bench_$n.err
python -c "
import json; d=json.loads(open('$OUT/3_bench_$n.json').read().strip().split('\n')[-1]); print('side CUs $v:', d['value'], 'fps', d['ms_per_step'])" || tail -3 $OUT/3_bench_$n.err
done
