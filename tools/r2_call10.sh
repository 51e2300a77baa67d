OUT=gpurun_out/c10; mkdir -p $OUT
for rep in 1 2; do
for v in 1 0; do
CUTIE_AMD_AHEAD_AFFINITY=$v timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --no-roofline > $OUT/bench_ahead$v.$rep.json 2> $OUT/err.log
python -c "
import json; d=json.loads(open('$OUT/bench_ahead$v.$rep.json').read().strip().split('\n')[-1]); print('ahead affinity $v rep $rep:', d['value'], 'fps', d['ms_per_step'])"
done; done
timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --no-roofline --no-lookahead > $OUT/bench_nola.json 2>> $OUT/err.log
python -c "
import json; d=json.loads(open('$OUT/bench_nola.json').read().strip().split('\n')[-1]); print('no lookahead:', d['value'], 'fps', d['ms_per_step'])"
