OUT=gpurun_out/c28; mkdir -p $OUT
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print(p,'->',s.priority)
    except Exception as e: print(p,'ERR',e)
"
for v in none 1 2 -1 none 1; do
if [ $v = none ]; then unset CUTIE_AMD_SIDE_PRIORITY; else export CUTIE_AMD_SIDE_PRIORITY=$v; fi
timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 > $OUT/3_bench_$v.json 2> $OUT/3_bench_$v.err
python -c "
import json; d=json.loads(open('$OUT/3_bench_$v.json').read().strip().split('\n')[-1]); print('side priority $v:', d['value'], 'fps', d['ms_per_step'], 'no-lookahead', d.get('value_no_lookahead'))"
done
