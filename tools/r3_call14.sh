cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c14
mkdir -p $O
for i in 1 2; do
CUTIE_AMD_ARENA=0 timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_noarena$i.json 2> $O/5_bench_noarena$i.err
timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_arena$i.json 2> $O/5_bench_arena$i.err
done
for f in $O/5_bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d.get('value_no_lookahead'), d['roofline']['frac'], d['device_us_by_kind'].get('CONV'), d.get('multi_clip',{}).get('value'))
"; done
CUTIE_AMD_ARENA_POISON=1 timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_kernels.py > $O/1_gpu_poison.log 2>&1
tail -n 5 $O/1_gpu_poison.log
