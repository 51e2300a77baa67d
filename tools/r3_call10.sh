cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c10
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --maxfail=40 -k "conv_pc or gap_accum" > $O/1_tests.log 2>&1
timeout 120 python tools/multi_conv.py 3,30,54,256,256,3,110,0,0,0 3,30,54,256,256,3,110,0,0,1 3,30,54,256,256,3,66,0,0,0 3,30,54,256,256,3,66,0,0,1 3,30,54,256,256,3,100,0,0,1 3,30,54,256,256,3,129,0,0,1 3,30,54,256,256,3,129,0,0,0 2>&1 | tee $O/gap.log
T=$PWD/gpurun_out_r3c6_tiles_merged.json
for i in 1 2; do
timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_old$i.json 2> $O/5_bench_old$i.err
CUTIE_AMD_TILE_CACHE=$T timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_new$i.json 2> $O/5_bench_new$i.err
done
tail -n 4 $O/1_tests.log
for f in $O/5_bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d.get('value_no_lookahead'), d['roofline']['frac'], d['device_us_by_kind'].get('CONV'))
"; done
