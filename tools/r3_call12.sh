cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c12
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --maxfail=40 -k "gap_accum or (conv_pc and (110 or 100 or 129))" > $O/1_tests.log 2>&1
timeout 120 python tools/multi_conv.py 3,30,54,256,256,3,110,0,0,0 3,30,54,256,256,3,110,0,0,1 3,30,54,256,256,3,66,0,0,1 3,30,54,256,256,3,100,0,0,1 3,30,54,256,256,3,129,0,0,1 2>&1 | tee $O/gap.log
timeout 1500 python tools/conv_sweep.py --objects 3 --cold 160 --reps 3 --iters 8 --families dma,pc,halo --out $O/sweepcold > $O/2_sweepcold.log 2>&1
python tools/merge_tile_tables.py $O/tiles_cold.json cutie_amd/tiles_gfx950.json $O/sweepcold_tiles.json > $O/4_merge.log 2>&1
for i in 1 2; do
timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_old$i.json 2> $O/5_bench_old$i.err
CUTIE_AMD_TILE_CACHE=$O/tiles_cold.json timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_cold$i.json 2> $O/5_bench_cold$i.err
done
tail -n 3 $O/1_tests.log
tail -n 48 $O/2_sweepcold.log
for f in $O/5_bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d.get('value_no_lookahead'), d['roofline']['frac'], d['device_us_by_kind'].get('CONV'))
"; done
