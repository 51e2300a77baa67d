# round 3, GPU call 6: conv_pc tests after the deep-ring fix, sweeps at K = 1, 2, 3 (480p) and K = 5 (1080p), bench with the merged table
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --maxfail=40 -k "conv" > $O/1_tests.log 2>&1
timeout 1500 python tools/conv_sweep.py --objects 1 2 3 --families dma,pc,halo --out $O/sweep480 > $O/2_sweep480.log 2>&1
timeout 1500 python tools/conv_sweep.py --objects 5 --height 1080 --width 1920 --families dma,pc,halo --out $O/sweep1080 > $O/3_sweep1080.log 2>&1
python tools/merge_tile_tables.py $O/tiles_merged.json cutie_amd/tiles_gfx950.json $O/sweep1080_tiles.json $O/sweep480_tiles.json > $O/4_merge.log 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_old.json 2> $O/5_bench_old.err
CUTIE_AMD_TILE_CACHE=$O/tiles_merged.json timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_new.json 2> $O/5_bench_new.err
timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_old2.json 2> $O/5_bench_old2.err
CUTIE_AMD_TILE_CACHE=$O/tiles_merged.json timeout 300 python bench.py --steps 100 --warmup 10 > $O/5_bench_new2.json 2> $O/5_bench_new2.err
tail -n 4 $O/1_tests.log
tail -n 3 $O/2_sweep480.log $O/3_sweep1080.log
cat $O/4_merge.log
for f in $O/5_bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d.get('value_no_lookahead'), d['roofline']['frac'], d['device_us_by_kind'].get('CONV'))
"; done
