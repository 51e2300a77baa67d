OUT=gpurun_out/c23; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gap or dma" > $OUT/1_kernel_tests.log 2>&1; tail -3 $OUT/1_kernel_tests.log
timeout 1200 python tools/conv_sweep.py --objects 3 1 2 --out $OUT/conv_sweep > $OUT/2_sweep.log 2>&1; tail -1 $OUT/2_sweep.log
for v in old new old new; do
cp cutie_amd/tiles_gfx950.json /tmp/tiles_orig.json
if [ $v = new ]; then python - <<'PY'
import json
a=json.load(open('cutie_amd/tiles_gfx950.json')); b=json.load(open('gpurun_out/c23/conv_sweep_tiles.json'))
nb={json.dumps(k):v for k,v in b['tiles']}
a['tiles']=[[k, nb.get(json.dumps(k), v)] for k,v in a['tiles']]
json.dump(a, open('cutie_amd/tiles_gfx950.json','w'))
PY
fi
timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 > $OUT/3_bench_$v.json 2> $OUT/3_bench_$v.err
cp /tmp/tiles_orig.json cutie_amd/tiles_gfx950.json
python -c "
import json; d=json.loads(open('$OUT/3_bench_$v.json').read().strip().split('\n')[-1]); print('$v table:', d['value'], 'fps', d['ms_per_step'], 'no-lookahead', d.get('value_no_lookahead'), 'conv ms', d['roofline']['ms_per_frame'], 'frac', d['roofline']['frac'])"
done
