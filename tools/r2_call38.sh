OUT=gpurun_out/c38; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -x -k "strip or conv_dma_tiles" > $OUT/1_tests.log 2>&1; tail -3 $OUT/1_tests.log
timeout 1200 python tools/conv_sweep.py --objects 3 --out $OUT/conv_sweep > $OUT/2_sweep.log 2>&1; tail -1 $OUT/2_sweep.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/c38/conv_sweep.json'))
for r in sorted(d, key=lambda r:-r['count']*r['best'][2])[:26]:
    k=r['key']
    a=sorted(r['all'].items(), key=lambda kv:kv[1])[:4]
    st={t:round(v,1) for t,v in r['all'].items() if t[:2] in ('88','89','91','95')}
    print(k[:6], r['count'], ' '.join(f"{t}:{v:.1f}" for t,v in a), '| new', st)
PY
