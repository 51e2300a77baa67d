OUT=gpurun_out/c36; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "gap_acc or strip" > $OUT/1_tests.log 2>&1; tail -3 $OUT/1_tests.log
timeout 1200 python tools/conv_sweep.py --objects 3 --out $OUT/conv_sweep > $OUT/2_sweep.log 2>&1; tail -1 $OUT/2_sweep.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/c36/conv_sweep.json'))
for r in sorted(d, key=lambda r:-r['count']*r['best'][2]):
    k=r['key']
    if k[3]==3 and k[4]==1 and any(t.startswith('9') for t in r['all']):
        a=sorted(r['all'].items(), key=lambda kv:kv[1])[:5]
        st={t:round(v,1) for t,v in r['all'].items() if t[:2] in ('90','91','92','93','94')}
        print(k[:6], r['count'], ' '.join(f"{t}:{v:.1f}" for t,v in a), '| strip', st)
PY
