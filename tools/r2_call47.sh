OUT=gpurun_out/c47; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "q2p or attention_with_fused" > $OUT/1_tests.log 2>&1; tail -2 $OUT/1_tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof0; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof0 -- python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0 > /tmp/prof0.log 2>&1
python - /tmp/prof0 <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith(('attn_', 'query_init2')): print(r['Name'][:40], r['Calls'], round(float(r['AverageNs']) / 1e3, 2), 'us')
PY
for v in a b; do
timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-roofline > $OUT/3_bench_$v.json 2> $OUT/3_bench_$v.err
python -c "
import json; d=json.loads(open('$OUT/3_bench_$v.json').read().strip().split('\n')[-1]); print('$v:', d['value'], 'fps', d['ms_per_step'], 'no-lookahead', d.get('value_no_lookahead'))"
done
