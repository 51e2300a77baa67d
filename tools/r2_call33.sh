OUT=gpurun_out/c33; mkdir -p $OUT
for v in 0 0; do
CUTIE_AMD_NOPROJ=$v timeout 300 python bench.py --steps 300 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-roofline > $OUT/3_bench_$v.json 2> $OUT/3_bench_$v.err
python -c "
import json; d=json.loads(open('$OUT/3_bench_$v.json').read().strip().split('\n')[-1]); print('noproj=$v:', d['value'], 'fps', d['ms_per_step'], 'no-lookahead', d.get('value_no_lookahead'))"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0; do
rm -rf /tmp/prof$v; CUTIE_AMD_NOPROJ=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$v -- python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0 > /tmp/prof$v.log 2>&1
python - /tmp/prof$v $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith(('attn_', 'linear_mfma', 'void linear_mfma')):
        print('noproj=' + sys.argv[2], r['Name'][:40], r['Calls'], round(float(r['AverageNs']) / 1e3, 2), 'us')
PY
done
