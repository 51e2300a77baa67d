# round 3, GPU call 21: chain v2 parity (teacher, scenarios, bike, kernels with poisoned arenas) + in-frame kernel durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c21
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_teacher.py -q -m gpu -x > $O/1_teacher.log 2>&1; tail -3 $O/1_teacher.log
CUTIE_AMD_ARENA_POISON=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/2_parity.log 2>&1; tail -4 $O/2_parity.log
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --no-lookahead --no-breakdown"
rm -rf /tmp/prof_q
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -- $BENCH > $O/stats.log 2>&1
f=$(find /tmp/prof_q -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3c21/kernel_stats.csv')))
nf=None
for r in rows:
    if 'query_init2' in r['Name']: nf=int(r['Calls'])
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('frames', nf, 'total us/frame', tot/nf/1e3)
for r in rows[:45]:
    print('%-60s calls/frame %5.1f avg %7.1f us  per frame %7.1f us' % (r['Name'][:60], int(r['Calls'])/nf, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/nf/1e3))
PY
