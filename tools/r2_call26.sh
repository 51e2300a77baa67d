OUT=gpurun_out/c26; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=16
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0"
rm -rf /tmp/prof; mkdir -p /tmp/prof
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats -- $BENCH > /tmp/prof/stats.log 2>&1
KT=$(find /tmp/prof/stats -name '*kernel_trace.csv' | head -1)
python - "$KT" $OUT <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
key = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
n = len(rows)
for name, lo, hi in (('lookahead', 0.30, 0.40), ('nolookahead', 0.88, 0.98)):
    seg = rows[int(n * lo):int(n * hi)]
    t0 = int(seg[0]['Start_Timestamp'])
    with open(f'{sys.argv[2]}/trace_{name}.csv', 'w') as f:
        w = csv.writer(f)
        w.writerow(['stream', 'start_us', 'dur_us', 'name', 'grid', 'wg'])
        for r in seg:
            w.writerow([r[key], round((int(r['Start_Timestamp']) - t0) / 1e3, 2), round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 2),
                        r['Kernel_Name'][:70], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))])
PY
tail -2 /tmp/prof/stats.log
