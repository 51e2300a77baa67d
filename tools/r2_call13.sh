OUT=gpurun_out/c13; mkdir -p $OUT
cp gpurun_out/c12/conv_sweep_tiles.json cutie_amd/tiles_gfx950.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=16
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0"
rm -rf /tmp/prof; mkdir -p /tmp/prof
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats -- $BENCH > /tmp/prof/stats.log 2>&1
python tools/profile_summarize.py /tmp/prof r02b > $OUT/summary.log 2>&1
cp profiles/r02b_* $OUT/
KT=$(find /tmp/prof/stats -name '*kernel_trace.csv' | head -1)
python - "$KT" $OUT/trace_tail.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * 0.85):]
t0 = int(rows[0]['Start_Timestamp'])
key = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
with open(sys.argv[2], 'w') as f:
    w = csv.writer(f)
    w.writerow(['stream', 'start_us', 'dur_us', 'name', 'grid', 'wg', 'lds'])
    for r in rows:
        w.writerow([r[key], round((int(r['Start_Timestamp']) - t0) / 1e3, 2), round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 2),
                    r['Kernel_Name'][:70], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), r.get('LDS_Block_Size', '')])
PY
tail -3 /tmp/prof/stats.log
