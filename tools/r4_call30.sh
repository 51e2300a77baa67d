# Round 4, call 30 (the last one of the round): final tree.
#   1. tools/kp_ab.py: KEY_PREP c_j forms, GRU forms, isolated (decides two Python-side defaults; both forms are in the suite)
#   2. the whole GPU suite
#   3. A/B inside this box against tools/abl/libcutie_hip_OLD.so (= the kernel library of 4fb4f97, the tree before this session's changes)
#   4. the driver's own command
#   5. rocprofv3 --kernel-trace --stats of the profile command (kernel stats only; the PMC passes of r04_summary.json stay those of tree db4ee82)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c30
mkdir -p $O
OLD=$GRAFT_REPO_ROOT/tools/abl/libcutie_hip_OLD.so
timeout 120 python tools/kp_ab.py > $O/kp_ab.log 2>&1; cat $O/kp_ab.log | tail -16
timeout 420 python -m pytest tests/ -q -m gpu --maxfail=8 > $O/suite.log 2>&1; tail -3 $O/suite.log; grep -E "^(FAILED|ERROR)" $O/suite.log | head -10
for v in NEW OLD NEW OLD; do
  if [ $v = OLD ]; then export CUTIE_AMD_LIB=$OLD; else unset CUTIE_AMD_LIB; fi
  timeout 200 python bench.py --full-bank-preroll 0 --cpu-frames 0 --clips-in-flight 0 > $O/line_$v.json 2> $O/line_$v.err
  python - <<PY
import json
d = json.loads(open('$O/line_$v.json').read().strip().split('\n')[-1])
k = d['device_us_by_kind']
g = lambda n: [v for kk, v in k.items() if kk.startswith(n)][0][1]
print('[$v]', d['value'], d['value_no_lookahead'], d['repeats']['median'], 'conv', d['roofline']['ms_per_frame'], d['roofline']['frac'],
      'aff', d['roofline_affinity']['ms_per_frame'], d['roofline_affinity']['matmul']['mfma_util'],
      'up4', g('UP4'), 'area3', g('AREA_DOWN3'), 'keyprep', g('KEY_PREP'), 'stem', g('STEM'), 'gru', g('GRU'), 'copy2d', g('COPY2D'))
PY
  cat $O/line_$v.json >> $O/lines_$v.jsonl
done
unset CUTIE_AMD_LIB
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver_line.err
t1=$(date +%s.%N)
python - <<PY
import json
d = json.loads(open('$O/driver_line.json').read().strip().split('\n')[-1])
print('driver cmd: wall %.1f s' % ($t1 - $t0), d['value'], d['value_no_lookahead'], d['repeats']['values'], 'full', d['full_bank']['value'], 'conv', d['roofline']['ms_per_frame'], d['roofline']['frac'],
      'aff', d['roofline_affinity']['ms_per_frame'], d['roofline_affinity']['matmul']['mfma_util'], 'cpu', d['cpu_baseline']['value'], 'multi', d.get('multi_clip', {}).get('value'))
PY
rm -rf /tmp/prof_r04f
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04f/stats -- python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 > $O/stats.log 2>&1
cp $(find /tmp/prof_r04f/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null && head -12 $O/kernel_stats.csv | cut -c1-150
