# Round 4, call 29: the serialized-load fixes found by tools/isa_waits.py (aff_score c_j / tau_j, conv_pc bias, UP4_SOFTMAX per object count,
# AREA_DOWN3 per ratio, mask-encoder STEM, KEY_PREP c_j, SUMMARIZE final sum) -- all bit-identical by construction.
#   1. the whole GPU suite on the new library (what the driver runs at round end)
#   2. A/B inside this box: tools/abl/libcutie_hip_OLD.so (= the library of the previous commit, built by tools/build_old_lib.sh) against
#      the new one under the same Python, interleaved, with the per-kind device times of the breakdown
#   3. affinity stage times (tools/aff_ab.py), old and new
#   4. the driver's own command on the new library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c29
mkdir -p $O
OLD=$GRAFT_REPO_ROOT/tools/abl/libcutie_hip_OLD.so
timeout 480 python -m pytest tests/ -x -q -m gpu > $O/suite.log 2>&1; tail -3 $O/suite.log
for v in NEW OLD NEW OLD; do
  if [ $v = OLD ]; then export CUTIE_AMD_LIB=$OLD; else unset CUTIE_AMD_LIB; fi
  timeout 200 python bench.py --full-bank-preroll 0 --cpu-frames 0 --clips-in-flight 0 > $O/line_$v.json 2> $O/line_$v.err
  python - <<PY
import json
d = json.loads(open('$O/line_$v.json').read().strip().split('\n')[-1])
k = d['device_us_by_kind']
g = lambda n: [v for kk, v in k.items() if kk.startswith(n)][0][1]
print('[$v]', d['value'], d['value_no_lookahead'], d['repeats']['median'], 'conv', d['roofline']['ms_per_frame'], d['roofline']['frac'],
      'aff', d['roofline_affinity']['ms_per_frame'], d['roofline_affinity']['matmul']['mfma_util'], d['roofline_affinity']['matmul']['stage_us'],
      'up4', g('UP4'), 'area3', g('AREA_DOWN3'), 'keyprep', g('KEY_PREP'), 'stem', g('STEM'), 'summ', g('SUMMARIZE'))
PY
  cat $O/line_$v.json >> $O/lines_$v.jsonl
done
unset CUTIE_AMD_LIB
timeout 120 python tools/aff_ab.py 300 2>&1 | grep -E "^tokens|^nq 2" | sed 's/^/[NEW] /'
CUTIE_AMD_LIB=$OLD timeout 120 python tools/aff_ab.py 300 2>&1 | grep -E "^tokens|^nq 2" | sed 's/^/[OLD] /'
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver_line.err
t1=$(date +%s.%N)
python - <<PY
import json
d = json.loads(open('$O/driver_line.json').read().strip().split('\n')[-1])
print('driver cmd: wall %.1f s' % ($t1 - $t0), d['value'], d['value_no_lookahead'], d['repeats']['values'], 'full', d['full_bank']['value'], 'conv', d['roofline']['ms_per_frame'], d['roofline']['frac'],
      'aff', d['roofline_affinity']['ms_per_frame'], d['roofline_affinity']['matmul']['mfma_util'], 'cpu', d['cpu_baseline']['value'], 'multi', d.get('multi_clip', {}).get('value'))
PY
