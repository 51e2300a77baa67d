# Round 4, call 2: cold tile sweep of the batched encoder geometries (window 4 and 8), and the full bench line (multi-clip leg included)
# with window 1 / 8 in this box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c2
mkdir -p $O
timeout 600 python tools/conv_sweep.py --window 4 8 --cold 160 --reps 3 --iters 8 --out $O/sweep_window > $O/1_sweep.log 2>&1; tail -70 $O/1_sweep.log
for w in 1 8; do
  CUTIE_AMD_WINDOW=$w timeout 400 python bench.py --cpu-frames 0 --full-bank-preroll 0 > $O/bench_w$w.json 2> $O/bench_w$w.err
  python - <<PY
import json
d = json.loads(open('$O/bench_w$w.json').read().strip().split('\n')[-1])
print('window $w:', d['value'], d['repeats'], d.get('value_no_lookahead'), d['roofline']['ms_per_frame'], d['roofline']['frac'], d['multi_clip'])
print(d['device_us_by_kind'])
PY
done
