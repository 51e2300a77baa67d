# round 3, GPU call 32: plans replayed as HIP graphs -- probe, parity, bench A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c32
mkdir -p $O
timeout 400 python tools/defer_probe.py > $O/probe.log 2>&1; tail -7 $O/probe.log
CUTIE_AMD_ARENA_POISON=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/1_parity.log 2>&1; tail -4 $O/1_parity.log
for g in 1 0; do
  CUTIE_AMD_GRAPHS=$g timeout 400 python bench.py --cpu-frames 0 --no-roofline --no-breakdown > $O/bench_g$g.json 2> $O/bench_g$g.err
  python - <<PY
import json
d=json.loads(open('$O/bench_g$g.json').read().strip().split('\n')[-1])
print('GRAPHS=$g', d['value'], d.get('value_no_lookahead'), (d.get('multi_clip') or {}).get('value'), d.get('plans_eager_vs_graph_replay'), (d.get('full_bank') or {}).get('value'))
PY
done
