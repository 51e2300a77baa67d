# round 3, GPU call 35: all per-frame plan outputs from the frame-slot pool -> graph replay; parity + bench (longer legs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c35
mkdir -p $O
CUTIE_AMD_ARENA_POISON=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/1_parity.log 2>&1; tail -4 $O/1_parity.log
for g in 1 0 1; do
  CUTIE_AMD_GRAPHS=$g timeout 400 python bench.py --steps 500 --cpu-frames 0 --no-roofline --no-breakdown --clips-in-flight 0 > $O/bench_g$g.json 2> $O/bench_g$g.err
  python - <<PY
import json
d=json.loads(open('$O/bench_g$g.json').read().strip().split('\n')[-1])
print('GRAPHS=$g', d['value'], d.get('value_no_lookahead'), d.get('plans_eager_vs_graph_replay'), (d.get('full_bank') or {}).get('value'))
PY
done
