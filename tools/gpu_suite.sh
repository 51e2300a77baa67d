# The whole GPU side of a round in ONE gpurun call: smoke(), the -m gpu suite, the driver's bench command (-> gpurun_out/<tag>_bench_line.json).
#   bash tools/gpu_suite.sh <tag> [extra pytest args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-suite}; shift
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests/ -x -q -m gpu "$@" 2>&1 | tail -12 | tee gpurun_out/${tag}_pytest_tail.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
python - <<PY
import json
d = json.loads(open('gpurun_out/${tag}_bench_line.json').read().strip().split('\n')[-1])
m = d.get('multi_clip') or {}
print('value', d['value'], 'first', d.get('value_first_region'), 'no_lookahead', d.get('value_no_lookahead'), 'multi_clip', m.get('value'), (m.get('roofline') or {}).get('frac'),
      'roofline', d['roofline']['frac'], 'aff', d['roofline_affinity']['matmul']['mfma_util'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
