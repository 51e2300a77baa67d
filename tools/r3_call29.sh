# round 3, GPU call 29: deferred memorising (no look-ahead hint) -- parity + bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c29
mkdir -p $O
CUTIE_AMD_ARENA_POISON=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/1_parity.log 2>&1; tail -5 $O/1_parity.log
for d in 1 0; do
  CUTIE_AMD_DEFER_MEM=$d timeout 400 python bench.py --cpu-frames 0 --no-roofline --no-breakdown --clips-in-flight 0 > $O/bench_d$d.json 2> $O/bench_d$d.err
  python - <<PY
import json
d=json.loads(open('$O/bench_d$d.json').read().strip().split('\n')[-1])
print('DEFER=$d', d['value'], d.get('value_no_lookahead'))
PY
done
