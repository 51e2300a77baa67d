# Round 5, call 40: affinity / commit plans patched in place (host time of memory frames): the look-ahead / long-term / batch tests on the GPU
# and the bench line (four interleaved clips are bound by the host's time per frame)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c40
mkdir -p $O
timeout 110 python -m pytest tests -m gpu -x -q -k "lookahead or long_term or small_lt or interleaved or two_bucket or affinity_batched or overwritten" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 60 python bench.py --steps 100 --warmup 10 --cpu-frames 0 --no-roofline --full-bank-preroll 0 --no-graph > $O/line.json 2> $O/line.err
python - <<PY
import json
d=json.loads(open('$O/line.json').read().strip().splitlines()[-1])
print(d['value'], d['value_no_lookahead'], d['no_lookahead']['eval_vos_protocol']['fps'], 'multi', d['multi_clip'].get('value'), d['multi_clip'].get('error'))
PY
