cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c40
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "cout1 or conv_auto or stem" > $O/1_kernels.log 2>&1; tail -3 $O/1_kernels.log
timeout 400 python bench.py --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1])
print(d['value'], d.get('value_no_lookahead'), d['device_us_by_kind'].get('STEM'), d['device_us_by_kind'].get('QUERY_INIT'), d['roofline']['ms_per_frame'], d['roofline']['frac'])
PY
