# ATTN_P2Q with read_from_query's output projection inside (plans.P2Q_OUT): kernel test, the frame's parity tests, A/B bench lines in one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c28
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "p2q or query_chain" > $O/kernels.log 2>&1; tail -5 $O/kernels.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher.py -x -q -m gpu > $O/parity.log 2>&1; tail -5 $O/parity.log
for v in "CUTIE_AMD_P2Q_OUT=1" "CUTIE_AMD_P2Q_OUT=0" "CUTIE_AMD_P2Q_OUT=1" "CUTIE_AMD_P2Q_OUT=0"; do
  env $v timeout 300 python bench.py --full-bank-preroll 0 --cpu-frames 0 --no-breakdown --clips-in-flight 0 > $O/line.json 2> $O/line.err
  python - <<PY
import json
d = json.loads(open('$O/line.json').read().strip().split('\n')[-1])
print('[$v]', d['value'], d['value_no_lookahead'], d.get('repeats'))
PY
done
