# round 3, GPU call 38: STEM kernel -- kernel test, parity, bench A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c38
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "stem or img_prep" > $O/1_kernels.log 2>&1; tail -8 $O/1_kernels.log
CUTIE_AMD_ARENA_POISON=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/2_parity.log 2>&1; tail -3 $O/2_parity.log
for st in 1 0 1 0; do
  CUTIE_AMD_STEM=$st timeout 400 python bench.py --steps 400 --cpu-frames 0 --no-roofline --no-breakdown --clips-in-flight 0 --full-bank-preroll 0 > $O/bench_s$st.json 2> $O/bench_s$st.err
  python - <<PY
import json
d=json.loads(open('$O/bench_s$st.json').read().strip().split('\n')[-1])
print('STEM=$st', d['value'], d.get('value_no_lookahead'))
PY
done
