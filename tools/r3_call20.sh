# round 3, GPU call 20: query chain v2 (fixed-point accumulators, MFMA attention cores) -- kernel tests, timeline, bench A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c20
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "chain or attn or q2p or fused_proj or query_init" > $O/1_kernels.log 2>&1; tail -15 $O/1_kernels.log
CUTIE_AMD_LIB=tools/abl/libcutie_hip_ATL.so timeout 300 python tools/attn_timeline.py > $O/timeline.log 2>&1
grep -E "launch alone|stamp  [0-9]:|stamp 1[0-9]:|cold:|warm:" $O/timeline.log | cut -c1-150
for q in 1; do
  CUTIE_AMD_QCHAIN=$q timeout 400 python bench.py --cpu-frames 0 --no-roofline --no-breakdown > $O/bench_q$q.json 2> $O/bench_q$q.err
  python - <<PY
import json
d=json.loads(open('$O/bench_q$q.json').read().strip().split('\n')[-1])
print('QCHAIN=$q', d['value'], d.get('value_no_lookahead'), (d.get('multi_clip') or {}).get('value'))
PY
done
