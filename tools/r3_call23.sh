# round 3, GPU call 23: ATTN_Q2P split form -- kernel tests, timeline, bench A/B (split 8 / 1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c23
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "chain or attn or q2p or fused_proj or query_init" > $O/1_kernels.log 2>&1; tail -15 $O/1_kernels.log
CUTIE_AMD_LIB=tools/abl/libcutie_hip_ATL.so timeout 300 python tools/attn_timeline.py > $O/timeline.log 2>&1
grep -A18 "ATTN_Q2P" $O/timeline.log | grep -E "launch alone|stamp|cold:|warm:" | cut -c1-150
for sp in 8 4 1; do
  CUTIE_AMD_Q2P_SPLIT=$sp timeout 400 python bench.py --cpu-frames 0 --no-roofline --no-breakdown --clips-in-flight 0 > $O/bench_s$sp.json 2> $O/bench_s$sp.err
  python - <<PY
import json
d=json.loads(open('$O/bench_s$sp.json').read().strip().split('\n')[-1])
print('SPLIT=$sp', d['value'], d.get('value_no_lookahead'))
PY
done
