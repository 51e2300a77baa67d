# round 3, GPU call 25: ATTN_Q2P with q handed in by the previous ATTN_P2Q -- kernel tests, timeline, bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c25
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "chain or attn or q2p or fused_proj or query_init" > $O/1_kernels.log 2>&1; tail -15 $O/1_kernels.log
CUTIE_AMD_LIB=tools/abl/libcutie_hip_ATL.so timeout 300 python tools/attn_timeline.py > $O/timeline.log 2>&1
grep -E "launch alone" $O/timeline.log
grep -A18 "ATTN_Q2P with q" $O/timeline.log | grep -E "stamp|cold:|warm:" | cut -c1-150
for qn in 1 0; do
  CUTIE_AMD_QNEXT=$qn timeout 400 python bench.py --cpu-frames 0 --no-roofline --no-breakdown --clips-in-flight 0 > $O/bench_n$qn.json 2> $O/bench_n$qn.err
  python - <<PY
import json
d=json.loads(open('$O/bench_n$qn.json').read().strip().split('\n')[-1])
print('QNEXT=$qn', d['value'], d.get('value_no_lookahead'))
PY
done
