# round 3, GPU call 39: ATTN_Q2P with 8 waves / logits templates / head stride -- tests, timeline, bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c39
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "chain or q2p" > $O/1_kernels.log 2>&1; tail -6 $O/1_kernels.log
CUTIE_AMD_LIB=tools/abl/libcutie_hip_ATL.so timeout 300 python tools/attn_timeline.py > $O/timeline.log 2>&1
grep -E "launch alone" $O/timeline.log
grep -A14 "ATTN_Q2P with q handed in:" $O/timeline.log | grep -E "stamp|cold:|warm:" | cut -c1-120
timeout 400 python bench.py --steps 400 --cpu-frames 0 --no-roofline --no-breakdown --clips-in-flight 0 --full-bank-preroll 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1])
print(d['value'], d.get('value_no_lookahead'))
PY
