# round 3, GPU call 18: query chain in four launches -- kernel tests, parity (bike, teacher), bench A/B, in-frame kernel durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c18
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "chain or attn or q2p or fused_proj or query_init or linear" > $O/1_kernels.log 2>&1; tail -5 $O/1_kernels.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bike_argmax or scenario or lookahead" -s > $O/2_parity.log 2>&1; grep -E "bike frame|passed|failed|Error" $O/2_parity.log | tail -8
timeout 600 python -m pytest tests/test_gpu_teacher.py -q -m gpu -x > $O/3_teacher.log 2>&1; tail -3 $O/3_teacher.log
for q in 0 1; do
  CUTIE_AMD_QCHAIN=$q timeout 400 python bench.py --cpu-frames 0 --no-roofline > $O/bench_q$q.json 2> $O/bench_q$q.err
  python - <<PY
import json
d=json.loads(open('$O/bench_q$q.json').read().strip().split('\n')[-1])
print('QCHAIN=$q', d['value'], d.get('value_no_lookahead'), d.get('multi_clip'), d.get('device_us_by_kind'))
PY
done
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --no-lookahead"
rm -rf /tmp/prof_q
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -- $BENCH > $O/stats.log 2>&1
f=$(find /tmp/prof_q -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
grep -E "attn|qffn|linear|query_init" $O/kernel_stats.csv | cut -c1-160
