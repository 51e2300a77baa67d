# Round 5, call 6: stacked read-outs with kernel forms by frame count; version counter in the frame key
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c6
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lookahead or overwritten" > $O/la_tests.log 2>&1; tail -3 $O/la_tests.log
bash tools/ab.sh affbatch2 2 "CUTIE_AMD_AFF_BATCH=1" "CUTIE_AMD_AFF_BATCH=8" "CUTIE_AMD_AFF_BATCH=8 CUTIE_AMD_AFF_BATCH_FORMS=0" 2>&1 | tee $O/ab.log
timeout 200 python bench.py --full-bank-preroll 0 --cpu-frames 0 --clips-in-flight 0 --steps 20 --warmup 5 > $O/line.json 2> $O/line.err; tail -3 $O/line.err
python - <<PY
import json
d = json.loads(open('$O/line.json').read().strip().split('\n')[-1])
print(d['value'], d['value_no_lookahead'], d['repeats'])
a = d['roofline_affinity']; m = a['matmul']
print('aff ms/frame', a['ms_per_frame'], 'launches/frame', a['launches_per_frame'], 'tokens', a['memory_tokens'])
print({k: m[k] for k in ('launches', 'frames_read', 'frames_per_launch', 'us_per_frame', 'mfma_util', 'stage_plan', 'stage_us_per_frame')})
print('graph', d['frame_as_one_hip_graph_ms'], d['frame_as_one_hip_graph_launches'])
PY
