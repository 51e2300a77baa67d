# Round 5, call 2: one affinity read-out per bank version (AFF_BATCH) + the store-lane fix of aff_score.
#   1. kernel tests of the affinity family (batched frames against one-frame plans), look-ahead parity tests
#   2. A/B inside this box: CUTIE_AMD_AFF_BATCH=1 (frame by frame, as round 4) against 8
#   3. the bench line with the affinity breakdown
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c2
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "affinity" > $O/aff_tests.log 2>&1; tail -3 $O/aff_tests.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lookahead" > $O/la_tests.log 2>&1; tail -3 $O/la_tests.log
bash tools/ab.sh affbatch 2 "CUTIE_AMD_AFF_BATCH=1" "CUTIE_AMD_AFF_BATCH=8" 2>&1 | tee $O/ab.log
timeout 200 python bench.py --full-bank-preroll 0 --cpu-frames 0 --clips-in-flight 0 > $O/line.json 2> $O/line.err; tail -3 $O/line.err
python - <<PY
import json
d = json.loads(open('$O/line.json').read().strip().split('\n')[-1])
print(d['value'], d['value_no_lookahead'], d['repeats'])
print(json.dumps(d['roofline_affinity'], indent=1))
print({k: d['roofline'][k] for k in ('frac', 'executed_frac', 'hbm_frac', 'ms_per_frame', 'algorithmic_gflop_per_frame', 'gflop_per_frame')})
print('graph', d['frame_as_one_hip_graph_ms'], d['frame_as_one_hip_graph_launches'])
PY
