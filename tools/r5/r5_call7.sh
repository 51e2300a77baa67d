# Round 5, call 7: consolidation kernels (rank select by split counting, similarities on fp32 MFMA, prototypes on bf16 MFMA)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c7
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "rank_select or consolidation or affinity or bank" > $O/k_tests.log 2>&1; tail -3 $O/k_tests.log
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small_lt or bank_contents or lookahead or overwritten or small_flip or small_cfg_lt" > $O/p_tests.log 2>&1; tail -3 $O/p_tests.log
timeout 400 python -m pytest tests/test_gpu_teacher.py -x -q -m gpu -k "480 or lt" > $O/t_tests.log 2>&1; tail -3 $O/t_tests.log
timeout 100 python tools/aff_batch_ab.py 12200 2>&1 | tee $O/aff_batch_ab.txt
timeout 200 python bench.py --full-bank-preroll 0 --cpu-frames 0 --clips-in-flight 0 > $O/line.json 2> $O/line.err; tail -3 $O/line.err
python - <<PY
import json
d = json.loads(open('$O/line.json').read().strip().split('\n')[-1])
print(d['value'], d['value_no_lookahead'], d['repeats']['values'], d['repeats']['mean_fps_all_regions'])
a = d['roofline_affinity']; m = a['matmul']
print('aff ms/frame', a['ms_per_frame'], 'launches/frame', a['launches_per_frame'], 'tokens', a['memory_tokens'])
print({k: m[k] for k in ('launches', 'frames_read', 'frames_per_launch', 'us_per_frame', 'mfma_util', 'stage_plan', 'stage_us_per_frame')})
k = d['device_us_by_kind']
print({n: v for n, v in k.items() if any(t in n for t in ('RANK', 'CONSOL', 'GATHER', 'COPY2D', 'BANK_WRITE', 'KEY_PREP', 'MEMSET'))})
PY
