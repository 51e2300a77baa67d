# The driver's own command (all legs), timed.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5drv
mkdir -p $O
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver_line.err
t1=$(date +%s.%N)
tail -3 $O/driver_line.err
python - <<PY
import json
d = json.loads(open('$O/driver_line.json').read().strip().split('\n')[-1])
print('wall %.1f s' % ($t1 - $t0), 'value', d['value'], 'no_la', d['value_no_lookahead'], d['repeats']['values'], 'mean', d['repeats']['mean_fps_all_regions'])
print('full', d['full_bank'], '\nmulti', d.get('multi_clip'), '\ncpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
r = d['roofline']; print({k: r[k] for k in ('frac', 'executed_frac', 'hbm_frac', 'ms_per_frame', 'launches_per_frame', 'traffic')})
a = d['roofline_affinity']; m = a['matmul']
print('aff ms/frame', a['ms_per_frame'], 'launches/frame', a['launches_per_frame'], 'tokens', a['memory_tokens'], 'frac', a['frac'])
print({k: m[k] for k in ('launches', 'frames_read', 'frames_per_launch', 'us_per_frame', 'mfma_util', 'stage_plan', 'stage_us_per_frame')})
print('graph', d['frame_as_one_hip_graph_ms'], d['frame_as_one_hip_graph_launches'])
k = d['device_us_by_kind']; print({n: v for n, v in k.items() if any(t in n for t in ('RANK', 'CONSOL', 'SUMM', 'STEM', 'GRU', 'UPSAMPLE', 'ECA', 'AREA', 'KEY_PREP', 'COPY', 'BANK'))})
PY
