# Round 5, final tree: the three bench lines (C2, C1, C4) in one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5lines
mkdir -p $O
timeout 600 python bench.py > $O/c2.json 2> $O/c2.err; tail -c 200 $O/c2.json
timeout 600 python bench.py --objects 1 --no-long-term --cpu-frames 0 --clips-in-flight 0 > $O/c1.json 2> $O/c1.err; tail -c 100 $O/c1.json
timeout 900 python bench.py --height 1080 --width 1920 --objects 5 --no-long-term --cpu-frames 0 --clips-in-flight 0 --preroll 100 > $O/c4.json 2> $O/c4.err; tail -c 100 $O/c4.json
python - <<PY
import json
for n in ('c2','c1','c4'):
    try:
        d=json.loads(open('$O/%s.json'%n).read().strip().split('\n')[-1])
        m=d['roofline_affinity']['matmul']
        print(n, d['value'], d.get('value_no_lookahead'), d['repeats']['values'], d['repeats']['mean_fps_all_regions'], 'conv', d['roofline']['ms_per_frame'], d['roofline']['frac'], d['roofline']['executed_frac'], d['roofline']['hbm_frac'],
              'aff', d['roofline_affinity']['ms_per_frame'], m['mfma_util'], m['frames_per_launch'], m['stage_us_per_frame'], d.get('multi_clip',{}).get('value'), d['config']['memory_tokens_end'], (d.get('full_bank') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(n,'FAILED',e, open('$O/%s.err'%n).read()[-500:])
PY
