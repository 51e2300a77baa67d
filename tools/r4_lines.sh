# Round 4 bench lines: the headline workload (C2), BASELINE configs[1] (C1: 480p, 1 object, no long-term memory) and configs[4] (C4: 1080p, 5 objects, no long-term)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4lines
mkdir -p $O
timeout 600 python bench.py > $O/c2.json 2> $O/c2.err; tail -c 400 $O/c2.json
timeout 600 python bench.py --objects 1 --no-long-term --cpu-frames 0 --clips-in-flight 0 > $O/c1.json 2> $O/c1.err; tail -c 300 $O/c1.json
timeout 900 python bench.py --height 1080 --width 1920 --objects 5 --no-long-term --cpu-frames 0 --clips-in-flight 0 --preroll 100 > $O/c4.json 2> $O/c4.err; tail -c 300 $O/c4.json
python - <<PY
import json
for n in ('c2','c1','c4'):
    try:
        d=json.loads(open('$O/%s.json'%n).read().strip().split('\n')[-1])
        print(n, d['value'], d.get('value_no_lookahead'), d['repeats']['values'], 'conv', d['roofline']['ms_per_frame'], d['roofline']['frac'], 'aff', d['roofline_affinity']['ms_per_frame'], d['roofline_affinity']['matmul']['mfma_util'], d['roofline_affinity']['matmul']['stage_us'], d.get('multi_clip',{}).get('value'), d['config']['memory_tokens_end'])
    except Exception as e:
        print(n,'FAILED',e, open('$O/%s.err'%n).read()[-500:])
PY
