# The bench lines of a round in ONE box: C2 (the driver's workload, 200 steps), C1 (1 object, no long-term), C4 (1080p, 5 objects) -> gpurun_out/<tag>_bench_line_c{2,1,4}.json
#   bash tools/bench_lines.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-lines}
mkdir -p gpurun_out
python bench.py --steps 200 --warmup 20 > gpurun_out/${tag}_bench_line_c2.json 2> gpurun_out/${tag}_c2.err
python bench.py --steps 200 --warmup 20 --objects 1 --no-long-term --cpu-frames 0 --full-bank-preroll 0 > gpurun_out/${tag}_bench_line_c1.json 2> gpurun_out/${tag}_c1.err
python bench.py --steps 100 --warmup 10 --preroll 60 --objects 5 --height 1080 --width 1920 --no-long-term --cpu-frames 0 --full-bank-preroll 0 --clips-in-flight 0 > gpurun_out/${tag}_bench_line_c4.json 2> gpurun_out/${tag}_c4.err
python - <<PY
import json
for c in ('c2', 'c1', 'c4'):
    try:
        d = json.loads(open('gpurun_out/${tag}_bench_line_%s.json' % c).read().strip().split('\n')[-1])
        m = d.get('multi_clip') or {}
        print(c, 'value', d['value'], 'all regions', d['repeats']['mean_fps_all_regions'], 'no_lookahead', d.get('value_no_lookahead'), 'conv ms', d['roofline']['ms_per_frame'], 'frac', d['roofline']['frac'],
              d['roofline']['executed_frac'], 'aff us', round(d['roofline_affinity']['ms_per_frame'] * 1e3, 1), d['roofline_affinity']['matmul']['mfma_util'], 'multi', m.get('value'), (m.get('roofline') or {}).get('frac'))
    except Exception as e:
        print(c, 'FAILED', e)
PY
