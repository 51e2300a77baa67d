OUT=gpurun_out/c40; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/1_smoke.log 2>&1; tail -2 $OUT/1_smoke.log
timeout 600 python bench.py --objects 1 --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-roofline > $OUT/bench_c1.json 2> $OUT/bench_c1.err
timeout 900 python bench.py --height 1080 --width 1920 --objects 5 --steps 60 --warmup 10 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
for f in c1 c4; do python -c "
import json; d=json.loads(open('$OUT/bench_$f.json').read().strip().split('\n')[-1]); print('$f:', d['value'], 'fps', d['ms_per_step'], 'no-lookahead', d.get('value_no_lookahead'), d['config'].get('memory_tokens_end'), (d.get('roofline') or {}).get('frac'))"; done
