# Round 5, call 28: host-side trims (slot pool, bind, frame_context, raw stream handle, cached commit launch): the reference's FPS
# protocol and the bench line before / after would need two trees -- here: the new tree's timeline + bench line + the GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c28
mkdir -p $O
python tools/host_timeline.py > $O/unhinted.txt 2>&1; head -12 $O/unhinted.txt
python tools/host_timeline.py --hints > $O/hinted.txt 2>&1; head -12 $O/hinted.txt
python bench.py --steps 100 --warmup 10 --cpu-frames 0 --no-roofline > $O/line.json 2> $O/line.err
python - <<PY
import json
d=json.loads(open('$O/line.json').read().strip().splitlines()[-1])
print(d['value'], d['repeats']['values'], d['no_lookahead'], d.get('multi_clip',{}).get('value'), (d.get('full_bank') or {}).get('value'))
PY
python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; tail -3 $O/suite.log
