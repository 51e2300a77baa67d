# Round 5, call 29: host-side trims, old tree (_ab_old = HEAD's package) against the working tree in ONE box, interleaved:
# (_ab_old/ = `git archive HEAD cutie_amd bench.py` + the built library, created for this call and removed after it)
# frames/s with / without hints, the reference's FPS protocol, four clips in flight
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c29
mkdir -p $O
for r in 1 2 3; do
  for v in old new; do
    B=bench.py; [ $v = old ] && B=_ab_old/bench.py
    python $B --steps 100 --warmup 10 --cpu-frames 0 --no-roofline --full-bank-preroll 0 > $O/${v}_$r.json 2> $O/${v}_$r.err
    python - <<PY
import json
d=json.loads(open('$O/${v}_$r.json').read().strip().splitlines()[-1])
print('$v', $r, d['value'], d['value_no_lookahead'], (d['no_lookahead'].get('eval_vos_protocol') or {}).get('fps'), 'multi', d.get('multi_clip',{}).get('value'))
PY
  done
done
