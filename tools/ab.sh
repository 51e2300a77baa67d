# A/B (/C ...) of environment settings inside ONE gpurun call (boxes of the pool differ by up to 15 %: only such comparisons count).
#   bash tools/ab.sh <name> <reps> "<env of variant 1>" "<env of variant 2>" ... [-- <bench.py arguments>]
# e.g.  gpurun --timeout 600 -- 'bash tools/ab.sh wpf 3 "CUTIE_AMD_WPF=0" "CUTIE_AMD_WPF=8388608" -- --objects 1'
# Every variant runs <reps> times, interleaved; per run: frames/s with and without the look-ahead hint, conv family ms per frame.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; reps=$2; shift 2
variants=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done
[ "$1" = "--" ] && shift
O=gpurun_out/ab_$name
mkdir -p $O
for r in $(seq 1 $reps); do
  for i in "${!variants[@]}"; do
    env ${variants[$i]} timeout 300 python bench.py --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown "$@" > $O/v${i}_r$r.json 2> $O/v${i}_r$r.err
    python - <<PY
import json
try:
    d = json.loads(open('$O/v${i}_r$r.json').read().strip().split('\n')[-1])
    print('[${variants[$i]}] run $r:', d['value'], d.get('value_no_lookahead'), (d.get('roofline') or {}).get('ms_per_frame'))
except Exception as e:
    print('[${variants[$i]}] run $r: FAILED', e, open('$O/v${i}_r$r.err').read()[-600:])
PY
  done
done
