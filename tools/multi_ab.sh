# A/B (/C ...) of environment settings for the SEVERAL-CLIPS-PER-GPU leg of bench.py (--multi-only) inside ONE gpurun call.
#   bash tools/multi_ab.sh <name> <reps> "<env of variant 1>" "<env of variant 2>" ... [-- <bench.py arguments>]
# e.g.  gpurun --timeout 900 -- 'bash tools/multi_ab.sh lswin 2 "CUTIE_AMD_LS_WINDOW=3 CUTIE_AMD_LS_LEAD=1" "CUTIE_AMD_LS_WINDOW=6" -- --multi-mode lockstep'
# Prints, per run, the aggregate frames/s of the leg (clips x steps / seconds).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; reps=$2; shift 2
variants=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do variants+=("$1"); shift; done
[ "$1" = "--" ] && shift
O=gpurun_out/mab_$name
mkdir -p $O
for r in $(seq 1 $reps); do
  for i in "${!variants[@]}"; do
    env ${variants[$i]} timeout 400 python bench.py --multi-only --steps 200 --warmup 5 --preroll 150 "$@" > $O/v${i}_r$r.json 2> $O/v${i}_r$r.err
    python - <<PY
import json, sys
args = "$*".split()
C = int(args[args.index('--clips-in-flight') + 1]) if '--clips-in-flight' in args else 4
ls = '--multi-mode' in args and args[args.index('--multi-mode') + 1] == 'lockstep'
C *= max(1, int(args[args.index('--lockstep-groups') + 1])) if '--lockstep-groups' in args else (3 if ls else 1)     # (bench.py's default: three lock-step groups in flight)
try:
    d = json.loads(open('$O/v${i}_r$r.json').read().strip().split('\n')[-1])
    print('[${variants[$i]}] run $r: %.1f frames/s (%d clips x %d steps in %.3f s)' % (C * d['steps_per_clip'] / d['seconds'], C, d['steps_per_clip'], d['seconds']))
except Exception as e:
    print('[${variants[$i]}] run $r: FAILED', e, open('$O/v${i}_r$r.err').read()[-600:])
PY
  done
done
