# is the four-clips-in-flight leg sensitive to the last two changes, or to the box's host?  (full bench lines, one box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c27
mkdir -p $O
for v in "CUTIE_AMD_PROJ_X=1 CUTIE_AMD_COUT1_TILE=1" "CUTIE_AMD_PROJ_X=0 CUTIE_AMD_COUT1_TILE=0" "CUTIE_AMD_PROJ_X=1 CUTIE_AMD_COUT1_TILE=1" "CUTIE_AMD_PROJ_X=0 CUTIE_AMD_COUT1_TILE=0"; do
  env $v timeout 400 python bench.py --full-bank-preroll 0 --cpu-frames 6 --no-breakdown > $O/line.json 2> $O/line.err
  python - <<PY
import json
d = json.loads(open('$O/line.json').read().strip().split('\n')[-1])
print('[$v]', d['value'], d['value_no_lookahead'], 'multi', d['multi_clip'].get('value'), 'cpu', d['cpu_baseline']['value'])
PY
done
