# stream priorities, A/B inside one box: CUTIE_AMD_MAIN_PRIO (bench's launch stream, torch numbering) x CUTIE_AMD_SIDE_PRIO (look-ahead stream, HIP numbering)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c45
mkdir -p $O
python - <<PY
import torch
print('torch priority range', torch.cuda.Stream.priority_range())
s = torch.cuda.Stream(priority=-1); print('torch -1 ->', s.priority)
PY
for w in "0 0" "-1 0" "0 1" "-1 1" "0 0" "-1 0" "0 1" "-1 1"; do
set -- $w
CUTIE_AMD_MAIN_PRIO=$1 CUTIE_AMD_SIDE_PRIO=$2 timeout 300 python bench.py --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1_$2.json').read().strip().split('\n')[-1])
    print("main $1 side $2:", d['value'], d.get('value_no_lookahead'))
except Exception as e:
    print("main $1 side $2: failed", e); print(open('$O/bench_$1_$2.err').read()[-800:])
PY
done
