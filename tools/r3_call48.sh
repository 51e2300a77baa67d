# next-weights touch against none at two intermediate frame sizes (where do the weights start to go cold between two uses?), A/B inside one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c48
mkdir -p $O
for cfg in "480 854 1" "240 432 3" "360 640 2"; do
set -- $cfg
for w in 8388608 0 8388608 0; do
CUTIE_AMD_WPF=$w timeout 300 python bench.py --height $1 --width $2 --objects $3 --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown > $O/bench_$1_$3_$w.json 2> $O/bench_$1_$3_$w.err
python - <<PY
import json
d=json.loads(open('$O/bench_$1_$3_$w.json').read().strip().split('\n')[-1])
print("$1 x $2, $3 objects, touch $w:", d['value'], d.get('value_no_lookahead'), d['roofline']['ms_per_frame'])
PY
done
done
