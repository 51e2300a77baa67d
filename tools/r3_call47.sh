# next-weights touch on small grids (240p, 1 object): per-block byte budget against none against no touch, A/B inside one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c47
mkdir -p $O
for w in "8388608 8388608" "8388608 49152" "0 49152" "8388608 8388608" "8388608 49152" "0 49152"; do
set -- $w
CUTIE_AMD_WPF=$1 CUTIE_AMD_WPF_BLOCK=$2 timeout 300 python bench.py --height 240 --width 432 --objects 1 --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
python - <<PY
import json
d=json.loads(open('$O/bench_$1_$2.json').read().strip().split('\n')[-1])
print("240p 1 object, touch $1 per-block $2:", d['value'], d.get('value_no_lookahead'), d['roofline']['ms_per_frame'])
PY
done
