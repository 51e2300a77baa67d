# next-weights touch: kernel tests, then A/B inside one box of the second range (CUTIE_AMD_WPF2) against one range and none
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c44
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "next_weights or conv_gap" > $O/1_kernels.log 2>&1; tail -2 $O/1_kernels.log
for w in "0 0" "8388608 0" "8388608 1" "0 0" "8388608 0" "8388608 1"; do
set -- $w
CUTIE_AMD_WPF=$1 CUTIE_AMD_WPF2=$2 timeout 300 python bench.py --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
python - <<PY
import json
d=json.loads(open('$O/bench_$1_$2.json').read().strip().split('\n')[-1])
print("$w", d['value'], d.get('value_no_lookahead'))
PY
done
