# A/B inside one box: conv_pc's next-weights touch (CUTIE_AMD_WPF bytes; 0 = off)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c42
mkdir -p $O
CUTIE_AMD_WPF=1048576 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "teacher or golden" > $O/1_parity.log 2>&1; tail -2 $O/1_parity.log
for w in 0 1048576 0 4194304 262144; do
CUTIE_AMD_WPF=$w timeout 300 python bench.py --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 > $O/bench_$w.json 2> $O/bench_$w.err
python - <<PY
import json
d=json.loads(open('$O/bench_$w.json').read().strip().split('\n')[-1])
print($w, d['value'], d.get('value_no_lookahead'), d['roofline']['ms_per_frame'], d['roofline']['frac'])
PY
done
