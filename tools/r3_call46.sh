# next-weights touch: per-block byte budget (CUTIE_AMD_WPF_BLOCK), A/B inside one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c46
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "next_weights" > $O/1_kernels.log 2>&1; tail -2 $O/1_kernels.log
for w in 8388608 49152 24576 98304 8388608 49152 24576 98304; do
CUTIE_AMD_WPF_BLOCK=$w timeout 300 python bench.py --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown > $O/bench_$w.json 2> $O/bench_$w.err
python - <<PY
import json
d=json.loads(open('$O/bench_$w.json').read().strip().split('\n')[-1])
print("block budget $w:", d['value'], d.get('value_no_lookahead'), d['roofline']['ms_per_frame'])
PY
done
