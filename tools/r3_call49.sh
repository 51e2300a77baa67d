# next-weights touch at 1080p / 5 objects (on by size class) against none, and the size-class gate at 240p; quick GPU tests on the final tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c49
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x -k "next_weights or conv_gap or stages or trajectory or 480p or other_baseline" > $O/1_tests.log 2>&1; tail -2 $O/1_tests.log
for w in 8388608 0 8388608 0; do
CUTIE_AMD_WPF=$w timeout 300 python bench.py --height 1080 --width 1920 --objects 5 --steps 100 --warmup 10 --preroll 60 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown > $O/bench_1080_$w.json 2> $O/bench_1080_$w.err
python - <<PY
import json
d=json.loads(open('$O/bench_1080_$w.json').read().strip().split('\n')[-1])
print("1080p, 5 objects, touch $w:", d['value'], d.get('value_no_lookahead'), d['roofline']['ms_per_frame'])
PY
done
for w in 8388608 0; do
CUTIE_AMD_WPF=$w timeout 300 python bench.py --height 240 --width 432 --objects 1 --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown > $O/bench_240_$w.json 2> $O/bench_240_$w.err
python - <<PY
import json
d=json.loads(open('$O/bench_240_$w.json').read().strip().split('\n')[-1])
print("240p, 1 object (gate: off either way), CUTIE_AMD_WPF $w:", d['value'], d.get('value_no_lookahead'), d['roofline']['ms_per_frame'])
PY
done
