# round 4, branch next/up4-vec: UP4_SOFTMAX (fused) with four pixels per thread: bit-identity test, bench against main in the same call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4up4
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "seg_epilogue or stages or trajectory or 480p" > $O/1_tests.log 2>&1; tail -3 $O/1_tests.log
for r in 1 2 3; do
timeout 300 python bench.py --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 > $O/bench_$r.json 2> $O/bench_$r.err
python - <<PY
import json
d=json.loads(open('$O/bench_$r.json').read().strip().split('\n')[-1])
print("run $r:", d['value'], d.get('value_no_lookahead'), d['device_us_by_kind'].get('UP4_SOFTMAX'))
PY
done
