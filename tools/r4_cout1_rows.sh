# round 4, branch next/cout1-rows: conv_cout1_rows_kernel (Cout = 1, 3x3, large maps): kernel tests, cold / warm timing of the decoder's logits
# head (main's conv_cout1_kernel: 21.6 us cold, profiles/r03_conv_sweep_cold_480p_k3.txt), parity, bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4cout1
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "cout1 or stages or trajectory or 480p" > $O/1_tests.log 2>&1; tail -3 $O/1_tests.log
timeout 60 python tools/cold_probe.py 3,120,216,128,1,3,19 1,120,216,128,1,3,19 5,272,480,128,1,3,19 2>&1 | tee $O/cold.log | tail -4
for r in 1 2 3; do
timeout 300 python bench.py --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown > $O/bench_$r.json 2> $O/bench_$r.err
python - <<PY
import json
d=json.loads(open('$O/bench_$r.json').read().strip().split('\n')[-1])
print("run $r:", d['value'], d.get('value_no_lookahead'))
PY
done
