# round 4, branch next/bank-write: BANK_WRITE (the copies and fills of a memory insertion in one launch): kernel test, bank parity, A/B is
# not possible by environment (the op replaces the copy2d sequence): compare bench lines of main and of this branch inside one call by
# checking both trees out side by side, or run the default bench here and tools/ab.sh on main in the same call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4bank
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "bank_write or bank_contents or trajectory or stages" > $O/1_tests.log 2>&1; tail -3 $O/1_tests.log
for r in 1 2 3; do
timeout 300 python bench.py --steps 400 --cpu-frames 0 --clips-in-flight 0 --full-bank-preroll 0 --no-breakdown > $O/bench_$r.json 2> $O/bench_$r.err
python - <<PY
import json
d=json.loads(open('$O/bench_$r.json').read().strip().split('\n')[-1])
print("run $r:", d['value'], d.get('value_no_lookahead'))
PY
done
