cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c37
mkdir -p $O
for v in "CUTIE_AMD_QCHAIN=0" "CUTIE_AMD_QCHAIN=1 CUTIE_AMD_QNEXT=0" "CUTIE_AMD_QCHAIN=1 CUTIE_AMD_QNEXT=1" "CUTIE_AMD_QCHAIN=1 CUTIE_AMD_POOL_SLOTS=0"; do
  echo "== $v"
  env $v CUTIE_RECORD_OBSERVED=$O/obs.json timeout 300 python -m pytest tests/test_gpu_teacher.py -q -m gpu -k "small_chunk or small_fifo-base or small_lt-base" -s 2>&1 | grep -E "passed|failed|small_chunk|observed" | head -5
  python - <<PY
import json
d=json.load(open('$O/obs.json'))
for k in sorted(d):
    if 'chunk' in k or 'fifo' in k or k.endswith('small_lt'): print('   ', k, {a: round(b, 5) for a, b in d[k].items()})
PY
done
