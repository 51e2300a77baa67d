cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c12
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "affinity" -s > $O/1_aff_tests.log 2>&1; grep -E "candidates per query|AssertionError|passed|failed" $O/1_aff_tests.log | head -12
timeout 300 python tools/aff_ab.py 300 > $O/2_aff_ab.log 2>&1; grep -E "^nq|score0 nq|readout alone|tokens" $O/2_aff_ab.log
