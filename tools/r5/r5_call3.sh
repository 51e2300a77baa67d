# Round 5, call 3: affinity after the grid-sizing / barrier changes: kernel tests, stacked-frame stage times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c3
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "affinity" > $O/aff_tests.log 2>&1; tail -3 $O/aff_tests.log
timeout 200 python tools/aff_batch_ab.py 12200 22500 2>&1 | tee $O/aff_batch_ab.txt
