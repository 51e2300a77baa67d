cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c30
mkdir -p $O
timeout 300 python tools/defer_probe.py > $O/probe.log 2>&1; tail -6 $O/probe.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "deferred" > $O/1_parity.log 2>&1; tail -5 $O/1_parity.log
