cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c8
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "affinity" > $O/1_aff_tests.log 2>&1; tail -12 $O/1_aff_tests.log
timeout 300 python tools/aff_ab.py 300 > $O/2_aff_ab.log 2>&1; tail -24 $O/2_aff_ab.log
