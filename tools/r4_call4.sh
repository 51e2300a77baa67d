# Round 4, call 4: aff_score4_kernel (64 queries per wave, LDS-DMA staging): parity of the affinity pipeline for nq = 1, 2, 4; isolated stage times.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c4
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "affinity" > $O/1_aff_tests.log 2>&1; tail -12 $O/1_aff_tests.log
timeout 300 python tools/aff_ab.py 300 > $O/2_aff_ab.log 2>&1; tail -22 $O/2_aff_ab.log
