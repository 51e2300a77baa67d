cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c16
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "query_init or bike_argmax or stages or small_add_del or small_chunk or small_flip or gap_acc or conv_gap or zero" > $O/1_tests.log 2>&1; tail -6 $O/1_tests.log
bash tools/ab.sh r4c16 2 "CUTIE_AMD_QINIT_SKIP=0" "CUTIE_AMD_QINIT_SKIP=1" 2>&1 | tee $O/2_ab.log
