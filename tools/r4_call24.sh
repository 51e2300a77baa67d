cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c24
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "stages or small_fifo or small_add_del or bike or query_init or lookahead_window or query_chain" > $O/1_tests.log 2>&1; tail -4 $O/1_tests.log
bash tools/ab.sh r4c24 3 "CUTIE_AMD_PROJ_X=0" "CUTIE_AMD_PROJ_X=1" 2>&1 | tee $O/2_ab.log
