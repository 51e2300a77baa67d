cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c20
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -x -k "lookahead or stages or small_fifo or small_lt or bike or gru or query_init" > $O/1_tests.log 2>&1; tail -4 $O/1_tests.log
bash tools/ab.sh r4c20 3 "CUTIE_AMD_SENS_AHEAD=0" "CUTIE_AMD_SENS_AHEAD=1" 2>&1 | tee $O/2_ab.log
