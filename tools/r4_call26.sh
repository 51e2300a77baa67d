cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c26
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher.py -q -m gpu -x -k "lookahead or stages or small_fifo or small_lt or bank_contents or teacher_forced_scenarios" > $O/1_tests.log 2>&1; tail -4 $O/1_tests.log
bash tools/ab.sh r4c26 3 "CUTIE_AMD_SUM_FORK=0" "CUTIE_AMD_SUM_FORK=1" 2>&1 | tee $O/2_ab.log
