cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c18
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lookahead or stages or small_fifo or small_lt or bike" > $O/1_tests.log 2>&1; tail -4 $O/1_tests.log
bash tools/ab.sh r4c18 3 "CUTIE_AMD_SEG_FORK=0" "CUTIE_AMD_SEG_FORK=1" 2>&1 | tee $O/2_ab.log
