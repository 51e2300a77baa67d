# Round 5, call 17: issue priority also for the query-chain / element-wise launches of the caller's stream (library A/B against HEAD's kernels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c17
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lookahead or small_fifo or stages" > $O/p_tests.log 2>&1; tail -2 $O/p_tests.log
bash tools/ab.sh prio2 3 "CUTIE_AMD_LIB=$GRAFT_REPO_ROOT/tools/abl/libcutie_hip_OLD.so" "CUTIE_AMD_X=1" 2>&1 | tee $O/ab.log
bash tools/ab.sh window 2 "CUTIE_AMD_WINDOW_LEAD=3" "CUTIE_AMD_WINDOW_LEAD=4" "CUTIE_AMD_WINDOW_LEAD=2" 2>&1 | tee $O/ab_lead.log
