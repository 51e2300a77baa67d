# Round 5, call 14: issue priority for the launches of the caller's stream (CUTIE_F_PRIO / s_setprio 1) against the look-ahead work that
# shares the compute units; consolidation after the LDS swizzle
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c14
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "affinity or conv_pc or consolidation" > $O/k_tests.log 2>&1; tail -2 $O/k_tests.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lookahead or small_fifo or small_lt" > $O/p_tests.log 2>&1; tail -2 $O/p_tests.log
bash tools/ab.sh prio 3 "CUTIE_AMD_PRIO=0" "CUTIE_AMD_PRIO=1" 2>&1 | tee $O/ab.log
CUTIE_AMD_PRIO=0 timeout 120 python tools/stream_waits.py --window 12 --lead 3 --frames 300 2>&1 | tail -4 | tee $O/stream_waits_prio0.txt
timeout 120 python tools/stream_waits.py --window 12 --lead 3 --frames 300 2>&1 | tail -4 | tee $O/stream_waits_prio1.txt
