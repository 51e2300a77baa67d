# Round 5, call 11: memory frames with the mask encoder on the auxiliary stream next to the sensory update (MEM_FORK)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c11
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lookahead or overwritten or small_fifo or small_lt" > $O/p_tests.log 2>&1; tail -3 $O/p_tests.log
CUTIE_AMD_ARENA_POISON=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lookahead" > $O/p_tests_poison.log 2>&1; tail -3 $O/p_tests_poison.log
bash tools/ab.sh memfork 3 "CUTIE_AMD_MEM_FORK=0" "CUTIE_AMD_MEM_FORK=1" 2>&1 | tee $O/ab.log
timeout 120 python tools/stream_waits.py --window 12 --lead 3 --frames 300 2>&1 | tee $O/stream_waits.txt
