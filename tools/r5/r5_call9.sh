# Round 5, call 9: the object summarizer as two composed convs + SUMMARIZE without its dependent load chain
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c9
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "summarize" > $O/k_tests.log 2>&1; tail -3 $O/k_tests.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stages or small_fifo or small_lt or bike" > $O/p_tests.log 2>&1; tail -3 $O/p_tests.log; grep "stage errors" $O/p_tests.log | cut -c1-600
timeout 400 python -m pytest tests/test_gpu_teacher.py -x -q -m gpu -k "480 or small_fifo" > $O/t_tests.log 2>&1; tail -3 $O/t_tests.log
bash tools/ab.sh sumfused 2 "CUTIE_AMD_SUM_FUSED=0" "CUTIE_AMD_SUM_FUSED=1" 2>&1 | tee $O/ab.log
timeout 120 python tools/stream_waits.py --window 12 --lead 3 --frames 300 2>&1 | tee $O/stream_waits.txt
