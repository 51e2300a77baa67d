# Round 5, call 10: does the stacked read-out stall the caller's stream by filling every CU's LDS?  A/B: its score launches at one block
# per CU (80 KB of extra dynamic LDS) against two; MASK_DOWN reuse in encode_mask is in both.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c10
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lookahead or overwritten or small_fifo" > $O/p_tests.log 2>&1; tail -3 $O/p_tests.log
bash tools/ab.sh afflds 2 "CUTIE_AMD_AFF_BATCH_LDS=0" "CUTIE_AMD_AFF_BATCH_LDS=80" "CUTIE_AMD_AFF_BATCH_LDS=80 CUTIE_AMD_AFF_BATCH_FORMS=0" 2>&1 | tee $O/ab.log
CUTIE_AMD_AFF_BATCH_LDS=80 timeout 120 python tools/stream_waits.py --window 12 --lead 3 --frames 300 2>&1 | tee $O/stream_waits_lds80.txt
timeout 120 python tools/stream_waits.py --window 12 --lead 3 --frames 300 2>&1 | tee $O/stream_waits.txt
