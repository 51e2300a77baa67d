# Round 5, call 21: the packaged tile table with the re-swept entries against the table of HEAD (in-frame, interleaved) + the conv bit-identity test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c21
mkdir -p $O
cp tools/abl/tiles_head.json /tmp/t_head.json
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_candidate or lookahead_window or stages or small_fifo or bike" > $O/tests.log 2>&1; tail -2 $O/tests.log
bash tools/ab.sh tiles2 3 "CUTIE_AMD_TILE_CACHE=/tmp/t_head.json" "CUTIE_AMD_X=1" 2>&1 | tee $O/ab.log
