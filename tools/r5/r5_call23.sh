# Round 5, call 23: final tile table against round 4's (in-frame, interleaved), C2 and C4; conv / parity tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c23
mkdir -p $O
cp tools/abl/tiles_head.json /tmp/t_head.json
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "every_conv_candidate or lookahead_window or stages or small_fifo or bike or conv_next_weights" > $O/tests.log 2>&1; tail -2 $O/tests.log
bash tools/ab.sh tiles3 3 "CUTIE_AMD_TILE_CACHE=/tmp/t_head.json" "CUTIE_AMD_X=1" 2>&1 | tee $O/ab.log
bash tools/ab.sh tiles3c4 2 "CUTIE_AMD_TILE_CACHE=/tmp/t_head.json" "CUTIE_AMD_X=1" -- --height 1080 --width 1920 --objects 5 --no-long-term --preroll 100 --steps 100 2>&1 | tee $O/ab_c4.log
