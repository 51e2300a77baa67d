cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c22
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x -k "up4 or mask_down or lookahead or stages or small_fifo or small_add_del or bike or query_init" > $O/1_tests.log 2>&1; tail -4 $O/1_tests.log
timeout 600 python tools/conv_sweep.py --window 12 --cold 160 --reps 3 --iters 8 --families dma,pc --out $O/sweep_window12 > $O/2_sweep.log 2>&1; tail -2 $O/2_sweep.log
bash tools/ab.sh r4c22 3 "CUTIE_AMD_SEG_MD=0" "CUTIE_AMD_SEG_MD=1" 2>&1 | tee $O/3_ab.log
