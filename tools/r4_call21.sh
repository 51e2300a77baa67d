cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c21
mkdir -p $O
timeout 600 python tools/conv_sweep.py --window 16 --cold 160 --reps 3 --iters 8 --families dma,pc --out $O/sweep_window16 > $O/1_sweep.log 2>&1; tail -3 $O/1_sweep.log
bash tools/ab.sh r4c21 2 "CUTIE_AMD_WINDOW=8 CUTIE_AMD_WINDOW_LEAD=2" "CUTIE_AMD_WINDOW=16 CUTIE_AMD_WINDOW_LEAD=3" "CUTIE_AMD_WINDOW=12 CUTIE_AMD_WINDOW_LEAD=3" 2>&1 | tee $O/2_ab.log
