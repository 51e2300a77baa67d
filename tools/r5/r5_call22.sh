# Round 5, call 22: cold re-sweeps of the batched encoder (window 12) and of the 1080p / 5-object frame
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c22
mkdir -p $O
timeout 600 python tools/conv_sweep.py --window 12 --cold 160 --reps 3 --iters 8 --families pc,halo,dma --out $O/sweep_w12 > $O/sweep_w12.txt 2>&1; tail -2 $O/sweep_w12.txt
timeout 900 python tools/conv_sweep.py --objects 5 --height 1080 --width 1920 --cold 160 --reps 2 --iters 6 --families pc,halo,dma --out $O/sweep_1080 > $O/sweep_1080.txt 2>&1; tail -2 $O/sweep_1080.txt
