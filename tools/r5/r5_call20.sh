# Round 5, call 20: cold re-sweeps at 480p K = 1, 2 (the other object counts of the table)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c20
mkdir -p $O
timeout 900 python tools/conv_sweep.py --objects 1 2 --cold 160 --reps 3 --iters 12 --families pc,halo,dma --out $O/sweep_k12 > $O/sweep_k12.txt 2>&1; tail -2 $O/sweep_k12.txt
