# Round 5, call 18: cold re-sweep of the frame's conv geometries over the producer / consumer tiles (does the table still hold the winners?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c18
mkdir -p $O
timeout 900 python tools/conv_sweep.py --objects 3 --cold 160 --reps 3 --iters 12 --families pc,halo,dma --out $O/sweep_k3 > $O/sweep_k3.txt 2>&1; tail -3 $O/sweep_k3.txt
