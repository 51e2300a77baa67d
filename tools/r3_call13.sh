cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c13
mkdir -p $O
timeout 1500 python tools/conv_sweep.py --objects 1 2 --cold 160 --reps 3 --iters 8 --families dma,pc,halo --out $O/sweepcold12 > $O/1_sweepcold12.log 2>&1
timeout 1500 python tools/conv_sweep.py --objects 5 --height 1080 --width 1920 --cold 160 --reps 3 --iters 6 --families dma,pc,halo --out $O/sweepcold1080 > $O/2_sweepcold1080.log 2>&1
tail -n 2 $O/1_sweepcold12.log $O/2_sweepcold1080.log
