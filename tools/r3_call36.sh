# round 3, GPU call 36: full GPU suite on the current tree + default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c36
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/1_gpu.log 2>&1; tail -6 $O/1_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json | cut -c1-1500
