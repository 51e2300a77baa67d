# tile 134 on the two 64 -> 64 stride-4 decoder convs (table entry): the frame's parity tests at 480p with it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c51
mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stages or 480p or trajectory" > $O/1_parity.log 2>&1; tail -2 $O/1_parity.log
