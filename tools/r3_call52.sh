# tile 134 among the candidates of the real small-map layers (the last GPU seconds of the round)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c52
timeout 60 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "every_conv_candidate" > gpurun_out/r3c52/1.log 2>&1; tail -3 gpurun_out/r3c52/1.log
