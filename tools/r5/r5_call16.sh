cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c16
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "argmax_decisive" > $O/argmax.log 2>&1; grep -E "frame|decisive:|passed|failed|Error|assert" $O/argmax.log | cut -c1-220 | tail -40
