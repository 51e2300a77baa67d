cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c17
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bike_argmax" -s > $O/2_bike.log 2>&1
grep -E "bike frame|passed|failed|Error" $O/2_bike.log | tail -12
