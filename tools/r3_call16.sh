cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c16
mkdir -p $O
CUTIE_RECORD_OBSERVED=$O/observed_r03.json timeout 1500 python -m pytest tests/test_gpu_teacher.py -q -m gpu --maxfail=20 -s > $O/1_teacher.log 2>&1
tail -n 6 $O/1_teacher.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bike_argmax" -s > $O/2_bike.log 2>&1
grep -E "bike frame|passed|failed|Error" $O/2_bike.log | tail -12
