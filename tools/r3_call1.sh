# round 3, GPU call 1: first run of conv_pc.hip (producer / consumer + halo tiles): kernel tests, timelines, per-layer sweep, frame report
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c1
mkdir -p $O
# canaries first: a deadlocked barrier protocol must not eat the whole call
timeout 120 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "test_conv_pc_tiles and (0-100 or 0-120)" > $O/0_canary.log 2>&1
echo "canary rc=$?" >> $O/0_canary.log
if grep -q "passed" $O/0_canary.log && ! grep -q "failed" $O/0_canary.log; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --maxfail=40 -k "conv_pc or gap_accum" > $O/1_tests.log 2>&1
else
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --maxfail=60 -k "conv_pc" > $O/1_tests.log 2>&1
fi
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu --maxfail=10 -k "affinity or bank_misc or lookahead" > $O/2_look.log 2>&1
CUTIE_AMD_LIB=tools/abl/libcutie_hip_TL.so timeout 400 python tools/conv_timeline.py --tiles 66 65 100 101 107 120 121 122 123 > $O/3_timeline.log 2>&1
timeout 1500 python tools/conv_sweep.py --objects 3 --families dma,strip,pc,halo --out $O/conv_sweep > $O/4_sweep.log 2>&1
timeout 300 python tools/frame_report.py > $O/5_frame.log 2>&1
tail -3 $O/0_canary.log $O/1_tests.log $O/2_look.log
tail -30 $O/4_sweep.log
