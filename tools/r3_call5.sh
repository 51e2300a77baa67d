# round 3, GPU call 5: conv_pc with fast epilogue paths, W-first prologue, patch shapes for <= 256 workgroups
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c5
mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "test_conv_pc_tiles and (0-100 or 0-120)" > $O/0_canary.log 2>&1
echo "canary rc=$?" >> $O/0_canary.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --maxfail=40 -k "conv_pc or gap_accum" > $O/1_tests.log 2>&1
CFG="3,30,54,256,256,3,100 3,30,54,256,256,3,120 1,30,54,256,256,3,100 1,30,54,1024,256,1,100 3,120,216,128,128,3,100 3,30,54,256,256,1,100 2,30,54,256,256,3,100"
for v in FULLQ NO_LOOP NO_LOOP_EPI NO_DMA; do
  CUTIE_AMD_LIB=tools/abl/libcutie_hip_$v.so timeout 120 python tools/multi_conv.py $CFG 2>&1 | grep -v amdgpu.ids >> $O/ablate.log
done
CUTIE_AMD_LIB=tools/abl/libcutie_hip_TL.so timeout 400 python tools/conv_timeline.py --tiles 100 110 129 > $O/3_timeline.log 2>&1
timeout 1500 python tools/conv_sweep.py --objects 3 --families dma,pc,halo --out $O/conv_sweep > $O/4_sweep.log 2>&1
tail -n 3 $O/0_canary.log $O/1_tests.log
cat $O/ablate.log
tail -n 50 $O/4_sweep.log
