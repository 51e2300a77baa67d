# round 3, GPU call 3: conv_pc ablations (which part of the split loop bounds a layer)
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c3
mkdir -p $O
CFG="3,30,54,256,256,3,100 3,30,54,256,256,3,120 1,30,54,256,256,3,100 1,30,54,1024,256,1,100 3,120,216,128,128,3,100 3,120,216,128,128,3,120 3,30,54,256,256,1,100"
for v in FULLQ NO_DMA NO_MFMA NO_READ NO_READ_MFMA SAME_TILE NO_LOOPBAR NO_WAIT EMPTY NO_LOOP NO_LOOP_EPI NO_EPI FULLQ; do
  CUTIE_AMD_LIB=tools/abl/libcutie_hip_$v.so timeout 120 python tools/multi_conv.py $CFG 2>&1 | grep -v amdgpu.ids >> $O/ablate.log
done
cat $O/ablate.log
