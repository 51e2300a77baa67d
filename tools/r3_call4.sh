cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c4
mkdir -p $O
CFG="3,30,54,256,256,3,100 3,30,54,256,256,3,120 1,30,54,256,256,3,100 1,30,54,1024,256,1,100 3,120,216,128,128,3,100 3,30,54,256,256,1,100 2,30,54,256,256,3,100"
for v in FULLQ FIXEPI FIXEPI_NOLOOP NO_LOOP NO_LOOP_EPI FULLQ FIXEPI; do
  CUTIE_AMD_LIB=tools/abl/libcutie_hip_$v.so timeout 120 python tools/multi_conv.py $CFG 2>&1 | grep -v amdgpu.ids >> $O/ablate.log
done
cat $O/ablate.log
