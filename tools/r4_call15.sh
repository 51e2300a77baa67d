cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c15
mkdir -p $O
timeout 200 python tools/stream_waits.py --window 8 --lead 2 2>&1 | grep -v amdgpu.ids | tee $O/1_waits.log
CUTIE_AMD_GRAPHS=1 timeout 200 python tools/stream_waits.py --window 8 --lead 2 2>&1 | grep -v amdgpu.ids | tee -a $O/1_waits.log
bash tools/ab.sh r4c15 2 "CUTIE_AMD_GRAPHS=0" "CUTIE_AMD_GRAPHS=1" 2>&1 | tee $O/2_ab.log
