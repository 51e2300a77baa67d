cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c9
mkdir -p $O
CUTIE_AMD_LIB=tools/abl/libcutie_hip_ATL.so timeout 200 python tools/aff_timeline.py 2:0 2:12 > $O/1_timeline.log 2>&1; grep -E "tiles per block|block 0 wave 0|block nb/2 wave 0" -A1 $O/1_timeline.log | cut -c1-1200
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "affinity_long" > $O/2_long.log 2>&1; grep -E "AssertionError|passed|failed" $O/2_long.log | head -5
