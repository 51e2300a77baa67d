# Round 5, call 1: measure what round 4 left unmeasured.
#   1. UPSAMPLE2X_ADD quad form: bit-identity test, then A/B inside this box (tools/ab.sh)
#   2. s_memtime timelines of both score passes at two blocks per CU (diagnostic library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c1
mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "up2 or upsample" > $O/up2_test.log 2>&1; tail -2 $O/up2_test.log
bash tools/ab.sh up2quad 2 "CUTIE_AMD_UP2_QUAD=0" "CUTIE_AMD_UP2_QUAD=1" 2>&1 | tee $O/ab.log
CUTIE_AMD_LIB=$GRAFT_REPO_ROOT/tools/abl/libcutie_hip_ATL.so timeout 90 python tools/aff_timeline.py 2:0 p1:2:0 > $O/aff_timeline.txt 2>&1; head -c 3000 $O/aff_timeline.txt
