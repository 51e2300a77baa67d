# Round 4, call 32 (the last seconds of the budget): s_memtime timeline of the candidate pass (aff_score_kernel<2, 1>) -- diagnostic library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c32
CUTIE_AMD_LIB=$GRAFT_REPO_ROOT/tools/abl/libcutie_hip_ATL.so timeout 30 python tools/aff_timeline.py p1:2:0 2:0 > gpurun_out/r4c32/timeline.txt 2>&1
head -c 1500 gpurun_out/r4c32/timeline.txt
