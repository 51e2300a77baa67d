# round 3, GPU call 19: timeline of the query-side launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c19
mkdir -p $O
CUTIE_AMD_LIB=tools/abl/libcutie_hip_ATL.so timeout 300 python tools/attn_timeline.py > $O/timeline.log 2>&1
cat $O/timeline.log
