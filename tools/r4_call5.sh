cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c5
mkdir -p $O
timeout 200 python tools/aff_debug.py > $O/1_debug.log 2>&1; tail -20 $O/1_debug.log
timeout 300 python tools/aff_ab.py 300 > $O/2_aff_ab.log 2>&1; tail -22 $O/2_aff_ab.log
