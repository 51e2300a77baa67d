cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c31
mkdir -p $O
timeout 300 python tools/host_profile.py > $O/host.log 2>&1; head -60 $O/host.log | cut -c1-170
