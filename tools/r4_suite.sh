# the whole GPU suite on the current tree (what the driver runs at round end), plus the observed-error record for the ratchet
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4suite
mkdir -p $O
CUTIE_REBASE_OBSERVED=${REBASE:-} CUTIE_RECORD_OBSERVED=$GRAFT_REPO_ROOT/$O/observed.json timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/suite.log 2>&1; tail -15 $O/suite.log
