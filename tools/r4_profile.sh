# Round 4 evidence: rocprofv3 kernel stats + PMC passes of the bench command (tools/profile_round.sh), then the summaries into gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04 > gpurun_out/r4_profile.log 2>&1
tail -5 gpurun_out/r4_profile.log
ls -la profiles/r04_* | head
mkdir -p gpurun_out/r4prof && cp profiles/r04_* gpurun_out/r4prof/
