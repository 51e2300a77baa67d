# Round 4, final evidence on the final tree: rocprofv3 stats + PMC (profiles/r04_*), then the three bench lines, in one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/r4_profile.sh
bash tools/r4_lines.sh
