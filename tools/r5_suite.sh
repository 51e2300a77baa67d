# Round 5: the whole GPU suite (what the driver runs at round end) + smoke + the driver's command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5suite
mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/ -q -m gpu --maxfail=10 > $O/suite.log 2>&1; tail -4 $O/suite.log; grep -E "^(FAILED|ERROR)" $O/suite.log | head -12
t1=$(date +%s); echo "suite: $((t1 - t0)) s"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/r5_driver_line.sh
