# round 3, last validation of the final tree (host-side changes only since tools/r3_final.sh ran): full GPU suite, smoke, default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final2
mkdir -p $O
timeout 200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/2_smoke.log 2>&1; tail -2 $O/2_smoke.log
timeout 400 python -m pytest tests -q -m gpu > $O/1_gpu.log 2>&1; tail -3 $O/1_gpu.log
