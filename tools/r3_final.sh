# round 3, final validation: observed-error ratchet recorded on this build, full GPU suite, smoke, default bench line, rocprofv3 evidence
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final
mkdir -p $O
if [ "${RECORD:-1}" = "1" ]; then    # (RECORD=0: the build computes what the committed ratchet was recorded on, e.g. after a change that only moves bytes)
CUTIE_RECORD_OBSERVED=$O/observed_r03.json timeout 900 python -m pytest tests/test_gpu_teacher.py -q -m gpu > $O/0_record.log 2>&1; tail -3 $O/0_record.log
cp $O/observed_r03.json tests/golden/observed_r03.json
fi
timeout 2400 python -m pytest tests -q -m gpu > $O/1_gpu.log 2>&1; tail -5 $O/1_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/2_smoke.log 2>&1; tail -2 $O/2_smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
export CUTIE_TREE=${CUTIE_TREE:-unknown}
timeout 1500 bash tools/profile_round.sh r03 > $O/3_profile.log 2>&1; tail -4 $O/3_profile.log
cp profiles/r03_bench_kernel_stats.csv profiles/r03_summary.json $O/ 2>/dev/null
ls -la $O
