# round 3, GPU call 8: in-frame per-kernel durations (rocprofv3 kernel trace) with the round-2 tile table and the merged round-3 table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c8
mkdir -p $O
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --no-lookahead"
for v in old new; do
  if [ $v = new ]; then export CUTIE_AMD_TILE_CACHE=$GRAFT_REPO_ROOT/gpurun_out_r3c6_tiles_merged.json; else unset CUTIE_AMD_TILE_CACHE; fi
  rm -rf /tmp/prof_$v
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- $BENCH > $O/stats_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/kernel_stats_$v.csv
done
head -40 $O/kernel_stats_old.csv | cut -c1-200
echo ----
head -40 $O/kernel_stats_new.csv | cut -c1-200
