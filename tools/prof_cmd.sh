# rocprofv3 kernel-trace statistics of ONE command on the MI355X box: per-kernel calls / total / average duration, top 45 by total time.
#   bash tools/prof_cmd.sh <name> <command ...>        -> gpurun_out/<name>_kernel_stats.csv (+ the top of it on stdout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
OUT=/tmp/prof_$name
rm -rf $OUT; mkdir -p $OUT gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT/cmd.log 2>&1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
if [ -z "$f" ]; then echo "no kernel stats"; tail -20 $OUT/cmd.log; exit 1; fi
cp $f gpurun_out/${name}_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/${name}_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.1f ms, %d kernels' % (tot / 1e6, len(rows)))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:45]:
    print('%8d %9.2f ms %8.2f us  %5.1f%%  %s' % (int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot, r['Name'][:150]))
PY
tail -2 $OUT/cmd.log
