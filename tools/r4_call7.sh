# Round 4: kernel trace of the bench command with the look-ahead window -- the hinted frames (60 pre-roll + 10 warm-up + 60 timed: main-queue
# frames 1..130 carry hints; the no-hint leg follows).  Launch lists of two hinted frames, per-queue busy times, and the trace itself.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c7
mkdir -p $O
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0 --repeats 1"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4c7 -- $BENCH > $O/stats.log 2>&1
tail -2 $O/stats.log
T=$(ls /tmp/prof_r4c7/*/*kernel_trace.csv | head -1)
S=$(ls /tmp/prof_r4c7/*/*kernel_stats.csv | head -1)
cp $S $O/kernel_stats.csv
gzip -c $T > $O/kernel_trace.csv.gz
python tools/trace_gaps.py $T 80 120 --dump 101 > $O/gaps_101.txt 2>&1
python tools/trace_gaps.py $T 80 120 --dump 104 > $O/gaps_104.txt 2>&1
head -24 $O/gaps_101.txt
