# Round 4, call 3: kernel trace of the bench command with the look-ahead window (8 frames): per-kernel stats and the main queue's
# launch list of two frames (tools/trace_gaps.py --dump) -- what is the critical path of a frame made of?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c3
mkdir -p $O
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0 --repeats 1"
$BENCH > $O/plain.json 2> $O/plain.err; tail -c 600 $O/plain.json
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4c3 -- $BENCH > $O/stats.log 2>&1
tail -3 $O/stats.log
T=$(ls /tmp/prof_r4c3/*/*kernel_trace.csv | head -1)
S=$(ls /tmp/prof_r4c3/*/*kernel_stats.csv | head -1)
cp $S $O/kernel_stats.csv
python tools/trace_gaps.py $T 150 190 --dump 171 > $O/gaps_171.txt 2>&1
python tools/trace_gaps.py $T 150 190 --dump 174 > $O/gaps_174.txt 2>&1
head -30 $O/gaps_171.txt
