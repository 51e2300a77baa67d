# Round 4, call 31: where does an UN-hinted frame's time go?  Kernel trace of the bench command; the no-hint leg follows the hinted frames
# (60 pre-roll + 10 warm-up + 60 timed hinted, then drain steps + 60 un-hinted frames on ONE queue): busy time, gaps, launch list.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c31
mkdir -p $O
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0 --repeats 1"
timeout 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r4c31 -- $BENCH > $O/trace.log 2>&1
tail -1 $O/trace.log | cut -c1-300
T=$(ls /tmp/prof_r4c31/*/*kernel_trace.csv | head -1)
gzip -c $T > $O/kernel_trace.csv.gz
ls -la $O/kernel_trace.csv.gz
