# round 3, GPU call 33: is the look-ahead frame device-bound or launch-bound?  kernel trace with timestamps, gaps per queue
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c33
mkdir -p $O
BENCH="python bench.py --steps 100 --warmup 20 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --no-breakdown --full-bank-preroll 0"
rm -rf /tmp/prof_t
CUTIE_AMD_GRAPHS=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- $BENCH > $O/trace.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
ls -la $f; head -2 $f | cut -c1-400
python tools/trace_gaps.py $f 90 170 > $O/gaps.txt 2>&1; cat $O/gaps.txt | cut -c1-200
tail -2 $O/trace.log | cut -c1-300
