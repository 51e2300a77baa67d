# Round 5, call 8: where does a hinted frame's time go now?  stream_waits (un-profiled: frame time by position in the memory cycle) and a
# kernel trace of hinted frames (launch list + durations of a memory frame and of a plain frame on the caller's queue).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c8
mkdir -p $O
timeout 120 python tools/stream_waits.py --window 12 --lead 3 --frames 300 2>&1 | tee $O/stream_waits.txt
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0 --repeats 1"
rm -rf /tmp/prof_r5c8
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r5c8 -- $BENCH > $O/trace.log 2>&1
tail -1 $O/trace.log | cut -c1-200
T=$(ls /tmp/prof_r5c8/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T 80 120 > $O/gaps.txt 2>&1; head -30 $O/gaps.txt
for f in 95 96 97 98 99 100; do python tools/trace_gaps.py $T 80 120 --dump $f > $O/frame_$f.txt 2>&1; done
gzip -c $T > $O/kernel_trace.csv.gz; ls -la $O/kernel_trace.csv.gz
