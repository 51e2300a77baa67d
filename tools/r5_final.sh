# Round 5, final tree: rocprofv3 summaries (tools/profile_round.sh r05), frame chain (stream_waits + kernel trace dumps), bench lines C2 / C1 / C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5final
mkdir -p $O
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
cp profiles/r05_summary.json profiles/r05_bench_kernel_stats.csv $O/ 2>/dev/null
timeout 120 python tools/stream_waits.py --window 12 --lead 3 --frames 300 2>&1 | tee $O/stream_waits.txt
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0 --repeats 1"
rm -rf /tmp/prof_r5f
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r5f -- $BENCH > $O/trace.log 2>&1
T=$(ls /tmp/prof_r5f/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $T 80 120 > $O/gaps.txt 2>&1; head -12 $O/gaps.txt
for f in 88 89 90 91 92 93 94; do python tools/trace_gaps.py $T 80 120 --dump $f > $O/frame_$f.txt 2>&1; done
python tools/trace_gaps.py $T 150 200 > $O/gaps_unhinted.txt 2>&1
for f in 160 161 162 163 164; do python tools/trace_gaps.py $T 150 200 --dump $f > $O/uframe_$f.txt 2>&1; done
gzip -c $T > $O/kernel_trace.csv.gz
timeout 600 python bench.py > $O/c2.json 2> $O/c2.err; tail -c 300 $O/c2.json
timeout 600 python bench.py --objects 1 --no-long-term --cpu-frames 0 --clips-in-flight 0 > $O/c1.json 2> $O/c1.err; tail -c 200 $O/c1.json
timeout 900 python bench.py --height 1080 --width 1920 --objects 5 --no-long-term --cpu-frames 0 --clips-in-flight 0 --preroll 100 > $O/c4.json 2> $O/c4.err; tail -c 200 $O/c4.json
python - <<PY
import json
for n in ('c2','c1','c4'):
    try:
        d=json.loads(open('$O/%s.json'%n).read().strip().split('\n')[-1])
        m=d['roofline_affinity']['matmul']
        print(n, d['value'], d.get('value_no_lookahead'), d['repeats']['values'], d['repeats']['mean_fps_all_regions'], 'conv', d['roofline']['ms_per_frame'], d['roofline']['frac'], d['roofline']['executed_frac'], d['roofline']['hbm_frac'],
              'aff', d['roofline_affinity']['ms_per_frame'], m['mfma_util'], m['frames_per_launch'], m['stage_us_per_frame'], d.get('multi_clip',{}).get('value'), d['config']['memory_tokens_end'], (d.get('full_bank') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(n,'FAILED',e, open('$O/%s.err'%n).read()[-500:])
PY
