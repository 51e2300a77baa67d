# Round 5, call 37: clips in flight driven by ONE host thread in turn (parallel.run_interleaved / bench.py --multi-mode interleaved)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c37
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "interleaved or concurrent_clips or lookahead_window_matches" > $O/tests.log 2>&1; tail -2 $O/tests.log
run() { python bench.py --steps 100 --warmup 10 --cpu-frames 0 --no-roofline --full-bank-preroll 0 --no-graph --clips-in-flight $2 --multi-mode $1 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['multi_clip']; print('$1', 'clips', m.get('clips_in_flight_per_gpu'), d['value'], m.get('value'), m.get('error'))
" || tail -5 $O/err.txt; }
for r in 1 2 3; do run interleaved 4; done
run interleaved 2; run interleaved 3; run interleaved 6; run interleaved 8
run threads 2
