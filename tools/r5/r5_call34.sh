# Round 5, call 34: clips in flight, one stream per clip, child process with GPU_MAX_HW_QUEUES (bench.py --multi-hw-queues)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c34
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "concurrent or fork or lookahead or window" > $O/tests.log 2>&1; tail -2 $O/tests.log
for q in 16 32; do for c in 3 4 5 4; do
python bench.py --steps 100 --warmup 10 --cpu-frames 0 --no-roofline --full-bank-preroll 0 --clips-in-flight $c --multi-hw-queues $q 2>$O/err_${q}_$c.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['multi_clip']; print('queues', m.get('hw_queues'), 'clips', m.get('clips_in_flight_per_gpu'), d['value'], d['value_no_lookahead'], m.get('value'), m.get('error'))
"; done; done
