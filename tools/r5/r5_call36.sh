# Round 5, call 36: clips in flight with staggered schedules (clip i is i * 3 frames further into its encoder batches / memory cycles)
# (the CUTIE_BENCH_STAGGER switch lived in bench.py for this call only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c36
mkdir -p $O
run() { python bench.py --steps 100 --warmup 10 --cpu-frames 0 --no-roofline --full-bank-preroll 0 --no-graph --clips-in-flight $2 --multi-hw-queues $1 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['multi_clip']; print('stagger', '$CUTIE_BENCH_STAGGER', 'queues', m.get('hw_queues'), 'clips', m.get('clips_in_flight_per_gpu'), d['value'], m.get('value'), m.get('error'))
" || tail -5 $O/err.txt; }
export CUTIE_BENCH_STAGGER=3
for r in 1 2 3; do run 0 4; done
for r in 1 2 3; do run 16 4; done
run 0 2; run 0 3; run 16 3
export CUTIE_BENCH_STAGGER=0
run 0 4; run 16 4
