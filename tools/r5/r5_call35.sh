# Round 5, call 35: clips in flight on streams with dedicated hardware queues (hipExtStreamCreateWithCUMask, all CUs)
# (the CUTIE_BENCH_DEDICATED switch lived in bench.py for this call only: hipExtStreamCreateWithCUMask through ctypes, torch.cuda.ExternalStream)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c35
mkdir -p $O
run() { python bench.py --steps 100 --warmup 10 --cpu-frames 0 --no-roofline --full-bank-preroll 0 --no-graph --clips-in-flight $2 --multi-hw-queues $1 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['multi_clip']; print('dedicated', '$CUTIE_BENCH_DEDICATED', 'queues', m.get('hw_queues'), 'clips', m.get('clips_in_flight_per_gpu'), d['value'], m.get('value'), m.get('error'))
" || tail -5 $O/err.txt; }
export CUTIE_BENCH_DEDICATED=1
for r in 1 2 3; do run 0 4; done
run 0 3; run 0 6; run 0 8
for r in 1 2; do run 16 4; done
