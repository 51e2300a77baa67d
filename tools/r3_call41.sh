# round 3, GPU call 41: robustness -- bench under torch.distributed.run (1 rank), C1 (1 object) and C4 (1080p, 5 objects) configurations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c41
mkdir -p $O
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0 > $O/torchrun.json 2> $O/torchrun.err; tail -c 400 $O/torchrun.json; tail -2 $O/torchrun.err
for cfg in "--objects 1" "--height 1080 --width 1920 --objects 5 --preroll 60"; do
  timeout 500 python bench.py $cfg --steps 60 --warmup 10 --cpu-frames 0 --no-roofline --clips-in-flight 0 --full-bank-preroll 0 > $O/b.json 2> $O/b.err
  python - <<PY
import json
d=json.loads(open('$O/b.json').read().strip().split('\n')[-1])
print('$cfg', d['value'], d.get('value_no_lookahead'), d['config']['workload'][:60])
PY
  tail -1 $O/b.err
done
