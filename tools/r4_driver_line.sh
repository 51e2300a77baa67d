# the driver's own bench command (20 timed steps after 5 warm-up steps), timed by the wall clock around it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4drv
mkdir -p $O
for n in 1 2; do
t0=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/line$n.json 2> $O/err$n.txt
t1=$(date +%s.%N)
python - <<PY
import json
d=json.loads(open("$O/line$n.json").read().strip().split("\n")[-1])
print("wall %.1f s" % ($t1 - $t0), d["value"], d["value_no_lookahead"], d["repeats"]["values"], "full", d["full_bank"]["value"], "conv", d["roofline"]["ms_per_frame"], d["roofline"]["frac"],
      "aff", d["roofline_affinity"]["ms_per_frame"], d["roofline_affinity"]["matmul"]["mfma_util"], "cpu", d["cpu_baseline"]["value"], "multi", d.get("multi_clip", {}).get("value"))
PY
done
