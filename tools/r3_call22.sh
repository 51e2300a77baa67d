cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c22
mkdir -p $O
for q in 0 1; do
CUTIE_AMD_QCHAIN=$q timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "trajectory" -s > $O/traj_q$q.log 2>&1
echo QCHAIN=$q; grep -E "max/mean|passed|failed|assert \(" $O/traj_q$q.log | cut -c1-400
done
