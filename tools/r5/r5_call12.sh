# Round 5, call 12: frames without hints -- the frame's own read-out on the side stream next to the tail of its encoder (SELF_AHEAD)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c12
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tools/ab.sh selfahead 3 "CUTIE_AMD_SELF_AHEAD=0" "CUTIE_AMD_SELF_AHEAD=1" 2>&1 | tee $O/ab.log
