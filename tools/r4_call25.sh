cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c25
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x -k "cout1 or stages or small_fifo or lookahead_window" > $O/1_tests.log 2>&1; tail -4 $O/1_tests.log
timeout 100 python tools/cold_probe.py 3,120,216,128,1,3,19 2>&1 | tail -2
CUTIE_AMD_COUT1_TILE=0 timeout 100 python tools/cold_probe.py 3,120,216,128,1,3,19 2>&1 | tail -1
bash tools/ab.sh r4c25 3 "CUTIE_AMD_COUT1_TILE=0" "CUTIE_AMD_COUT1_TILE=1" 2>&1 | tee $O/2_ab.log
