cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c23
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lookahead or small_lt or 480p or bank_contents" > $O/1_tests.log 2>&1; tail -4 $O/1_tests.log
bash tools/ab.sh r4c23 3 "CUTIE_AMD_COMMIT_SIDE=0" "CUTIE_AMD_COMMIT_SIDE=1" 2>&1 | tee $O/2_ab.log
