cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c17
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lookahead or query_init or bike_argmax or 480p or trajectory" > $O/1_tests.log 2>&1; tail -6 $O/1_tests.log
bash tools/ab.sh r4c17 2 "CUTIE_AMD_MEM_SPLIT=0" "CUTIE_AMD_MEM_SPLIT=1" 2>&1 | tee $O/2_ab.log
