# Round 5, call 4: pass 0 on two tiles at a time + policy "first frame alone, the rest of the memory cycle batched"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c4
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "affinity" > $O/aff_tests.log 2>&1; tail -3 $O/aff_tests.log
timeout 200 python tools/aff_batch_ab.py 12200 22500 2>&1 | tee $O/aff_batch_ab.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lookahead" > $O/la_tests.log 2>&1; tail -3 $O/la_tests.log
bash tools/ab.sh affbatch 2 "CUTIE_AMD_AFF_BATCH=1" "CUTIE_AMD_AFF_BATCH=8" "CUTIE_AMD_AFF_BATCH=8 CUTIE_AMD_AFF_FIRST_ALONE=0" 2>&1 | tee $O/ab.log
