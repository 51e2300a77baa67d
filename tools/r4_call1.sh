# Round 4, call 1: the look-ahead WINDOW of the image encoder (InferenceCore.prefetch_window, plans.build_encode(B=...)).
#   1. bit-identity tests (window vs plain order; tiles of one K-order class agree bitwise on the real layers)
#   2. A/B inside this box: next_image (window 1) | window 4, lead 1 | window 4, lead 2 | window 8, lead 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c1
mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lookahead" > $O/1_lookahead.log 2>&1; tail -15 $O/1_lookahead.log
timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "every_conv_candidate" > $O/2_classes.log 2>&1; tail -8 $O/2_classes.log
bash tools/ab.sh r4c1 2 "CUTIE_AMD_WINDOW=1" "CUTIE_AMD_WINDOW=4 CUTIE_AMD_WINDOW_LEAD=1" "CUTIE_AMD_WINDOW=4 CUTIE_AMD_WINDOW_LEAD=2" \
  "CUTIE_AMD_WINDOW=8 CUTIE_AMD_WINDOW_LEAD=2" 2>&1 | tee $O/3_ab.log
