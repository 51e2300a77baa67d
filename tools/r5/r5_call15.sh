# Round 5, call 15: pass 1 of the score kernel -- both query sets' chains interleaved when a tile is needed, next tile's fragments requested ahead
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c15
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "affinity" > $O/k_tests.log 2>&1; tail -2 $O/k_tests.log
timeout 100 python tools/aff_batch_ab.py 12200 22500 2>&1 | tee $O/aff_batch_ab.txt
CUTIE_AMD_LIB=$GRAFT_REPO_ROOT/tools/abl/libcutie_hip_ATL.so timeout 90 python tools/aff_timeline.py p1:2:0 > $O/aff_timeline_p1.txt 2>&1; grep -A2 "pass 1 block 0 wave 0" $O/aff_timeline_p1.txt | cut -c1-900; grep "PASS 1" $O/aff_timeline_p1.txt
