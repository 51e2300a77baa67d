# Round 5, call 13: isolated launch times of a memory frame (which conv of the composed summarizer is slow?) + consolidation tests after the LDS swizzle
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c13
mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "consolidation or rank_select" > $O/k_tests.log 2>&1; tail -2 $O/k_tests.log
timeout 200 python tools/frame_report.py --mem-frame > $O/mem_frame.txt 2>&1; tail -75 $O/mem_frame.txt | cut -c1-110
