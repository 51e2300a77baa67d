cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c14
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x -k "eca or stages or small_fifo or lookahead_window" > $O/1_tests.log 2>&1; tail -4 $O/1_tests.log
for a in "--window 8 --lead 2" "--window 8 --lead 2 --no-affinity-ahead" "--window 1" "--window 16 --lead 4"; do timeout 200 python tools/stream_waits.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O/2_waits.log; done
bash tools/ab.sh r4c14 2 "CUTIE_AMD_ECA_HEAD=0" "CUTIE_AMD_ECA_HEAD=1" 2>&1 | tee $O/3_ab.log
