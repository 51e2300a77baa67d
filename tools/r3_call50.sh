# tile 134 (320 x 64, 20 x 16 halo patches: the stride-4 maps of 3 objects in one round of workgroups): kernel tests, real layers, cold / warm timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c50
mkdir -p $O
timeout 50 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "134" > $O/1_kernels.log 2>&1; tail -2 $O/1_kernels.log
timeout 30 python tools/cold_probe.py 3,120,216,128,128,3,122 3,120,216,128,128,3,127 3,120,216,128,128,3,134 3,120,216,64,64,3,100 3,120,216,64,64,3,134 3,60,108,128,128,3,108 3,60,108,128,128,3,134 2>&1 | tee $O/cold.log | tail -8
