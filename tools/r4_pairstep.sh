# round 4, first call on branch next/pc-pairstep: pair-step tiles 140..147 (two K tiles per barrier, tools/pc_ring_model.py): kernel tests,
# then cold / warm timing against the table's tiles on the frame's layer classes (tools/cold_probe.py: B,H,W,Cin,Cout,k,tile)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4pair
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "140 or 141 or 142 or 143 or 144 or 145 or 146 or 147" > $O/1_kernels.log 2>&1; tail -3 $O/1_kernels.log
timeout 120 python tools/cold_probe.py \
  3,30,54,256,256,3,129 3,30,54,256,256,3,145 3,30,54,256,256,3,110 3,30,54,256,256,3,141 3,30,54,256,256,3,142 3,30,54,256,256,3,101 3,30,54,256,256,3,140 \
  1,30,54,1024,256,1,102 1,30,54,1024,256,1,143 1,30,54,256,1024,1,101 1,30,54,256,1024,1,140 1,30,54,256,1024,1,110 1,30,54,256,1024,1,141 \
  1,30,54,256,256,3,109 1,30,54,256,256,3,143 1,30,54,256,256,3,130 1,30,54,256,256,3,147 \
  3,120,216,128,128,3,122 3,120,216,128,128,3,146 3,120,216,128,128,3,103 3,120,216,128,128,3,144 \
  3,60,108,128,128,3,108 3,60,108,128,128,3,144 3,60,108,128,128,3,141 2>&1 | tee $O/cold.log | tail -30
