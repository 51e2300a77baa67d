# Round 5, call 24: the Cout = 1 logits head on MFMA (conv_cout1_tile_kernel): kernel tests, isolated time, in-frame A/B against HEAD's library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c24
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cout1" > $O/k_tests.log 2>&1; tail -2 $O/k_tests.log
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher.py -x -q -m gpu -k "stages or small_fifo or bike or 480" > $O/p_tests.log 2>&1; tail -2 $O/p_tests.log
python - <<PY
import math, torch, sys
sys.path.insert(0, '.')
from cutie_amd import _lib, ops as O
from cutie_amd.model.weights import pack_conv
ex = _lib.get_executor()
pc = pack_conv(torch.randn(1, 128, 3, 3) / math.sqrt(1152), torch.zeros(1), 'cuda')
x = torch.randn(3, 120, 216, 128).to(torch.bfloat16).cuda(); y = torch.zeros(3, 120, 216, 1, dtype=torch.float32, device='cuda')
for off in (0, 128):
    O.F_TILE_OFF = off
    ol = O.OpList(); ol.conv(x, pc, y, B=3, H=120, W=216, C1=128, ldx1=128, OH=120, OW=216, ldy=1, pad=1, relu_in=True, out_f32=True, tile=O.COUT1_TILE)
    arr = ol.finalize()
    for _ in range(5): ex.run(arr)
    torch.cuda.synchronize()
    print('logits head, tile kernel' if off == 0 else 'logits head, rows kernel', 'warm us', round(min(ex.time_ops(arr, 20) for _ in range(3)) * 1e3, 2))
PY
bash tools/ab.sh cout1 3 "CUTIE_AMD_LIB=$GRAFT_REPO_ROOT/tools/abl/libcutie_hip_OLD.so" "CUTIE_AMD_X=1" 2>&1 | tee $O/ab.log
