"""Launch ONE conv configuration repeatedly (target of tools/pmc_kernel.sh).  python tools/one_conv.py B H W Cin Cout k tile [reps]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
from cutie_amd.model.weights import pack_conv
B, H, W, Cin, Cout, k, tile = [int(a) for a in sys.argv[1:8]]
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
ex = _lib.get_executor()
pc = pack_conv(torch.randn(Cout, Cin, k, k) / math.sqrt(Cin * k * k), torch.zeros(Cout), 'cuda')
x = torch.randn(B, H, W, Cin).to(torch.bfloat16).cuda()
y = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device='cuda')
ol = O.OpList()
ol.conv(x, pc, y, B=B, H=H, W=W, C1=Cin, ldx1=Cin, OH=H, OW=W, ldy=Cout, pad=(k - 1) // 2, tile=tile)
arr = ol.finalize()
for _ in range(reps):
    ex.run(arr)
torch.cuda.synchronize()
print('us', ex.time_ops(arr, 20) * 1e3)
