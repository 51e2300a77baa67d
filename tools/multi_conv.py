"""Isolated device time of several conv configurations on the library $CUTIE_AMD_LIB names (diagnostic libraries of
tools/build_diag.sh).   python tools/multi_conv.py B,H,W,Cin,Cout,k,tile[,relu_in,res,gap] ..."""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
from cutie_amd.model.weights import pack_conv
ex = _lib.get_executor()
g = torch.Generator().manual_seed(0)
out = []
for spec in sys.argv[1:]:
    v = [int(a) for a in spec.split(',')]
    B, H, W, Cin, Cout, k, tile = v[:7]
    relu, res, gap = (v[7] if len(v) > 7 else 0), (v[8] if len(v) > 8 else 0), (v[9] if len(v) > 9 else 0)
    sums = torch.zeros((B, Cout), dtype=torch.int64, device='cuda') if gap else None
    pc = pack_conv(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), torch.zeros(Cout), 'cuda')
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
    y = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device='cuda')
    r = torch.randn(B, H, W, Cout, generator=g).to(torch.bfloat16).cuda() if res else None
    ol = O.OpList()
    ol.conv(x, pc, y, B=B, H=H, W=W, C1=Cin, ldx1=Cin, OH=H, OW=W, ldy=Cout, pad=(k - 1) // 2, tile=tile, relu_in=bool(relu), res=r, ldr=Cout,
            act=O.ACT_NONE if gap else O.ACT_RELU, gap_acc=sums)
    arr = ol.finalize()
    for _ in range(5):
        ex.run(arr)
    torch.cuda.synchronize()
    us = min(ex.time_ops(arr, 30) for _ in range(3)) * 1e3
    out.append(f'{spec}={us:.2f}')
print(os.path.basename(_lib.LIB_PATH), ' '.join(out))
