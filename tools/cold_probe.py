"""Conv kernels with COLD caches: every conv launch is preceded by a 192 MB device copy that evicts the conv's code and operands from the L2s
(and most of the Infinity Cache), as the ~1 GB of traffic between two uses of a layer does inside a frame.  Reported: time of
(flush + conv) minus the time of the flush alone, next to the warm back-to-back time of the same launch.
    python tools/cold_probe.py B,H,W,Cin,Cout,k,tile ..."""
import math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
from cutie_amd.model.weights import pack_conv
ex = _lib.get_executor()
g = torch.Generator().manual_seed(0)
src = torch.zeros(192 << 20, dtype=torch.uint8, device='cuda')
dst = torch.zeros(192 << 20, dtype=torch.uint8, device='cuda')
fl = O.OpList()
fl.copy2d(src, dst, rows=192, rowbytes=1 << 20, src_stride=1 << 20, dst_stride=1 << 20)
flush = fl.finalize()
for _ in range(3):
    ex.run(flush)
torch.cuda.synchronize()
t_flush = min(ex.time_ops(flush, 10) for _ in range(3)) * 1e3
print(f'flush alone: {t_flush:.1f} us')
for spec in sys.argv[1:]:
    v = [int(a) for a in spec.split(',')]
    B, H, W, Cin, Cout, k, tile = v[:7]
    pc = pack_conv(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), torch.zeros(Cout), 'cuda')
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
    y = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device='cuda')
    ol = O.OpList()
    ol.conv(x, pc, y, B=B, H=H, W=W, C1=Cin, ldx1=Cin, OH=H, OW=W, ldy=Cout, pad=(k - 1) // 2, tile=tile, act=O.ACT_RELU)
    arr = ol.finalize()
    for _ in range(3):
        ex.run(arr)
    torch.cuda.synchronize()
    warm = min(ex.time_ops(arr, 30) for _ in range(3)) * 1e3
    seq = np.concatenate([flush, arr])
    cold = min(ex.time_ops(seq, 10) for _ in range(3)) * 1e3 - t_flush
    print(f'{spec}: warm {warm:.2f} us, cold {cold:.2f} us (+{cold - warm:.2f})', flush=True)
