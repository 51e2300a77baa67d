"""Time the two convs of a CAResBlock of four clips in lock step (19440 x 256 x 256, 3x3; conv1 clears the ECA accumulator, conv2 adds its stored
output to it) with and without the side jobs, on the tiles the plans use.   python tools/gap_conv_ab.py      (A/B of libraries: $CUTIE_AMD_LIB)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
from cutie_amd.model.weights import pack_conv
ex = _lib.get_executor()
B, H, W, C = 12, 30, 54, 256
g = torch.Generator().manual_seed(1)
pc = pack_conv(torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9), torch.randn(C, generator=g) * 0.1, 'cuda', segs=[(C, C)])
x = (torch.randn(B, H, W, C, generator=g) * 0.5).to(torch.bfloat16).cuda()
y = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device='cuda')
sums = torch.zeros((B, C), dtype=torch.int64, device='cuda')
t1 = lambda a: min(ex.time_ops(a, 20) for _ in range(5)) * 1e3
for tile in (104, 122, 146, 129):
    if tile not in O.ALL_TILES:
        continue
    row = []
    for kw in (dict(), dict(zero=sums), dict(gap_acc=sums)):
        ol = O.OpList()
        try:
            ol.conv(x, pc, y, B=B, H=H, W=W, C1=C, ldx1=C, OH=H, OW=W, ldy=C, pad=1, tile=tile, **kw)
        except Exception as e:
            row.append('n/a'); continue
        arr = ol.finalize()
        ex.run(arr); torch.cuda.synchronize()
        row.append('%.1f' % t1(arr))
    print('tile %d: plain %s us, clearing the accumulator %s us, adding to it %s us' % (tile, *row))
