"""What does the fused input ReLU (CONV flags&1: conv(F.relu(x))) cost on the frame's 3x3 convs?  The LDS-DMA path brings the operand into
LDS untouched, so the ReLU is applied to every A fragment after it is read (VALU work in the MFMA loop).  Same tile, same geometry, with and
without the flag; warm replays and cold ones (a 160 MB flush in front of every timed launch).
    python tools/relu_in_ab.py            (on the MI355X)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
from cutie_amd.model.weights import pack_conv
ex = _lib.get_executor()
flush = torch.zeros(160 << 20, dtype=torch.uint8, device='cuda')
def cold(arr, n=12):
    ts = []
    for _ in range(n):
        flush.add_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ex.run(arr); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
# (B, H, W, Cin, Cout, k, tiles to try)
cases = [(3, 120, 216, 128, 128, 3, (121, 126, 122, 134)), (3, 60, 108, 256, 128, 3, (120, 124, 127)), (3, 30, 54, 256, 256, 3, (120, 124, 125))]
for B, H, W, Cin, Cout, k, tiles in cases:
    pc = pack_conv(torch.randn(Cout, Cin, k, k) / math.sqrt(Cin * k * k), torch.zeros(Cout), 'cuda')
    x = torch.randn(B, H, W, Cin).to(torch.bfloat16).cuda()
    y = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device='cuda')
    for tile in tiles:
        if tile not in O.ALL_TILES or not O.pc_tile_ok(tile, cin=Cin, kh=k):
            continue
        out = []
        for relu in (False, True):
            ol = O.OpList()
            ol.conv(x, pc, y, B=B, H=H, W=W, C1=Cin, ldx1=Cin, OH=H, OW=W, ldy=Cout, pad=1, tile=tile, relu_in=relu)
            arr = ol.finalize()
            try:
                for _ in range(5): ex.run(arr)
                torch.cuda.synchronize()
                out.append((min(ex.time_ops(arr, 20) for _ in range(3)) * 1e3, cold(arr)))
            except Exception as e:
                out.append((float('nan'), float('nan')))
        print(f'M={B*H*W} {Cin}->{Cout} k{k} tile {tile} {O.ALL_TILES[tile]}: plain warm {out[0][0]:.1f} cold {out[0][1]:.1f} | relu_in warm {out[1][0]:.1f} cold {out[1][1]:.1f}')
