"""Does a launch pay for code that is not in the instruction cache?  One conv geometry on N DISTINCT kernels (tile ids): the sum of the
isolated (same kernel back to back: warm) times against the time of the N kernels launched in rotation (each kernel's code is evicted
by the others once their hot paths exceed the 64 KB instruction cache of a CU pair).    python tools/icache_probe.py"""
import math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
from cutie_amd.model.weights import pack_conv
ex = _lib.get_executor()
g = torch.Generator().manual_seed(0)


def probe(B, H, W, Cin, Cout, k, tiles, label):
    pc = pack_conv(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), torch.zeros(Cout), 'cuda')
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
    y = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device='cuda')
    arrs = []
    for t in tiles:
        ol = O.OpList()
        ol.conv(x, pc, y, B=B, H=H, W=W, C1=Cin, ldx1=Cin, OH=H, OW=W, ldy=Cout, pad=(k - 1) // 2, tile=t, act=O.ACT_RELU)
        arrs.append(ol.finalize())
    for a in arrs:
        for _ in range(3):
            ex.run(a)
    torch.cuda.synchronize()
    iso = [min(ex.time_ops(a, 30) for _ in range(3)) * 1e3 for a in arrs]
    out = [f'{label} {B}x{H}x{W} {Cin}->{Cout} k{k}: isolated sum of {len(tiles)} kernels {sum(iso):.1f} us']
    for n in (1, 2, 4, 8, len(tiles)):
        if n > len(tiles):
            continue
        seq = np.concatenate(arrs[:n])
        us = min(ex.time_ops(seq, 20) for _ in range(3)) * 1e3
        out.append(f'  rotation of {n:2d} distinct kernels: {us:.1f} us vs isolated {sum(iso[:n]):.1f} us  (+{(us - sum(iso[:n])) / n:.2f} us per launch)')
    # same kernel repeated n times in one list (no rotation) as control
    seq = np.concatenate([arrs[0]] * 8)
    us = min(ex.time_ops(seq, 20) for _ in range(3)) * 1e3
    out.append(f'  control: the same kernel 8x in one list: {us:.1f} us vs 8 x isolated {8 * iso[0]:.1f} us')
    print('\n'.join(out), flush=True)


probe(3, 30, 54, 256, 256, 3, [100, 101, 107, 110, 103, 104, 105, 106, 108, 120, 121, 129, 131, 132, 122, 123], 'conv_pc')
probe(3, 30, 54, 256, 256, 3, [66, 65, 72, 73, 74, 63, 64, 60, 61, 62, 70, 71, 76, 77, 86, 87], 'conv_dma')
probe(1, 30, 54, 1024, 256, 1, [100, 101, 107, 110, 103, 104, 105, 106, 108, 102, 109], 'conv_pc')
probe(1, 30, 54, 1024, 256, 1, [66, 65, 72, 73, 74, 63, 64, 60, 61, 62, 70], 'conv_dma')
