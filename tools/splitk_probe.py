"""Time one conv shape over (tile, split-K) on the device.  python tools/splitk_probe.py B H W Cin Cout k"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O                      # noqa: E402
from cutie_amd.model.weights import pack_conv             # noqa: E402


def main():
    B, H, W, Cin, Cout, k = [int(a) for a in sys.argv[1:7]]
    dev = 'cuda'
    ex = _lib.get_executor()
    w = torch.randn(Cout, Cin, k, k) / math.sqrt(Cin * k * k)
    pc = pack_conv(w, torch.zeros(Cout), dev)
    x = torch.randn(B, H, W, Cin).to(torch.bfloat16).to(dev)
    y = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    for t in O.tile_candidates(B * H * W, Cout, Cin, pc.kpad, geom=dict(kh=k, stride=1, pad=(k - 1) // 2, W=W)):
        row = []
        for sk in (1, 2, 3):
            nk = pc.kpad // O.TILES[t][2] // O.TILE_WK.get(t, 1) if t in O.TILES else 0
            if t not in O.TILES or nk % sk or nk // sk < 1 or sk * B * H * W * Cout > O.SPLITK_PART_FLOATS:
                continue
            ol = O.OpList()
            ol.conv(x, pc, y, B=B, H=H, W=W, C1=Cin, ldx1=Cin, OH=H, OW=W, ldy=Cout, pad=(k - 1) // 2, tile=t, splitk=sk)
            arr = ol.finalize()
            us = min(ex.time_ops(arr, 10) for _ in range(3)) * 1e3
            row.append(f'sk{sk}:{us:6.1f}us({fl / us / 1e6:5.0f}TF)')
        print(t, O.TILES.get(t), 'wk%d' % O.TILE_WK.get(t, 1), ' '.join(row))


if __name__ == '__main__':
    main()
