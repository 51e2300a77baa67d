"""s_memtime timeline of a conv kernel (conv_dma.hip tiles 60.., conv_pc.hip tiles 100..): where inside a block the time goes.
Needs the timeline library (bash tools/build_diag.sh TL=CONV_TIMELINE) and the MI355X:

    CUTIE_AMD_LIB=tools/abl/libcutie_hip_TL.so python tools/conv_timeline.py [--geo B,H,W,Cin,Cout,k ...] [--tiles 66 100 120]

Stamps (cutie_amd/csrc/conv_common.h TL): every wave of the blocks with logical tile id 0, 1, nb/2, nb-1.
  conv_dma : 0 entry | 1 prologue done | 2 ring primed + first sync | per K step: 3 start, 4 reads + MFMA + DMA issued, 5 counted wait done,
             (barrier) | 6 loop done | 7 fp32 tile in LDS + sync | 8 stores issued
  conv_pc  : consumer: 0 entry | 1 prologue done | 2 first barrier passed | per K step: 3 start, 4 reads + MFMAs issued (then lgkmcnt(0) +
             barrier) | 5 loop done | 6 epilogue issued.   producer: 0 | 1 prologue | 2 ring primed, tile 0 landed | per K step: 3 start,
             4 DMA issued, 5 counted wait done (then barrier) | 6 done
Output: per (tile, block) the segment means in cycles, per role."""
import argparse
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))

TL_MAX = 96


def parse(raw):
    """raw int64 [4 slots, 16 waves, TL_MAX + 2] -> {slot: {wave: [(id, cycles), ...]}}"""
    out = {}
    for s in range(4):
        for w in range(16):
            rec = raw[s, w]
            head = int(rec[0])
            if (head >> 16) != 0x544c:
                continue
            n = head & 0xffff
            st = [(int(v) & 0xff, int(v) >> 8) for v in rec[2:2 + min(n, TL_MAX)]]
            out.setdefault(s, {})[w] = dict(logical=int(rec[1]) >> 32, nwaves=int(rec[1]) & 0xffff, stamps=st)
    return out


def segments(st):
    """mean cycle deltas between consecutive stamp ids + the one-off segments"""
    seg = {}
    for (a, ta), (b, tb) in zip(st[:-1], st[1:]):
        seg.setdefault((a, b), []).append(tb - ta)
    return {k: (float(np.mean(v)), len(v)) for k, v in seg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--geo', nargs='+', default=['3,30,54,256,256,3', '3,120,216,128,128,3', '1,30,54,1024,256,1'])
    ap.add_argument('--tiles', type=int, nargs='+', default=[66, 100, 120])
    args = ap.parse_args()
    from cutie_amd import _lib, ops as O
    from cutie_amd.model.weights import pack_conv
    assert 'TL' in os.path.basename(_lib.LIB_PATH), 'set CUTIE_AMD_LIB to the timeline library (tools/build_diag.sh)'
    ex = _lib.HipExecutor()
    dev = 'cuda'
    for geo in args.geo:
        B, H, W, C, Cout, k = (int(v) for v in geo.split(','))
        g = torch.Generator().manual_seed(1)
        w = torch.randn(Cout, C, k, k, generator=g) / math.sqrt(C * k * k)
        pc = pack_conv(w, torch.randn(Cout, generator=g) * 0.1, dev, segs=[(C, C)])
        x = (torch.randn((B, H, W, C), generator=g)).to(torch.bfloat16).to(dev)
        y = torch.zeros((B, H, W, Cout), dtype=torch.bfloat16, device=dev)
        for tile in args.tiles:
            if tile in O.PC_TILES and not O.pc_tile_ok(tile, cin=C, kh=k):
                continue
            ol = O.OpList()
            ol.conv(x, pc, y, B=B, H=H, W=W, C1=C, ldx1=C, OH=H, OW=W, ldy=Cout, pad=(k - 1) // 2, tile=tile, act=O.ACT_RELU)
            arr = ol.finalize()
            scratch = O.splitk_scratch(pc.weight.device)
            for _ in range(3):
                ex.run(arr)
            torch.cuda.synchronize()
            us = ex.time_ops(arr, 20) * 1e3
            scratch[:4 * 16 * (TL_MAX + 2) * 2].zero_()
            ex.run(arr)
            torch.cuda.synchronize()
            raw = scratch[:4 * 16 * (TL_MAX + 2) * 2].view(torch.int64).cpu().numpy().reshape(4, 16, TL_MAX + 2)
            rec = parse(raw)
            print(f'\n==== conv B={B} {H}x{W} Cin={C} Cout={Cout} k={k}  tile {tile}  {us:.2f} us per launch (timeline build)')
            for s, waves in sorted(rec.items()):
                t0 = min(v['stamps'][0][1] for v in waves.values())
                t1 = max(v['stamps'][-1][1] for v in waves.values())
                any_w = next(iter(waves.values()))
                print(f'  block logical {any_w["logical"]}: {len(waves)} waves, span {t1 - t0} cycles')
                for wv, v in sorted(waves.items()):
                    st = v['stamps']
                    seg = segments(st)
                    nfirst = next((q for q, (i, _) in enumerate(st) if i == 3), 3)
                    first = ' '.join(f'{i}@{t - t0}' for i, t in st[:nfirst])
                    loop = ' '.join(f'{a}>{b}:{m:.0f}x{n}' for (a, b), (m, n) in sorted(seg.items()) if n > 1)
                    nl = max(q for q, (i, _) in enumerate(st) if i in (3, 4, 5)) if any(i in (3, 4, 5) for i, _ in st) else 0
                    tail = ' '.join(f'{i}@{t - t0}' for i, t in st[nl:])
                    print(f'    wave {wv:2d}: {first} | {loop} | {tail}')


if __name__ == '__main__':
    main()
