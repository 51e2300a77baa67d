"""s_memtime timeline of aff_score4_kernel<0> (pass 0 of the affinity read-out) at the bench's working point: where inside a block the
time goes.  Needs the timeline library (bash tools/build_diag_aff.sh) and the MI355X:
    CUTIE_AMD_LIB=tools/abl/libcutie_hip_ATL.so python tools/aff_timeline.py [tiles per block ...]
Stamps: 0 entry | 1 first A group requested (DMA issued) | 2 B fragments + c requested | per group: 3 loop top, 4 barrier passed (group
landed), 5 next group's DMA + previous maxima stores issued, 6..9 tile 0..3 starts, 10 tiles done | 11 end.
"p1:2:0" as an argument: PASS 1 of aff_score_kernel<2, 1> (the candidate pass, with the tile skipping of the frame) behind a real pass 0 and
selection: the same stamps, then 12 = the wave's candidate flush is through (global atomics answered), 13 + 16 n = the wave flushed
16 n ... 16 n + 15 candidates.  Round 4 left pass 1 (30-44 us for ~15 % of pass 0's MFMA work) without a timeline of its own."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
BF16, F32 = torch.bfloat16, torch.float32
ATL_MAX = 95                                                 # (= ATL_MAX of csrc/affinity.hip)
g = torch.Generator().manual_seed(7)
HW, slots = 1620, 11400
ranges = [(0, 2000), (2100, 1620), (4000, 7400)]
HWp = -(-HW // 64) * 64
dev = 'cuda'
mkey = (torch.randn((slots, 64), generator=g) * 0.8).to(dev); mshr = (torch.rand((slots,), generator=g) * 2 + 1).to(dev)
qkey = (torch.randn((HW, 64), generator=g) * 0.8).to(dev); qsel = torch.rand((HW, 64), generator=g).to(dev)
z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
Ahi, Alo, scale = z((slots + 16, 128), BF16), z((slots + 16, 128), BF16), z((slots + 16,), F32)
Bhi, Blo, cq = z((HWp, 128), BF16), z((HWp, 128), BF16), z((HWp,), F32)
G = sum(-(-n // 16) for _, n in ranges)
Gld = -(-G // 64) * 64
gmax = z((HWp, Gld), F32)
prep = O.OpList()
prep.key_prep(mkey, mshr, Ahi, Alo, scale, n=slots, query=False)
prep.key_prep(qkey, qsel, Bhi, Blo, cq, n=HW, query=True)
prep.run()
ex = _lib.get_executor()
def decode(raw, label):
    for slot in range(2):
        for w in range(4):
            head = int(raw[slot, w, 0])
            if (head >> 16) != 0x4154:
                continue
            n = head & 0xffff
            st = [(int(v) & 0xff, int(v) >> 8) for v in raw[slot, w, 1:1 + min(n, ATL_MAX)]]
            t0 = st[0][1]
            line = ' '.join(f'{i}:{t - t0}' for i, t in st)
            print(f'  {label} block {"0" if slot == 0 else "nb/2"} wave {w}: {n} stamps, total {st[-1][1] - t0} cycles\n    {line}')


for spec in (sys.argv[1:] or ['0']):
    if spec.startswith('p1:'):                               # "p1:2:0": pass 1 with 2 query sets per wave, tiles per block from the heuristic
        _, nq, tpb = spec.split(':')
        nq, tpb = int(nq), int(tpb)
        top_k, cap = 30, 1024
        gbuf = z((HWp * Gld + HWp,), F32)
        gm, tau = gbuf[:HWp * Gld].view(HWp, Gld), gbuf[HWp * Gld:HWp * Gld + HW]
        cval, cidx, count = z((HW, cap), F32), z((HW, cap), torch.int32), z((HW * 32,), torch.int32)
        dbg = torch.zeros((2 * 4 * (ATL_MAX + 1),), dtype=torch.int64, device=dev)
        common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=cap)
        ol = O.OpList()
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, gm, None, None, None, mode=0, nq=nq, **common)
        ol.aff_select(gm, tau, HW=HW, HWp=HWp, G=G, top_k=top_k, clear_count=count)
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, tau, cval, cidx, count, mode=1, gmax_precedes_tau=True, nq=nq, **common)
        arr = ol.finalize()
        arr['i'][0, 13] = arr['i'][2, 13] = tpb
        arr['p'][2, 10] = dbg.data_ptr()                     # (diagnostic library: ScoreParams.tl)
        for _ in range(3):
            ex.run(arr)
        torch.cuda.synchronize()
        t01 = min(ex.time_ops(arr[:2], 20) for _ in range(3)) * 1e3
        t012 = min(ex.time_ops(arr, 20) for _ in range(3)) * 1e3
        dbg.zero_()
        ex.run(arr)
        torch.cuda.synchronize()
        cn = count.view(HW, 32)[:, 0].float()
        print(f'PASS 1, nq {nq}, tiles per block {tpb}: G {G}, pass 1 {t012 - t01:.2f} us (difference of back-to-back replays); candidates per query: '
              f'mean {float(cn.mean()):.1f}, max {int(cn.max())}, total {int(cn.sum())}')
        decode(dbg.cpu().numpy().reshape(2, 4, ATL_MAX + 1), 'pass 1')
        continue
    nq = 4
    if ':' in spec:                                          # "2:20": the 32-query kernel (aff_score_kernel<2, 0>), 20 tiles per block
        nq, spec = int(spec.split(':')[0]), spec.split(':')[1]
    tpb, _, pad = spec.partition('+')                        # "24+70": 24 tiles per block, 70 KB of extra dynamic LDS (one block per CU)
    tpb, pad = int(tpb), int(pad or 0)
    dbg = torch.zeros((2 * 4 * (ATL_MAX + 1),), dtype=torch.int64, device=dev)
    ol = O.OpList()
    ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, gmax, dbg, None, None, mode=0, nq=nq, HW=HW, HWp=HWp, ranges=ranges, cap=1024)
    arr = ol.finalize()
    arr['i'][0, 13] = tpb
    arr['i'][0, 14] = pad
    for _ in range(3):
        ex.run(arr)
    torch.cuda.synchronize()
    us = min(ex.time_ops(arr, 20) for _ in range(3)) * 1e3
    dbg.zero_()
    ex.run(arr)
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy().reshape(2, 4, ATL_MAX + 1)
    print(f'nq {nq}, tiles per block {tpb}, extra LDS {pad} KB: G {G}, kernel {us:.2f} us (launch-to-launch, back to back)')
    for slot in range(2):
        for w in range(4):
            head = int(raw[slot, w, 0])
            if (head >> 16) != 0x4154:
                continue
            n = head & 0xffff
            st = [(int(v) & 0xff, int(v) >> 8) for v in raw[slot, w, 1:1 + min(n, ATL_MAX)]]
            t0 = st[0][1]
            line = ' '.join(f'{i}:{t - t0}' for i, t in st)
            print(f'  block {"0" if slot == 0 else "nb/2"} wave {w}: {n} stamps, total {st[-1][1] - t0} cycles\n    {line}')
