"""aff_score4_kernel against aff_score_kernel<2, *> on the same operands: where do the pass-0 maxima differ?  (MI355X box)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import ops as O
BF16, F32 = torch.bfloat16, torch.float32
g = torch.Generator().manual_seed(7)
for HW, ranges, slots in ((1620, [(0, 1620)], 1620), (1620, [(0, 2000), (2100, 1620), (4000, 8097)], 12200), (100, [(0, 1003)], 1003)):
    HWp = -(-HW // 64) * 64
    dev = 'cuda'
    mkey = (torch.randn((slots, 64), generator=g) * 0.8).to(dev); mshr = (torch.rand((slots,), generator=g) * 2 + 1).to(dev)
    qkey = (torch.randn((HW, 64), generator=g) * 0.8).to(dev); qsel = torch.rand((HW, 64), generator=g).to(dev)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    Ahi, Alo, scale = z((slots + 16, 128), BF16), z((slots + 16, 128), BF16), z((slots + 16,), F32)
    Bhi, Blo, cq = z((HWp, 128), BF16), z((HWp, 128), BF16), z((HWp,), F32)
    G = sum(-(-n // 16) for _, n in ranges if n > 0)
    Gld = -(-G // 64) * 64
    res = {}
    for nq in (2, 4):
        gmax = torch.full((HWp, Gld), 7.0, dtype=F32, device=dev)
        ol = O.OpList()
        ol.key_prep(mkey, mshr, Ahi, Alo, scale, n=slots, query=False)
        ol.key_prep(qkey, qsel, Bhi, Blo, cq, n=HW, query=True)
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, gmax, None, None, None, mode=0, nq=nq, HW=HW, HWp=HWp, ranges=ranges, cap=1024)
        ol.run()
        torch.cuda.synchronize()
        res[nq] = gmax[:HW, :G].clone().cpu()
    a, b = res[2], res[4]
    bad = ~((a == b) | (torch.isinf(a) & torch.isinf(b)))
    print(f'HW {HW} G {G}: {int(bad.sum())} of {bad.numel()} maxima differ; untouched (7.0) entries in nq 4: {int((b == 7.0).sum())}')
    if bad.any():
        q, t = bad.nonzero(as_tuple=True)
        print('  by tile % 4:', [int((t % 4 == k).sum()) for k in range(4)], ' by (query // 16) % 4:', [int(((q // 16) % 4 == k).sum()) for k in range(4)],
              ' by (query // 64) % 4:', [int(((q // 64) % 4 == k).sum()) for k in range(4)], ' by query % 16:', [int((q % 16 == k).sum()) for k in range(16)])
        print('  tiles hit:', sorted(set(int(x) for x in t))[:40], ' first:', [(int(q[i]), int(t[i]), float(a[q[i], t[i]]), float(b[q[i], t[i]])) for i in range(min(6, len(q)))])
