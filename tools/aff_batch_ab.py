"""Isolated stage times of the affinity plan [score/0, select, score/1, readout] with F frames' queries stacked (one read-out per bank
version, MemoryManager._affinity_batch) against the one-frame plan, synthetic operands at the bench's sizes (480p: 1620 queries per frame,
K = 3).  Stage times by prefix differences of back-to-back replays (hipEvents on the launch stream).
    python tools/aff_batch_ab.py [tokens ...]        (on the MI355X)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
BF16, F32 = torch.bfloat16, torch.float32
dev = 'cuda'
HW, K, CV, cap, top_k = 1620, 3, 256, 1024, 30
HWp = -(-HW // 64) * 64
FMAX = 8
g = torch.Generator().manual_seed(7)
z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
ex = _lib.get_executor()
t1 = lambda a, it=20: min(ex.time_ops(a, it) for _ in range(4)) * 1e3
for ntok in ([int(a) for a in sys.argv[1:]] or [12200, 22500]):
    slots = ntok + 300
    ranges = [(0, ntok // 3), (ntok // 3 + 100, 1620), (ntok // 3 + 1720 + 100, ntok - ntok // 3 - 1620)]
    assert ranges[2][0] + ranges[2][1] <= slots
    mkey = (torch.randn((slots, 64), generator=g) * 0.8).to(dev); mshr = (torch.rand((slots,), generator=g) * 2 + 1).to(dev)
    qkey = (torch.randn((FMAX * HW, 64), generator=g) * 0.8).to(dev); qsel = torch.rand((FMAX * HW, 64), generator=g).to(dev)
    Ahi, Alo, scale = z((slots + 16, 128), BF16), z((slots + 16, 128), BF16), z((slots + 16,), F32)
    Bhi, Blo, cq = z((FMAX, HWp, 128), BF16), z((FMAX, HWp, 128), BF16), z((FMAX, HWp), F32)
    vals = [(torch.randn((slots + 16, CV), generator=g)).to(BF16).to(dev) for _ in range(K)]
    vptrs = torch.tensor([v.data_ptr() for v in vals], dtype=torch.int64).to(dev)
    prep = O.OpList()
    prep.key_prep(mkey, mshr, Ahi, Alo, scale, n=slots, query=False)
    for f in range(FMAX):
        prep.key_prep(qkey[f * HW:(f + 1) * HW], qsel[f * HW:(f + 1) * HW], Bhi[f], Blo[f], cq[f], n=HW, query=True)
    prep.run()
    G = sum(-(-n // 16) for _, n in ranges)
    Gld = -(-G // 64) * 64
    print(f'--- {ntok} tokens ({G} tiles), {HW} queries per frame, K = {K}')
    variants = [(2, 0)] if not os.environ.get('AFF_VARIANTS') else [(2, 0), (2, 1), (4, 1)]
    for F, (nq, dma) in [(F, v) for F in (1, 2, 4, 5, 8) for v in variants]:
        rows = F * HWp
        gbuf = z((rows * Gld + rows,), F32)
        gmax, tau = gbuf[:rows * Gld], gbuf[rows * Gld:]
        cval, cidx, count = z((rows, cap), F32), z((rows, cap), torch.int32), z((rows * 32,), torch.int32)
        use, y, ovf = z((F, slots + 16), F32), z((F, K, HW, CV), BF16), z((1,), torch.int32)
        common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=cap, frames=F, nq=nq, dma=dma)
        ol = O.OpList()
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, gmax, None, None, None, mode=0, **common)
        ol.aff_select(gmax, tau, HW=HW, HWp=HWp, G=G, top_k=top_k, clear_count=count, zero=(use, F * (slots + 16)), frames=F)
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, tau, cval, cidx, count, mode=1, gmax_precedes_tau=True, **common)
        ol.aff_readout(cval, cidx, count, vptrs, use, y, ovf, HW=HW, cap=cap, top_k=top_k, K=K, CV=CV, frames=F, HWp=HWp, usage_stride=slots + 16)
        a = ol.finalize()
        p = [t1(a[:k]) for k in range(1, 5)]
        st = [p[0], p[1] - p[0], p[2] - p[1], p[3] - p[2]]
        issued = 3 * 2.0 * 128 * G * 16 * rows
        print(f'nq {nq} dma {dma} ' + 'F %d: score0 %6.1f  select %5.1f  score1 %6.1f  readout %5.1f | plan %6.1f us = %5.1f us per frame | score0 mfma util %.3f' %
              (F, st[0], st[1], st[2], st[3], p[3], p[3] / F, issued / (st[0] * 1e-6) / 1e12 / 2500.0))
