"""Model check of the producer / consumer ring protocols of cutie_amd/csrc/conv_pc.hip (no GPU needed).

The producer and consumer loops are transcribed as event lists; between two consecutive workgroup barriers ("interval j" = after
barrier j, before barrier j + 1) the two roles run concurrently.  Checked, for every combination of K tiles, ring depth, patch pieces
and slices:
  * both roles execute the same number of barriers;
  * a consumer reads tile t from stage S in interval j only if every load of tile t into S was covered by a producer wait
    (in-order vmcnt) BEFORE the producer reached barrier j, and no load into S was issued in an interval <= j after those;
  * the same for the halo patch buffers (slice s in buffer s & 1).
A load may land any time after its issue; it is only GUARANTEED landed once a counted wait has retired it (vmcnt retires in order).

    python tools/pc_ring_model.py          # all combinations; prints the number checked, raises on the first violation
"""
import itertools
import sys


class Producer:
    def __init__(self):
        self.fifo = []            # outstanding loads, oldest first: dict(kind, where, what, issued_iv)
        self.done = []            # retired loads with 'ok_iv' = interval in which the covering wait completed
        self.iv = -1              # current interval (-1 = before barrier 0)
        self.barriers = 0

    def issue(self, kind, where, what, n=1):
        # n machine loads make up the piece / tile; they retire together for our purpose
        self.fifo.append(dict(kind=kind, where=where, what=what, issued=self.iv, n=n))

    def wait(self, allowed):
        """s_waitcnt vmcnt(allowed): retire the oldest loads until at most `allowed` machine loads are outstanding."""
        out = sum(l['n'] for l in self.fifo)
        while self.fifo and out > allowed:
            l = self.fifo[0]
            # a multi-load entry retires only when ALL of its loads are within the retired prefix
            if out - l['n'] < allowed:
                # partially covered entry: not guaranteed -- stays outstanding (conservative)
                break
            out -= l['n']
            l['ok'] = self.iv
            self.done.append(self.fifo.pop(0))

    def barrier(self):
        self.iv += 1
        self.barriers += 1


def stream_producer(nk, NS, KP, LPT=2):
    """Stream mode: one entry per tile (LPT machine loads per producer wave and tile).  Stage of tile t = t % NS."""
    P = Producer()
    for s in range(NS - 1):
        if s < nk:
            P.issue('T', s % NS, s, LPT)
    if KP == 1:
        if NS == 3 and nk >= NS - 1:
            # queue order W(0..NS-2), X(0..NS-2): waiting for all but the X pieces of tile 1 completes tile 0.  Modelled per tile:
            # tile 0 guaranteed, tile 1 not.
            P.wait((NS - 2) * LPT)
        else:
            P.wait(0)
        P.barrier()
        ld = min(nk, NS - 1) % NS
        kt = 0
        while kt < nk - (NS - 1):
            P.issue('T', ld, kt + NS - 1, LPT)
            P.wait((NS - 2) * LPT)
            P.barrier()
            ld = (ld + 1) % NS
            kt += 1
        while kt < nk:
            P.wait(0)
            P.barrier()
            kt += 1
    else:
        NP = NS // 2
        P.wait(0)
        P.barrier()
        t = min(nk, NS - 1)
        ld = t % NS
        npairs = (nk + 1) >> 1
        u = 0
        while 2 * (u + NP - 1) + 1 < nk:
            while t <= 2 * (u + NP - 1) + 1:
                P.issue('T', ld, t, LPT)
                ld = (ld + 1) % NS
                t += 1
            P.wait((NP - 2) * 2 * LPT)
            P.barrier()
            u += 1
        while u < npairs:
            while t < nk and t <= 2 * (u + NP - 1) + 1:
                P.issue('T', ld, t, LPT)
                ld = (ld + 1) % NS
                t += 1
            P.wait(0)
            P.barrier()
            u += 1
    return P


def halo_producer(nslice, NS, KP, NXP, NWI=2):
    """Halo mode: weight tile of step t in ring stage t % NS (NWI loads), patch pieces of slice s in buffer s & 1 (1 load each)."""
    nk = 9 * nslice
    P = Producer()
    wld = 0
    wt = 0                          # next weight tile

    def wtile():
        nonlocal wld, wt
        P.issue('W', wld, wt, NWI)
        wld = (wld + 1) % NS
        wt += 1
    for s in range(NS - 1):
        if s < nk:
            wtile()
    for i in range(NXP):
        P.issue('X', 0, (0, i), 1)
    P.wait(0)
    P.barrier()
    if KP == 1:
        sl = tap = xprev = 0
        kt = 0
        while kt < nk - (NS - 1):
            xp = 1 if (sl + 1 < nslice and tap < NXP) else 0
            if xp:
                P.issue('X', (sl + 1) & 1, (sl + 1, tap), 1)
            wtile()
            xin = xp if NS == 3 else xp + xprev
            P.wait((NS - 2) * NWI + xin)
            P.barrier()
            xprev = xp
            tap += 1
            if tap == 9:
                tap = 0
                sl += 1
            kt += 1
        while kt < nk:
            P.wait(0)
            P.barrier()
            kt += 1
    else:
        NP = NS // 2
        npairs = (nk + 1) >> 1
        ps, pk, pnext = 0, 0, 5
        u = 0
        while 2 * (u + NP - 1) + 1 < nk:
            if u == pnext:
                ps += 1
                pk = 0
                pnext += 4 if (ps & 1) else 5
            if ps + 1 < nslice and pk < 4:
                for i in (2 * pk, 2 * pk + 1):
                    if i < NXP:
                        P.issue('X', (ps + 1) & 1, (ps + 1, i), 1)
                pk += 1
            while wt <= 2 * (u + NP - 1) + 1:
                wtile()
            P.wait((NP - 2) * 2 * NWI)
            P.barrier()
            u += 1
        while u < npairs:
            while wt < nk and wt <= 2 * (u + NP - 1) + 1:
                wtile()
            P.wait(0)
            P.barrier()
            u += 1
    assert not P.fifo or all('ok' not in l for l in P.fifo)
    return P


def consumer_reads(nk, NS, KP, halo_slices=0):
    """[(interval, kind, where, what)] and the number of barriers the consumer executes."""
    reads = []
    iv = 0                          # after barrier 0
    barriers = 1
    rd = 0
    sl = tap = 0
    for kt in range(nk):
        reads.append((iv, 'W' if halo_slices else 'T', rd, kt))
        if halo_slices:
            reads.append((iv, 'Xall', sl & 1, sl))
        if KP == 1 or (kt & 1) or kt == nk - 1:
            iv += 1
            barriers += 1
        rd = (rd + 1) % NS
        if halo_slices:
            tap += 1
            if tap == 9:
                tap = 0
                sl += 1
    return reads, barriers


def check(P, reads, cbarriers, NXP=0, tag=''):
    assert P.barriers == cbarriers, (tag, 'barrier counts', P.barriers, cbarriers)
    loads = P.done + P.fifo
    for (iv, kind, where, what) in reads:
        if kind == 'Xall':
            need = [l for l in loads if l['kind'] == 'X' and l['what'][0] == what]
            assert len(need) == NXP, (tag, 'patch pieces of slice', what, len(need))
            others = [l for l in loads if l['kind'] == 'X' and l['where'] == where and l['what'][0] != what]
        else:
            need = [l for l in loads if l['kind'] == kind and l['what'] == what]
            assert len(need) == 1 and need[0]['where'] == where, (tag, 'tile', what, 'expected in stage', where, need)
            others = [l for l in loads if l['kind'] == kind and l['where'] == where and l['what'] != what]
        for l in need:
            # guaranteed by a wait that completed in an interval < iv (the producer then passed barrier iv with it retired)
            assert 'ok' in l and l['ok'] < iv, (tag, 'read of', kind, what, 'in interval', iv, 'but load retired in', l.get('ok'))
        newest_need = max(l['issued'] for l in need)
        for l in others:
            later = l['issued'] > newest_need or (l['issued'] == newest_need and loads.index(l) > max(loads.index(x) for x in need))
            if later:
                # a younger load into the same place must not be issued while (or before) this read can still be in flight
                assert l['issued'] > iv, (tag, 'stage', kind, where, 'overwritten by', l['what'], 'issued in interval', l['issued'],
                                          'while', what, 'is read in interval', iv)
            else:
                # an older occupant: the place was reused for `what` -- nothing to check here (checked from the older read's side)
                pass


def main():
    n = 0
    for nk in range(1, 41):
        for NS, KP in ((3, 1), (4, 1), (6, 1), (4, 2), (6, 2)):
            P = stream_producer(nk, NS, KP)
            reads, cb = consumer_reads(nk, NS, KP)
            check(P, reads, cb, tag=f'stream nk={nk} NS={NS} KP={KP}')
            n += 1
    for nslice in range(1, 9):
        for NXP in range(1, 9):
            for NS, KP in ((3, 1), (4, 1), (4, 2)):
                if KP == 1 and NXP > 11 - NS:
                    continue
                P = halo_producer(nslice, NS, KP, NXP)
                reads, cb = consumer_reads(9 * nslice, NS, KP, halo_slices=nslice)
                check(P, reads, cb, NXP=NXP, tag=f'halo nslice={nslice} NS={NS} KP={KP} NXP={NXP}')
                n += 1
    print(n, 'configurations: barrier counts match, every read is of retired data, no stage is overwritten under a reader')


if __name__ == '__main__':
    main()
