"""Isolated A/B of small launches (MI355X box): KEY_PREP query side with c_j handed on between the row's lanes (default) against the
one-lane-per-row loop (flags&2), GRU four channels per thread against one (flags&1), and a COPY2D timed right behind each of them (the
bench breakdown of call 29 showed COPY2D at 2x behind the new KEY_PREP: an artefact of the replay order or real?).
hipEvents around `iters` back-to-back replays (cutie_time_ops); min and median of 7 rounds, us per launch."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
ex = _lib.get_executor()
dev = 'cuda'
g = torch.Generator().manual_seed(0)
HW, HWp = 1620, 1664
qkey = (torch.randn((12, HW, 64), generator=g) * 0.8).to(dev); qsel = torch.rand((12, HW, 64), generator=g).to(dev)
Bhi = torch.zeros((12, HWp, 128), dtype=torch.bfloat16, device=dev); Blo = torch.zeros_like(Bhi); cq = torch.zeros((12, HWp), device=dev)
n, C = 4860, 256
v = (torch.randn((n, 3 * C), generator=g) * 2).to(dev); h = torch.randn((n, C), generator=g).to(dev); hb = torch.zeros((n, C), dtype=torch.bfloat16, device=dev)
src = torch.randn((1620, 256), generator=g).to(dev); dst = torch.zeros((4000, 256), device=dev)


def oplist(fn):
    ol = O.OpList(); fn(ol); return ol.finalize().copy(), ol


def kp(loop, nb):
    def f(ol):
        O.KEYPREP_LOOP = loop
        for b in range(nb):
            ol.key_prep(qkey[b], qsel[b], Bhi[b], Blo[b], cq[b], n=HW, query=True)
    return f


def gru(scalar):
    def f(ol):
        O.GRU_SCALAR = scalar
        ol.gru(v, h, hb, n=n, C=C)
    return f


keep = []
cases = {}
for name, fn, per in (('key_prep lanes x1', kp(0, 1), 1), ('key_prep loop  x1', kp(2, 1), 1), ('key_prep lanes x12', kp(0, 12), 12), ('key_prep loop  x12', kp(2, 12), 12),
                      ('gru x4', gru(0), 1), ('gru x1', gru(1), 1),
                      ('copy2d', lambda ol: ol.copy2d(src, dst, rows=1620, rowbytes=1024, src_stride=1024, dst_stride=1024), 1)):
    arr, ol = oplist(fn); keep.append(ol); cases[name] = (arr, per)
t = lambda a, it=50: ex.time_ops(a, it) * 1e3
for rnd in range(2):
    for name in cases:
        arr, per = cases[name]
        ts = sorted(t(arr) / per for _ in range(7))
        tc = sorted(t(cases['copy2d'][0]) for _ in range(3))
        print('%-20s min %.2f median %.2f us per launch | copy2d right behind it: %.2f' % (name, ts[0], ts[3], tc[0]))
