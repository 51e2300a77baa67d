"""Isolated timing of the affinity plan stages [memset, score/0, select, score/1, readout] and of variants (pass-1 tile skipping
on / off, tiles per block).  Run on the MI355X box: python tools/aff_ab.py [preroll frames]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import Recorder
from cutie_amd import _lib, ops as O
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval(); net.load_weights(make_state_dict(0))
rec = Recorder(_lib.get_executor()); _lib.set_executor_for_testing(rec)
clip = SyntheticClip(480, 854, 3, 64, seed=1)
proc = InferenceCore(net, cfg=cfg)
t1 = lambda a, it=10: min(rec.ex.time_ops(a, it) for _ in range(3)) * 1e3
with torch.inference_mode(), torch.cuda.stream(torch.cuda.Stream()):
    proc.step(clip.frame(0).cuda(), clip.first_mask().cuda(), objects=clip.objects)
    for t in range(1, pre): proc.step(clip.frame(t % 64).cuda())
    torch.cuda.synchronize()
    rec.rec, rec.on = [], True
    proc.step(clip.frame(5).cuda()); rec.on = False
    torch.cuda.synchronize()
    affs = np.concatenate([a[a['kind'] != O.USAGE_TICK] for a in rec.rec if (a['kind'] == O.AFF_SCORE).any()])
    assert [int(k) for k in affs['kind']] == [O.MEMSET32, O.AFF_SCORE, O.AFF_SELECT, O.AFF_SCORE, O.AFF_READOUT], affs['kind']
    print('tokens', sum(b.size() for b in proc.memory.buckets.values()), 'whole plan', round(t1(affs), 1), 'us')
    print('alone: memset %.1f  score0 %.1f  select %.1f  memset+score1 %.1f  readout %.1f' %
          (t1(affs[0:1]), t1(affs[1:2]), t1(affs[2:3]), t1(affs[[0, 3]]), t1(affs[4:5])))
    a = affs.copy(); a['flags'][3] = 0
    print('memset+score1 without tile skipping %.1f' % t1(a[[0, 3]]))
    for tpb in (8, 12, 20, 32, 48, 80):
        a = affs.copy(); a['i'][1, 13] = tpb; a['i'][3, 13] = tpb
        print('tiles per block', tpb, ': score0 %.1f  memset+score1 %.1f' % (t1(a[1:2]), t1(a[[0, 3]])))
    for nq in (1, 2):
        a = affs.copy(); a['i'][1, 12] = nq; a['i'][3, 12] = nq
        print('queries per wave', 16 * nq, ': score0 %.1f  memset+score1 %.1f' % (t1(a[1:2]), t1(a[[0, 3]])))
    rec.ex.run(affs); torch.cuda.synchronize()
    for k, v in proc.memory._scratch.items():
        if 'count' in str(k):
            c = v.float().view(-1, 32)[:, 0]
            print('candidates per query: mean %.1f  median %.1f  p99 %.1f  max %d' % (c.mean(), c.median(), c.quantile(0.99), int(c.max())))
