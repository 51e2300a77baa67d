"""A/B timing of the affinity plan stages [memset, score/0, select, score/1, readout] for 16 vs 32 queries per wave.
Run on the MI355X box: python tools/aff_ab.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import Recorder
from cutie_amd import _lib, ops as O
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval(); net.load_weights(make_state_dict(0))
rec = Recorder(_lib.get_executor()); _lib.set_executor_for_testing(rec)
clip = SyntheticClip(480, 854, 3, 64, seed=1)
proc = InferenceCore(net, cfg=cfg)
with torch.inference_mode(), torch.cuda.stream(torch.cuda.Stream()):
    proc.step(clip.frame(0).cuda(), clip.first_mask().cuda(), objects=clip.objects)
    for t in range(1, 300): proc.step(clip.frame(t % 64).cuda())
    torch.cuda.synchronize()
    rec.rec, rec.on = [], True
    proc.step(clip.frame(5).cuda()); rec.on = False
    torch.cuda.synchronize()
    affs = np.concatenate([a[a['kind'] != O.USAGE_TICK] for a in rec.rec if (a['kind'] == O.AFF_SCORE).any()])
    for tpb_hint in (0,):
        for nq in (1, 2):
            a = affs.copy(); a['i'][1, 12] = nq; a['i'][3, 12] = nq
            pre = [min(rec.ex.time_ops(a[:k], 10) for _ in range(3)) * 1e3 for k in (1, 2, 3, 4, 5)]
            print('nq', nq, 'tokens', sum(b.size() for b in proc.memory.buckets.values()), [round(pre[0],1)] + [round(pre[k] - pre[k-1], 1) for k in range(1, 5)])
    rec.ex.run(affs); torch.cuda.synchronize()
    for k, v in proc.memory._scratch.items():
        if 'count' in str(k):
            c = v.float()
            print('candidates per query: mean %.1f  median %.1f  p99 %.1f  max %d' % (c.mean(), c.median(), c.quantile(0.99), int(c.max())))
