"""Isolated timing of the affinity plan [score/0, select, score/1, readout] at the bench's working point and of variants: query column
sets per wave (i[12]: 1, 2 = aff_score_kernel; 4 = aff_score4_kernel), tiles per block (i[13]).  Stage times by prefix differences
of back-to-back replays (hipEvents on the launch stream).  Run on the MI355X box: python tools/aff_ab.py [preroll frames]"""
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
t1 = lambda a, it=20: min(rec.ex.time_ops(a, it) for _ in range(4)) * 1e3
PEAK = 2500.0
with torch.inference_mode(), torch.cuda.stream(torch.cuda.Stream()):
    proc.step(clip.frame(0).cuda(), clip.first_mask().cuda(), objects=clip.objects)
    for t in range(1, pre): proc.step(clip.frame(t % 64).cuda())
    torch.cuda.synchronize()
    rec.rec, rec.on = [], True
    proc.step(clip.frame(5).cuda()); rec.on = False
    torch.cuda.synchronize()
    affs = np.concatenate([a[a['kind'] != O.USAGE_TICK] for a in rec.rec if (a['kind'] == O.AFF_SCORE).any()])
    kinds = [int(k) for k in affs['kind']]
    assert kinds == [O.AFF_SCORE, O.AFF_SELECT, O.AFF_SCORE, O.AFF_READOUT], kinds
    ii = affs['i'][0]
    issued = 3 * 2.0 * 128 * (int(ii[9]) * 16) * int(ii[1])
    print('tokens', sum(b.size() for b in proc.memory.buckets.values()), 'tiles', int(ii[9]), 'HWp', int(ii[1]), 'whole plan %.1f us' % t1(affs))

    def stages(a):
        s0, s01, s012, s0123 = t1(a[0:1]), t1(a[0:2]), t1(a[0:3]), t1(a)
        return s0, s01 - s0, s012 - s01, s0123 - s012, s0123

    for nq in (2, 4, 1):
        for tpb in ((0,) if nq != 4 else (0, 8, 12, 16, 24)):
            a = affs.copy(); a['i'][0, 12] = nq; a['i'][2, 12] = nq; a['i'][0, 13] = tpb; a['i'][2, 13] = tpb
            s = stages(a)
            print('nq %d tpb %2d: score0 %.2f (mfma util %.3f)  select %.2f  score1 %.2f  readout %.2f  | plan %.1f us' %
                  (nq, tpb, s[0], issued / (s[0] * 1e-6) / 1e12 / PEAK, s[1], s[2], s[3], s[4]))
    # mixed: score0 on the 64-query kernel, score1 on the 32-query one and vice versa
    for n0, n1 in ((4, 2), (2, 4)):
        a = affs.copy(); a['i'][0, 12] = n0; a['i'][2, 12] = n1
        s = stages(a)
        print('score0 nq %d, score1 nq %d: score0 %.2f  score1 %.2f | plan %.1f us' % (n0, n1, s[0], s[2], s[4]))
    rec.ex.run(affs); torch.cuda.synchronize()
    for k, v in proc.memory._scratch.items():
        if 'count' in str(k):
            c = v.float().view(-1, 32)[:, 0]
            print(k, 'candidates per query: mean %.1f  median %.1f  p99 %.1f  max %d' % (c.mean(), c.median(), c.quantile(0.99), int(c.max())))
