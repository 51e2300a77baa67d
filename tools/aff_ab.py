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

    for nq in (2, 12, 4, 1):
        for tpb in ((0,) if nq not in (4, 12) else (0, 8, 12, 16, 24)):
            a = affs.copy(); a['i'][0, 12] = nq % 10; a['i'][2, 12] = nq % 10; a['i'][0, 13] = tpb; a['i'][2, 13] = tpb
            a['i'][0, 15] = a['i'][2, 15] = int(nq >= 10)
            s = stages(a)
            print('nq %d tpb %2d: score0 %.2f (mfma util %.3f)  select %.2f  score1 %.2f  readout %.2f  | plan %.1f us' %
                  (nq, tpb, s[0], issued / (s[0] * 1e-6) / 1e12 / PEAK, s[1], s[2], s[3], s[4]))
    # mixed: score0 on the 64-query kernel, score1 on the 32-query one and vice versa
    for n0, n1 in ((4, 2), (2, 4)):
        a = affs.copy(); a['i'][0, 12] = n0; a['i'][2, 12] = n1
        s = stages(a)
        print('score0 nq %d, score1 nq %d: score0 %.2f  score1 %.2f | plan %.1f us' % (n0, n1, s[0], s[2], s[4]))
    # what bounds the read-out?  without the usage atomics (p4 = 0), with one object instead of three (a third of the gather bytes),
    # with top_k = 8 (a quarter of the rows, one gather round instead of two)
    a = affs.copy(); base = t1(a[3:4])
    a = affs.copy(); a['p'][3, 4] = 0; no_usage = t1(a[3:4])
    a = affs.copy(); a['i'][3, 3] = 1; one_obj = t1(a[3:4])
    a = affs.copy(); a['i'][3, 2] = 8; k8 = t1(a[3:4])
    a = affs.copy(); a['i'][3, 2] = 8; a['p'][3, 4] = 0; a['i'][3, 3] = 1; mini = t1(a[3:4])
    # the heavy tail of the candidate lists (1 % of the queries hold ~260 candidates): the same launch with every list cut to 40 / 32
    rec.ex.run(affs); torch.cuda.synchronize()
    cnt_t = proc.memory._scratch['count']
    for cut in (64, 40, 32):
        c2 = cnt_t.clone().clamp_(max=cut)
        a = affs.copy(); a['p'][3, 2] = c2.data_ptr()
        print('readout with the candidate lists cut to <= %d: %.1f us' % (cut, t1(a[3:4])))
    print('readout alone %.1f us | no usage atomics %.1f | K = 1 %.1f | top_k = 8 %.1f | K = 1, top_k = 8, no usage %.1f' % (base, no_usage, one_obj, k8, mini))
    rec.ex.run(affs); torch.cuda.synchronize()
    # overlap of the selected tokens between neighbouring queries: unique tokens among the top-30 of 16 consecutive queries (of 480)
    sc = proc.memory._scratch
    cv, ci, cn = sc['cand_val'].float().cpu(), sc['cand_idx'].cpu(), sc['count'].view(-1, 32)[:, 0].cpu()
    uni, uni4 = [], []
    sets = []
    for j in range(cv.shape[0]):
        n = int(cn[j])
        v, i_ = cv[j, :n], ci[j, :n]
        top = i_[torch.argsort(v, descending=True)[:30]]
        sets.append(set(int(x) for x in top))
    for j0 in range(0, len(sets) - 15, 16):
        uni.append(len(set().union(*sets[j0:j0 + 16])))
    W = proc.memory.W
    for y in range(0, proc.memory.H - 3, 4):
        for x in range(0, W - 3, 4):
            uni4.append(len(set().union(*[sets[(y + dy) * W + x + dx] for dy in range(4) for dx in range(4)])))
    print('unique selected tokens per 16 consecutive queries: mean %.1f max %d (of 480); per 4 x 4 pixel block: mean %.1f max %d' %
          (sum(uni) / len(uni), max(uni), sum(uni4) / len(uni4), max(uni4)))
    for k, v in proc.memory._scratch.items():
        if 'count' in str(k):
            c = v.float().view(-1, 32)[:, 0]
            print(k, 'candidates per query: mean %.1f  median %.1f  p99 %.1f  max %d' % (c.mean(), c.median(), c.quantile(0.99), int(c.max())))
