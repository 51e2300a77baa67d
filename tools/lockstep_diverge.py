"""Where does a lock-step clip leave its own InferenceCore run?  C clips advanced frame by frame both ways (no hints), the per-clip state compared
after every frame: probabilities, sensory state (fp32 master), object summaries, bank sizes / keys / values / usage.  Stops at the first difference.
    python tools/lockstep_diverge.py [--frames 200] [--clips 4]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.inference.lockstep import LockstepCores
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict

ap = argparse.ArgumentParser()
ap.add_argument('--clips', type=int, default=4)
ap.add_argument('--frames', type=int, default=200)
ap.add_argument('--objects', type=int, default=3)
ap.add_argument('--height', type=int, default=480)
ap.add_argument('--width', type=int, default=854)
ap.add_argument('--no-long-term', action='store_true')
a = ap.parse_args()
cfg = default_config(use_long_term=not a.no_long_term)
net = CUTIE(cfg).cuda().eval()
net.load_weights(make_state_dict(seed=0))
NF = 48
clips = [SyntheticClip(a.height, a.width, a.objects, NF, seed=300 + c) for c in range(a.clips)]
frames = [[cl.frame(t).cuda() for t in range(NF)] for cl in clips]
fr = lambda c, t: frames[c][t % NF] if (t // NF) % 2 == 0 else frames[c][NF - 1 - t % NF]
C, T = a.clips, a.frames


def state(core, prob):
    mm = core.memory
    b = next(iter(mm.buckets.values()))
    n = b.slots
    d = dict(prob=prob, sensory_f32=mm._sens_f32, sensory_bf16=mm._sens_bf16.view(torch.int16), objv=mm._objv, last_mask=core.last_mask,
             Ahi=b.Ahi[:n].view(torch.int16), Alo=b.Alo[:n].view(torch.int16), scale=b.scale[:n])
    if b.lt:
        d.update(use=b.use[:n], life=b.life[:n])
    for k, o in enumerate(b.objects):
        d['value%d' % k] = b.values[o][:n].view(torch.int16)
    meta = (b.n_long, b.n_perm, b.n_work, b.perm_start, b.work_start, n)
    return {k: v.clone() for k, v in d.items()}, meta


# every launch plan's bound tensors of the current frame, cloned behind its run (which stage of the frame differs first?)
from cutie_amd.model import plans as _plans
LOG = []
_run = _plans.Plan.run


def _logged_run(self, **dyn):
    _run(self, **dyn)
    torch.cuda.synchronize()
    LOG.append({k: v.clone() for k, v in dyn.items() if isinstance(v, torch.Tensor)})


_plans.Plan.run = _logged_run


def stage_of(d):
    ks = set(d)
    if 'prob' in ks: return 'segment'
    if 'obj_mem' in ks: return 'readout_query'
    if 'fused' in ks: return 'pixel_fusion'
    if 'f16' in ks: return 'encode'
    return 'other(' + ','.join(sorted(ks))[:60] + ')'


def inside(c, seq_logs, ls_logs, K):
    sq = {stage_of(d): d for d in seq_logs}
    lq = {stage_of(d): d for d in ls_logs}
    def cmp(name, x, y):
        if x.shape != y.shape:
            print('      %-28s shapes %s / %s' % (name, tuple(x.shape), tuple(y.shape))); return
        same = torch.equal(x, y)
        xd, yd = x.double(), y.double()
        print('      %-28s %s' % (name, 'identical' if same else '%d of %d elements differ, max |d| %.3g (max |x| %.3g)' % (int((x != y).sum()), x.numel(), float((xd - yd).abs().max()), float(xd.abs().max()))))
    for st in ('encode', 'pixel_fusion', 'readout_query', 'segment'):
        if st not in sq or st not in lq:
            print('   stage %s: not recorded on both sides (%s / %s)' % (st, st in sq, st in lq)); continue
        a, b = sq[st], lq[st]
        print('   stage', st)
        for k in sorted(a):
            if k.startswith('image') or k not in b and not (k == 'pixel' and ('pixel%d' % c) in b) and not (k == 'last_mask' and ('last_mask%d' % c) in b):
                continue
            x = a[k]
            if k == 'pixel' and st == 'pixel_fusion' and ('pixel%d' % c) in b:
                y = b['pixel%d' % c]
            elif k == 'last_mask' and ('last_mask%d' % c) in b:
                y = b['last_mask%d' % c]
            else:
                y = b[k]
                if st == 'encode':
                    y = y[c:c + 1] if y.dim() == x.dim() and y.shape[0] != x.shape[0] else (y[c] if y.dim() == x.dim() + 1 else y)
                    if y.shape != x.shape and y.numel() >= x.numel():
                        y = y.reshape(-1)[:x.numel()].view(x.shape) if k in ('Bhi', 'Blo', 'cq') else y
                elif y.shape[0] == C * x.shape[0]:
                    y = y[c * x.shape[0]:(c + 1) * x.shape[0]]
                elif st == 'segment' and k == 'prob':
                    y = y[c]
                elif k in ('fuse_xt', 'f8p', 'f4p', 'pix_feat'):
                    continue                                    # (the lock-step plan binds clip 0's slice of the window output: compared in `encode`)
            cmp(k, x, y)


with torch.inference_mode():
    procs = [InferenceCore(net, cfg=cfg) for _ in range(C)]
    ls = LockstepCores(net, cfg, C)
    for t in range(T):
        if t == 0:
            ps = [p.step(fr(c, 0), clips[c].first_mask().cuda(), objects=clips[c].objects) for c, p in enumerate(procs)]
            pl = ls.step([fr(c, 0) for c in range(C)], [cl.first_mask().cuda() for cl in clips], [cl.objects for cl in clips])
        else:
            seq_logs = []
            for c, p in enumerate(procs):
                del LOG[:]
                if c == 0:
                    ps = []
                ps.append(p.step(fr(c, t)))
                seq_logs.append(list(LOG))
            del LOG[:]
            pl = ls.step([fr(c, t) for c in range(C)])
            ls_logs = list(LOG)
        torch.cuda.synchronize()
        found = False
        for c in range(C):
            s1, m1 = state(procs[c], ps[c])
            s2, m2 = state(ls[c], pl[c])
            diffs = []
            if m1 != m2:
                diffs.append('bank layout %s against %s' % (m2, m1))
            for k in s1:
                if s1[k].shape != s2[k].shape:
                    diffs.append('%s shape' % k)
                elif not torch.equal(s1[k], s2[k]):
                    x, y = s1[k].double(), s2[k].double()
                    nd = int((s1[k] != s2[k]).sum())
                    diffs.append('%s: %d of %d elements, max |d| %.3g (max |x| %.3g)' % (k, nd, x.numel(), float((x - y).abs().max()), float(x.abs().max())))
            if diffs:
                found = True
                is_mem = (procs[c].last_mem_ti == procs[c].curr_ti)
                print('frame %d clip %d (memory frame: %s, batched steps so far %d): ' % (t, c, is_mem, ls.batched_steps) + '; '.join(diffs))
                if t > 0:
                    inside(c, seq_logs[c], ls_logs, a.objects)
        if found:
            break
    else:
        print('no difference in %d frames' % T)
