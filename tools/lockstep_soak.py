"""Soak test on the MI355X: C clips x T frames (480p, K objects, long-term memory with the default settings: consolidations and prunings at the
bench's cadence), every frame's probabilities compared bit for bit (a checksum per frame + the object-id map) between
  seq     every clip's own InferenceCore, step(image)                         (the reference run)
  seq2    the same again                                                      (run-to-run determinism)
  seqh    every clip's own InferenceCore, step(image, next_images=...)        (look-ahead lanes of one clip)
  ls      LockstepCores with hints (joint encoder window, joint read-out pass)
  lsn     LockstepCores without hints
  lsl     LockstepCores with hints, every clip's own read-out lane (JOINT off)
  il      the clips in flight next to each other (parallel.run_interleaved: a stream + CUTIE.fork() per clip, one issuing thread), hinted
  grp     parallel.run_batched: lock-step groups of two clips in flight next to each other
Stream-ordering mistakes between the look-ahead lanes show up rarely, if at all, in the short suite cases.
    python tools/lockstep_soak.py [--clips 4] [--frames 300] [--objects 3] [--modes seq2,seqh,ls,lsn,lsl]"""
import argparse, os, sys, time
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
ap.add_argument('--frames', type=int, default=300)
ap.add_argument('--objects', type=int, default=3)
ap.add_argument('--height', type=int, default=480)
ap.add_argument('--width', type=int, default=854)
ap.add_argument('--modes', default='seq2,seqh,ls,lsn,lsl')
ap.add_argument('--no-long-term', action='store_true')
ap.add_argument('--model', default='base', choices=['base', 'small'])
a = ap.parse_args()
cfg = default_config(use_long_term=not a.no_long_term, **({'model': 'small'} if a.model == 'small' else {}))
net = CUTIE(cfg).cuda().eval()
if a.model == 'small':
    from cutie_amd.utils.synth_weights import MODEL_CFG_SMALL
    net.load_weights(make_state_dict(seed=0, m=MODEL_CFG_SMALL))
else:
    net.load_weights(make_state_dict(seed=0))
NF = 48
clips = [SyntheticClip(a.height, a.width, a.objects, NF, seed=300 + c) for c in range(a.clips)]
frames = [[cl.frame(t).cuda() for t in range(NF)] for cl in clips]
fr = lambda c, t: frames[c][t % NF] if (t // NF) % 2 == 0 else frames[c][NF - 1 - t % NF]      # (forth and back: no jump at the wrap)
T, C = a.frames, a.clips
chk = lambda p: (p.double() * torch.arange(1, p.shape[0] + 1, device=p.device, dtype=torch.float64).view(-1, 1, 1)).sum()      # stays on the device


def bank(mm):
    b = next(iter(mm.buckets.values()))
    return (b.n_long, b.n_perm, b.n_work, float(b.use[:b.slots].double().sum()) if b.lt else 0.0)


def run_seq(hinted):
    res = []
    for c, cl in enumerate(clips):
        proc = InferenceCore(net, cfg=cfg)
        sums = [chk(proc.step(fr(c, 0), cl.first_mask().cuda(), objects=cl.objects))]
        for t in range(1, T):
            hint = dict(next_images=[fr(c, u) for u in range(t + 1, min(T, t + 13))]) if hinted and t + 1 < T else {}
            sums.append(chk(proc.step(fr(c, t), **hint)))
        res.append((torch.stack(sums).cpu(), bank(proc.memory)))
    return res


def run_ls(hinted, joint):
    LockstepCores.JOINT = joint
    ls = LockstepCores(net, cfg, C)
    sums = [[] for _ in clips]
    for c, p in enumerate(ls.step([fr(c, 0) for c in range(C)], [cl.first_mask().cuda() for cl in clips], [cl.objects for cl in clips])):
        sums[c].append(chk(p))
    for t in range(1, T):
        hint = dict(next_images=[[fr(c, u) for u in range(t + 1, min(T, t + 13))] for c in range(C)]) if hinted and t + 1 < T else {}
        for c, p in enumerate(ls.step([fr(c, t) for c in range(C)], **hint)):
            sums[c].append(chk(p))
    print('   (batched steps %d, joint passes %d, stacked steps %d)' % (ls.batched_steps, ls.joint_passes, ls.stacked_steps))
    return [(torch.stack(sums[c]).cpu(), bank(ls[c].memory)) for c in range(C)]


def run_il():
    from cutie_amd.parallel import run_interleaved

    def gen(view, c):
        proc = InferenceCore(view, cfg=cfg)
        sums = [chk(proc.step(fr(c, 0), clips[c].first_mask().cuda(), objects=clips[c].objects))]
        yield
        for t in range(1, T):
            hint = dict(next_images=[fr(c, u) for u in range(t + 1, min(T, t + 13))]) if t + 1 < T else {}
            sums.append(chk(proc.step(fr(c, t), **hint)))
            yield
        return (torch.stack(sums).cpu(), bank(proc.memory))
    got = run_interleaved(net, list(range(C)), gen, streams=C)
    return [got[c] for c in range(C)]


def run_grp():
    from cutie_amd.parallel import run_batched
    sums = [[None] * T for _ in clips]

    def on_frame(c, t, prob, core):
        sums[c][t] = chk(prob)
    run_batched(net, cfg, [dict(frames=[fr(c, t) for t in range(T)], mask=clips[c].first_mask().cuda(), objects=clips[c].objects) for c in range(C)],
                lockstep=2, in_flight=max(1, C // 2), on_frame=on_frame)
    return [(torch.stack(sums[c]).cpu(), None) for c in range(C)]


bad = 0
with torch.inference_mode():
    t0 = time.time()
    ref = run_seq(False)
    torch.cuda.synchronize()
    print('seq: %.1f s; bank of clip 0 (long, perm, work, sum of usage) = %s' % (time.time() - t0, ref[0][1]))
    for mode in a.modes.split(','):
        t0 = time.time()
        got = {'seq2': lambda: run_seq(False), 'seqh': lambda: run_seq(True), 'ls': lambda: run_ls(True, True), 'lsn': lambda: run_ls(False, True),
               'lsl': lambda: run_ls(True, False), 'il': run_il, 'grp': run_grp}[mode]()
        torch.cuda.synchronize()
        n = 0
        for c in range(C):
            same = torch.equal(got[c][0], ref[c][0])
            if not same or (got[c][1] is not None and got[c][1][:3] != ref[c][1][:3]):
                n += 1
                d = (got[c][0] != ref[c][0]).nonzero().flatten()
                first = int(d[0]) if len(d) else None
                rel = float(((got[c][0] - ref[c][0]).abs() / ref[c][0].abs()).max())
                print('   MISMATCH %s clip %d: first differing frame %s of %d differing, max relative checksum difference %.3g, bank %s against %s' % (
                    mode, c, first, len(d), rel, got[c][1], ref[c][1]))
        bad += n
        print('%s: %.1f s, %d of %d clips differ from seq' % (mode, time.time() - t0, n, C))
print('SOAK', 'FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
