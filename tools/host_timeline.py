"""The reference's FPS protocol (cutie/eval_vos.py:126-145: synchronize, event, step, event, synchronize) frame by frame: where does the
host stand while the device works?  Every cutie_exec call of a frame is stamped (host clock, relative to the entry of `step`), so the
Python between two plans and the issue time of each plan can be read off; the frame's event interval says what the protocol charges.
    python tools/host_timeline.py [--frames 200] [--hints] [--profile]"""
import argparse, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=200)
ap.add_argument('--hints', action='store_true', help='step(next_images=...) instead of step(image)')
ap.add_argument('--profile', action='store_true', help='cProfile over the timed loop (slows it down: read the shares, not the times)')
args = ap.parse_args()
from cutie_amd import _lib
from cutie_amd.config import default_config
from cutie_amd.inference import inference_core as IC
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval(); net.load_weights(make_state_dict(0))
clip = SyntheticClip(480, 854, 3, 128, seed=1)
frames = torch.stack([clip.frame(t) for t in range(128)]).cuda()
mask = clip.first_mask().cuda()
proc = IC.InferenceCore(net, cfg=cfg)
hint = (lambda t: {'next_images': [frames[(t + 1 + j) % 128] for j in range(IC.WINDOW + 4)]}) if args.hints else (lambda t: {})

stamps = None
ex = _lib.get_executor()
run0 = ex.run


def run(arr):
    if stamps is None:
        return run0(arr)
    a = time.perf_counter()
    run0(arr)
    stamps.append((a, time.perf_counter(), len(arr), int(arr['kind'][0])))


ex.run = run
with torch.inference_mode(), torch.cuda.stream(torch.cuda.Stream()):
    proc.step(frames[0], mask, objects=clip.objects, **hint(0))
    for t in range(1, 100):
        proc.step(frames[t % 128], **hint(t))
    recs = []
    prof = None
    if args.profile:
        import cProfile
        prof = cProfile.Profile()
    for t in range(100, 100 + args.frames):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stamps = []
        t0 = time.perf_counter()
        e0.record()
        if prof:
            prof.enable()
        proc.step(frames[t % 128], **hint(t))
        if prof:
            prof.disable()
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        recs.append((t0, t1, t2, e0.elapsed_time(e1), stamps, proc.curr_ti == proc.last_mem_ti))
        stamps = None
ev = np.array([r[3] for r in recs])
print(f'{args.frames} frames, hints {args.hints}: event interval mean {ev.mean():.3f} ms ({1e3 / ev.mean():.1f} fps by the reference protocol), '
      f'host in step {np.mean([r[1] - r[0] for r in recs]) * 1e3:.3f} ms, entry -> device idle {np.mean([r[2] - r[0] for r in recs]) * 1e3:.3f} ms')
for mem in (False, True):
    sel = [r for r in recs if r[5] == mem]
    if not sel:
        continue
    n = int(np.median([len(r[4]) for r in sel]))
    sel = [r for r in sel if len(r[4]) == n]
    print(f'--- {"memory" if mem else "plain"} frames ({len(sel)} with {n} cutie_exec calls): event interval {np.mean([r[3] for r in sel]):.3f} ms, '
          f'host in step {np.mean([r[1] - r[0] for r in sel]) * 1e3:.3f} ms')
    print('   call   ops  first kind |  python before (us)   issue (us)   starts at (us after entry)')
    prev = np.array([r[0] for r in sel])
    tot_py = tot_issue = 0.0
    for j in range(n):
        a = np.array([r[4][j][0] for r in sel]); b = np.array([r[4][j][1] for r in sel])
        py, iss = (a - prev).mean() * 1e6, (b - a).mean() * 1e6
        tot_py += py; tot_issue += iss
        print(f'   {j:4d}  {sel[0][4][j][2]:4d}  {sel[0][4][j][3]:10d} |  {py:10.1f}  {iss:14.1f}  {(a - np.array([r[0] for r in sel])).mean() * 1e6:12.1f}')
        prev = b
    tail = (np.array([r[1] for r in sel]) - prev).mean() * 1e6
    print(f'   python {tot_py:.0f} us + issue {tot_issue:.0f} us + tail {tail:.0f} us')
if prof:
    import pstats
    pstats.Stats(prof).sort_stats('cumulative').print_stats(45)
