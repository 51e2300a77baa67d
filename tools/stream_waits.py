"""Un-profiled, in-situ: how long does the caller's stream WAIT for the look-ahead lanes (window encoder + ahead affinity read-out) at the
start of a frame, and how long is a frame on that stream?  Timing events around InferenceCore.step's wait_event (inference_core.WAIT_TRACE)
and around whole frames.  (rocprofv3 --kernel-trace serialises the queues: its gaps say nothing about overlap.)
    python tools/stream_waits.py [--window 8] [--lead 2] [--frames 300]"""
import argparse, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
ap = argparse.ArgumentParser()
ap.add_argument('--window', type=int, default=8)
ap.add_argument('--lead', type=int, default=2)
ap.add_argument('--frames', type=int, default=300)
ap.add_argument('--no-affinity-ahead', action='store_true')
args = ap.parse_args()
from cutie_amd.config import default_config
from cutie_amd.inference import inference_core as IC
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict
IC.WINDOW, IC.WINDOW_LEAD = args.window, args.lead
if args.no_affinity_ahead:
    IC.AHEAD_AFFINITY = False
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval(); net.load_weights(make_state_dict(0))
clip = SyntheticClip(480, 854, 3, 128, seed=1)
frames = torch.stack([clip.frame(t) for t in range(128)]).cuda()
mask = clip.first_mask().cuda()
proc = IC.InferenceCore(net, cfg=cfg)
hint = lambda t: {'next_images': [frames[(t + 1 + j) % 128] for j in range(args.window + 4)]} if args.window > 1 else {'next_image': frames[(t + 1) % 128]}
with torch.inference_mode(), torch.cuda.stream(torch.cuda.Stream()):
    proc.step(frames[0], mask, objects=clip.objects, **hint(0))
    for t in range(1, 300):
        proc.step(frames[t % 128], **hint(t))
    torch.cuda.synchronize()
    IC.WAIT_TRACE = []
    marks = []
    import time
    t0 = time.perf_counter()
    for t in range(300, 300 + args.frames):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
        proc.step(frames[t % 128], **hint(t))
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(e)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
frame_ms = np.array([a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])])
waits = {ti: a.elapsed_time(b) for ti, a, b in IC.WAIT_TRACE}
w = np.array(list(waits.values()))
print(f'window {args.window} lead {args.lead} affinity ahead {IC.AHEAD_AFFINITY}: {args.frames} frames, wall {wall / args.frames * 1e3:.3f} ms per frame '
      f'({args.frames / wall:.1f} fps), host issue {host / args.frames * 1e3:.3f} ms per frame')
print(f'  frame on the caller\'s stream (event to event): mean {frame_ms.mean():.3f} ms, median {np.median(frame_ms):.3f}, p90 {np.percentile(frame_ms, 90):.3f}')
print(f'  wait for the look-ahead at the start of a frame: mean {w.mean() * 1e3:.1f} us, median {np.median(w) * 1e3:.1f}, p90 {np.percentile(w, 90) * 1e3:.1f}, max {w.max() * 1e3:.1f} '
      f'({len(w)} waits; an event pair with nothing to wait for measures ~{np.percentile(w, 5) * 1e3:.1f} us)')
k = np.arange(len(frame_ms)) % 5
print('  frame ms by position in the memory cycle (mod 5):', [round(float(frame_ms[k == i].mean()), 3) for i in range(5)])
