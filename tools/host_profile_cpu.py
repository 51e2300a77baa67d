"""Host-side cost of `InferenceCore.step` WITHOUT a device: the descriptors go to an executor that does nothing, so what is timed is
the Python between and around the cutie_exec calls (plan look-up, slot pool, pointer binding, bank bookkeeping).  CPU tensors are
not HIP tensors (no stream queries, cheaper allocator), so read the shares and the order of magnitude; `tools/host_timeline.py` has
the numbers of the GPU box.
    python tools/host_profile_cpu.py [--frames 200] [--hints] [--profile] [--small]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=200)
ap.add_argument('--hints', action='store_true')
ap.add_argument('--profile', action='store_true')
ap.add_argument('--small', action='store_true', help='240p frames (less memory while the plans are built)')
args = ap.parse_args()
import numpy as np, torch
from cutie_amd import _lib
from cutie_amd.config import default_config
from cutie_amd.inference import inference_core as IC
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict
from mock_exec import MockExecutor


class NullExecutor(MockExecutor):
    calls = None

    def run(self, arr):
        if self.calls is not None:
            self.calls.append((time.perf_counter(), len(arr)))


ex = NullExecutor()
_lib.set_executor_for_testing(ex)
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).eval(); net.load_weights(make_state_dict(0))
H, W = (240, 432) if args.small else (480, 854)
clip = SyntheticClip(H, W, 3, 32, seed=1)
frames = torch.stack([clip.frame(t) for t in range(32)])
mask = clip.first_mask()
proc = IC.InferenceCore(net, cfg=cfg)
hint = (lambda t: {'next_images': [frames[(t + 1 + j) % 32] for j in range(IC.WINDOW + 4)]}) if args.hints else (lambda t: {})
with torch.inference_mode():
    proc.step(frames[0], mask, objects=clip.objects, **hint(0))
    for t in range(1, 60):
        proc.step(frames[t % 32], **hint(t))
    prof = None
    if args.profile:
        import cProfile
        prof = cProfile.Profile()
    rec = []
    for t in range(60, 60 + args.frames):
        ex.calls = []
        t0 = time.perf_counter()
        if prof:
            prof.enable()
        proc.step(frames[t % 32], **hint(t))
        if prof:
            prof.disable()
        rec.append((t0, time.perf_counter(), ex.calls, proc.curr_ti == proc.last_mem_ti))
for mem in (False, True):
    sel = [r for r in rec if r[3] == mem]
    if not sel:
        continue
    n = int(np.median([len(r[2]) for r in sel]))
    sel = [r for r in sel if len(r[2]) == n]
    print(f'{"memory" if mem else "plain"} frames ({len(sel)}, {n} exec calls): host in step {np.mean([r[1] - r[0] for r in sel]) * 1e6:.1f} us; '
          'python before each call: ' + ' '.join(f'{np.mean([(r[2][j][0] - (r[2][j - 1][0] if j else r[0])) for r in sel]) * 1e6:.1f}' for j in range(n))
          + f' tail {np.mean([r[1] - r[2][-1][0] for r in sel]) * 1e6:.1f}')
if prof:
    import pstats
    pstats.Stats(prof).sort_stats('tottime').print_stats(40)
