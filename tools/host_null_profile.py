"""Pure host cost of InferenceCore.step, measurable WITHOUT a GPU: the descriptors are handed to an executor that does nothing
(CPU tensors stand in for the buffers), so what is timed is the Python frame scheduler + descriptor filling + torch view /
allocation calls -- the part that bounds the multi-clip mode (one GIL) once the device is fast enough.
    python tools/host_null_profile.py [--profile]"""
import argparse, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict


class NullExecutor:
    is_mock = True
    launches = 0
    calls = 0

    def run(self, arr):
        self.launches += len(arr)
        self.calls += 1

    def time_ops(self, arr, iters):
        return 1.0


ap = argparse.ArgumentParser()
ap.add_argument('--profile', action='store_true')
ap.add_argument('--h', type=int, default=480)
ap.add_argument('--w', type=int, default=854)
args = ap.parse_args()
os.environ.setdefault('CUTIE_AMD_TUNE', '0')
ex = NullExecutor()
_lib.set_executor_for_testing(ex)
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).eval(); net.load_weights(make_state_dict(0))
clip = SyntheticClip(args.h, args.w, 3, 8, seed=1)
frames = [clip.frame(t) for t in range(8)]
proc = InferenceCore(net, cfg=cfg)
with torch.inference_mode():
    proc.step(frames[0], clip.first_mask(), objects=clip.objects)
    for t in range(1, 60): proc.step(frames[t % 8], next_image=frames[(t + 1) % 8])
    ex.launches = ex.calls = 0
    n = 200
    t0 = time.perf_counter()
    for t in range(n): proc.step(frames[t % 8], next_image=frames[(t + 1) % 8])
    dt = time.perf_counter() - t0
    print(f'host time {1e3 * dt / n:.3f} ms/frame, {ex.launches / n:.1f} descriptors in {ex.calls / n:.1f} exec calls per frame')
    if args.profile:
        pr = cProfile.Profile(); pr.enable()
        for t in range(100): proc.step(frames[t % 8], next_image=frames[(t + 1) % 8])
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(40)
