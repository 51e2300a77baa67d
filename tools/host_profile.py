"""Host-side cost of InferenceCore.step (time to ISSUE a frame, no device sync) + cProfile of 100 frames."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval(); net.load_weights(make_state_dict(0))
clip = SyntheticClip(480, 854, 3, 64, seed=1)
frames = torch.stack([clip.frame(t) for t in range(64)]).cuda()
proc = InferenceCore(net, cfg=cfg)
with torch.inference_mode(), torch.cuda.stream(torch.cuda.Stream()):
    proc.step(frames[0], clip.first_mask().cuda(), objects=clip.objects)
    for t in range(1, 320): proc.step(frames[t % 64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(100): proc.step(frames[t % 64])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host issue time {1e3 * (t1 - t0) / 100:.3f} ms/frame ; with device drain {1e3 * (t2 - t0) / 100:.3f} ms/frame')
    iso = []
    for t in range(40):
        torch.cuda.synchronize()
        a = time.perf_counter(); proc.step(frames[t % 64]); iso.append(time.perf_counter() - a)
    iso.sort()
    print(f'host issue time with an EMPTY queue: median {1e3 * iso[len(iso) // 2]:.3f} ms/frame, min {1e3 * iso[0]:.3f}')
    pr = cProfile.Profile(); pr.enable()
    for t in range(100): proc.step(frames[t % 64])
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
