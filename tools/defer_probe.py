"""No-look-ahead loop: wall time per frame and host time per step(), memory frames vs others, with the deferred memorising on / off."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from cutie_amd.inference import inference_core as IC
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.config import default_config
from cutie_amd.utils.synth_weights import make_state_dict

net = CUTIE(default_config()).cuda().eval()
net.load_weights(make_state_dict(seed=0))
clip = SyntheticClip(480, 854, 3, 64, seed=1)
frames = torch.stack([clip.frame(t) for t in range(64)]).cuda()
mask = clip.first_mask().cuda()
from cutie_amd.model import plans as PL
from cutie_amd import _lib
for defer, graphs in ((True, True), (False, True), (True, False), (False, False), (True, True), (False, True)):
    IC.DEFER_MEM = defer
    PL.GRAPHS = graphs
    _lib.get_executor().graph_stats[:] = [0, 0]
    proc = IC.InferenceCore(net, cfg=default_config(use_long_term=True))
    with torch.inference_mode():
        proc.step(frames[0], mask, objects=clip.objects)
        for t in range(1, 120):
            proc.step(frames[t % 64])
        torch.cuda.synchronize()
        host = {True: [], False: []}
        t0 = time.perf_counter()
        for t in range(120, 320):
            a = time.perf_counter()
            proc.step(frames[t % 64])
            host[(t % 5) == 0].append(time.perf_counter() - a)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 200
    print(f'defer={defer} graphs={graphs} (plans eager / replayed: {_lib.get_executor().graph_stats}): {1 / wall:.1f} fps, wall {wall * 1e6:.0f} us/frame; host per step: mem frames {sum(host[True]) / len(host[True]) * 1e6:.0f} us, '
          f'others {sum(host[False]) / len(host[False]) * 1e6:.0f} us', flush=True)
