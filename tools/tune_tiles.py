"""Thorough conv autotuning for the packaged tile table cutie_amd/tiles_gfx950.json (run on the MI355X box):
    CUTIE_AMD_TUNE=7x20 CUTIE_AMD_TILE_CACHE=gpurun_out/tiles_gfx950.json python tools/tune_tiles.py
Runs 480p clips with 1, 2 and 3 objects (long-term memory on, so the memory-frame plans are built too) long enough for every
plan to be built; every conv geometry met is timed with the effort given by $CUTIE_AMD_TUNE and written to the cache file."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd.config import default_config            # noqa: E402
from cutie_amd.inference.inference_core import InferenceCore   # noqa: E402
from cutie_amd.model.cutie import CUTIE               # noqa: E402
from cutie_amd.utils.synth import SyntheticClip       # noqa: E402
from cutie_amd.utils.synth_weights import make_state_dict            # noqa: E402

assert os.environ.get('CUTIE_AMD_TILE_CACHE', 'none') != 'none', 'set CUTIE_AMD_TILE_CACHE to the output file'
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval()
net.load_weights(make_state_dict(0))
net.engine().tile_cache.clear()                          # re-time everything, ignore the packaged table
net.engine().autotune = True
with torch.inference_mode():
    for K in (3, 1, 2):
        clip = SyntheticClip(480, 854, K, 14, seed=K)
        proc = InferenceCore(net, cfg=cfg)
        proc.step(clip.frame(0).cuda(), clip.first_mask().cuda(), objects=clip.objects)
        for t in range(1, 14):
            proc.step(clip.frame(t).cuda())
        torch.cuda.synchronize()
        print('K', K, 'geometries tuned so far:', len(net.engine().tile_cache), flush=True)
