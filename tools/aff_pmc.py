"""Replay the affinity score pass (mode 0; AFF_PMC_ALL=1: the whole 5-op affinity plan) of a warmed-up 480p clip N times:
a target for tools/pmc_kernel.sh, e.g.  PMC_KERNEL=aff_score bash tools/pmc_kernel.sh python tools/aff_pmc.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import Recorder
from cutie_amd import _lib, ops as O
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval(); net.load_weights(make_state_dict(0))
rec = Recorder(_lib.get_executor()); _lib.set_executor_for_testing(rec)
clip = SyntheticClip(480, 854, 3, 64, seed=1)
proc = InferenceCore(net, cfg=cfg)
with torch.inference_mode(), torch.cuda.stream(torch.cuda.Stream()):
    proc.step(clip.frame(0).cuda(), clip.first_mask().cuda(), objects=clip.objects)
    for t in range(1, 300): proc.step(clip.frame(t % 64).cuda())
    torch.cuda.synchronize()
    rec.rec, rec.on = [], True
    proc.step(clip.frame(5).cuda()); rec.on = False
    torch.cuda.synchronize()
    affs = np.concatenate([a[a['kind'] != O.USAGE_TICK] for a in rec.rec if (a['kind'] == O.AFF_SCORE).any()])
    a = affs.copy() if os.environ.get('AFF_PMC_ALL') else affs[1:2].copy()
    for v in sys.argv[1:]:
        k, x = v.split('='); a['i'][0, int(k)] = int(x)
    for _ in range(20):
        rec.ex.run(a)
    torch.cuda.synchronize()
    print('done', sum(b.size() for b in proc.memory.buckets.values()))
