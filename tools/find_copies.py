"""Which host lines still launch torch kernels / device copies inside a steady-state InferenceCore.step?  (torch.profiler, stacks)
Run on the MI355X box: python tools/find_copies.py"""
import os, sys, collections, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict
from torch.profiler import profile, ProfilerActivity
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval(); net.load_weights(make_state_dict(0))
clip = SyntheticClip(480, 854, 3, 64, seed=1)
frames = torch.stack([clip.frame(t) for t in range(64)]).cuda()
proc = InferenceCore(net, cfg=cfg)
with torch.inference_mode():
    proc.step(frames[0], clip.first_mask().cuda(), objects=clip.objects, next_image=frames[1])
    for t in range(1, 60): proc.step(frames[t % 64], next_image=frames[(t + 1) % 64])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for t in range(60, 70): proc.step(frames[t % 64], next_image=frames[(t + 1) % 64])
        torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.name.startswith('aten::') and e.name not in ('aten::empty', 'aten::empty_strided', 'aten::view', 'aten::permute', 'aten::unsqueeze',
            'aten::slice', 'aten::select', 'aten::as_strided', 'aten::reshape', 'aten::_unsafe_view', 'aten::squeeze', 'aten::t', 'aten::transpose',
            'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::clone', 'aten::empty_like', 'aten::resolve_conj', 'aten::resolve_neg',
            'aten::lift_fresh', 'aten::detach_', 'aten::alias', 'aten::expand', 'aten::is_nonzero', 'aten::item', 'aten::_local_scalar_dense', 'aten::result_type', 'aten::can_cast'):
        st = [s for s in (e.stack or []) if 'cutie_amd' in s or 'bench' in s or 'find_copies' in s]
        agg[(e.name, st[0] if st else '?')] += 1
for (name, where), n in agg.most_common(40):
    print(f'{n / 10:6.1f} per frame  {name:28s} {where}')
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12))
