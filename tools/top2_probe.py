"""How many (query, 16-token tile) pairs would a candidate pass still have to recompute if the score pass kept, per tile, the maximum WITH its token
and the second largest score?  (Today: every tile whose maximum reaches the query's threshold, ~top_k tiles per query.)  Dense fp32 similarity of the
bench clip's bank against one query frame; prints the two counts per query.   python tools/top2_probe.py [--frames 60]"""
import argparse
import torch
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=60)
ap.add_argument('--objects', type=int, default=3)
a = ap.parse_args()
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval()
net.load_weights(make_state_dict(seed=0))
clip = SyntheticClip(480, 854, a.objects, a.frames + 1, seed=101)
with torch.inference_mode():
    proc = InferenceCore(net, cfg=cfg)
    proc.step(clip.frame(0).cuda(), clip.first_mask().cuda(), objects=clip.objects)
    for t in range(1, a.frames):
        proc.step(clip.frame(t).cuda())
    b = next(iter(proc.memory.buckets.values()))
    img = clip.frame(a.frames).cuda()
    # query operands of the last frame: recompute key / selection through the facade
    H, W = 480, 864
    x = torch.zeros((1, 3, H, W), device='cuda'); x[:, :, :, 5:859] = img
    ms, pix = net.encode_image(x)
    key, shr, sel = net.transform_key(ms[0])
    qk = key[0].flatten(1).float()                      # [64, HW]
    qe = sel[0].flatten(1).float()
    slots = torch.cat([torch.arange(s, s + n) for s, n in b.ranges() if n > 0]).cuda()
    mk = b.rawkey[slots].float().t()                    # [64, N] (kept for the long-term consolidation)
    ms_ = b.rawshr[slots].float()
    # memory_utils.get_similarity
    a_sq = (mk.pow(2).t() @ qe)
    two_ab = 2 * (mk.t() @ (qk * qe))
    b_sq = (qe * qk.pow(2)).sum(0, keepdim=True)
    sim = (-a_sq + two_ab - b_sq) * ms_[:, None] / 8.0  # [N, HW]
    N = sim.shape[0]
    T = -(-N // 16)
    pad = torch.full((T * 16 - N, sim.shape[1]), float('-inf'), device='cuda')
    s3 = torch.cat([sim, pad]).view(T, 16, -1)
    top2 = s3.topk(2, dim=1)[0]                         # [T, 2, HW]
    tau = top2[:, 0].topk(30, dim=0)[0][-1]             # 30th largest tile maximum per query
    n1 = (top2[:, 0] >= tau).sum(0).float()
    n2 = (top2[:, 1] >= tau).sum(0).float()
    ncand = (sim >= tau).sum(0).float()
    print('tokens', N, 'tiles', T, 'queries', sim.shape[1])
    print('tiles with max >= tau per query: mean %.1f' % n1.mean().item())
    print('tiles with SECOND >= tau per query: mean %.2f  (%.1f %% of the above)' % (n2.mean().item(), 100 * n2.sum().item() / n1.sum().item()))
    print('candidates per query: mean %.1f max %d' % (ncand.mean().item(), int(ncand.max())))
    # per (tile, 32-query wave) pairs that need a recompute
    HW = sim.shape[1]
    q32 = -(-HW // 32)
    def pairs(m):
        mm = torch.cat([m, torch.zeros((T, q32 * 32 - HW), dtype=torch.bool, device='cuda')], 1).view(T, q32, 32).any(2)
        return mm.float().mean().item()
    print('fraction of (tile, 32-query) pairs executed: today %.3f, with top-2 kept %.3f' % (pairs(top2[:, 0] >= tau), pairs(top2[:, 1] >= tau)))
    # the candidate pass's wave-private lists: hits per (32-query wave, block row of the launch) -- a list holds 256 entries (AFF_WCAP), what does not
    # fit goes to the global lists entry by entry
    thr = tau - tau.abs() * 1e-6 - 1e-30
    hits = (sim >= thr[None, :]).float()                                # [N, HW]
    for ny in (39, 20):
        tiles_per_row = -(-T // ny)
        rows_tok = tiles_per_row * 16
        hp = torch.cat([hits, torch.zeros((ny * rows_tok - N, HW), device='cuda')]) if ny * rows_tok > N else hits[:ny * rows_tok]
        hq = torch.cat([hp, torch.zeros((hp.shape[0], q32 * 32 - HW), device='cuda')], 1)
        per = hq.view(ny, rows_tok, q32, 32).sum((1, 3))                # [block row, wave]
        print('block rows %d: hits per (wave, block row): mean %.1f, p99 %.0f, max %.0f; lists over 256 entries: %d of %d (overflowing entries %.0f of %.0f)' % (
            ny, per.mean().item(), per.flatten().kthvalue(int(0.99 * per.numel()))[0].item(), per.max().item(), int((per > 256).sum()), per.numel(),
            (per - 256).clamp(min=0).sum().item(), per.sum().item()))
