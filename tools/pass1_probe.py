"""What would the candidate pass (AFF_SCORE mode 1) cost if it only had to recompute the tiles whose SECOND largest score reaches the query's
threshold (a score pass that kept maximum + token + second per tile; tools/top2_probe.py counts the pairs)?  The recorded read-out plan of a bench
frame, its pass 1 timed (a) as it runs today -- skipping by the tile maxima -- and (b) with a dense-fp32 second-largest matrix in the maxima's place
(the execution pattern of such a pass; the candidates it finds are not used).   python tools/pass1_probe.py [preroll frames]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import Recorder
from cutie_amd import _lib, ops as O
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval(); net.load_weights(make_state_dict(0))
rec = Recorder(_lib.get_executor()); _lib.set_executor_for_testing(rec)
clip = SyntheticClip(480, 854, 3, 64, seed=1)
proc = InferenceCore(net, cfg=cfg)
t1 = lambda a, it=20: min(rec.ex.time_ops(a, it) for _ in range(4)) * 1e3
with torch.inference_mode(), torch.cuda.stream(torch.cuda.Stream()):
    proc.step(clip.frame(0).cuda(), clip.first_mask().cuda(), objects=clip.objects)
    for t in range(1, pre): proc.step(clip.frame(t % 64).cuda())
    torch.cuda.synchronize()
    rec.rec, rec.on = [], True
    proc.step(clip.frame(5).cuda()); rec.on = False
    torch.cuda.synchronize()
    affs = np.concatenate([a[a['kind'] != O.USAGE_TICK] for a in rec.rec if (a['kind'] == O.AFF_SCORE).any()])
    assert [int(k) for k in affs['kind']] == [O.AFF_SCORE, O.AFF_SELECT, O.AFF_SCORE, O.AFF_READOUT]
    ii = affs['i'][0]
    HW, HWp, G = int(ii[0]), int(ii[1]), int(ii[9])
    Gld = -(-G // 64) * 64
    rec.ex.run(affs[:2]); torch.cuda.synchronize()                 # maxima + thresholds of this frame
    gbuf = proc.memory._scratch['gmax_tau']
    assert gbuf.data_ptr() == int(affs['p'][0, 6]) and int(affs['p'][2, 6]) == gbuf.data_ptr() + 4 * HWp * Gld
    gmax = gbuf[:HWp * Gld].view(HWp, Gld)
    tau = gbuf[HWp * Gld:HWp * Gld + HWp]
    # dense second-largest per 16-token tile from the bank's fp32 keys (memory_utils.get_similarity); tiles in the launch's order (range by range)
    b = next(iter(proc.memory.buckets.values()))
    ranges = [(int(ii[3 + 2 * r]), int(ii[4 + 2 * r])) for r in range(int(ii[2]))]
    # the query's key / selection: re-derive from the recorded KEY_PREP (query side) launch of the frame
    kp = [a for arr in rec.rec for a in arr if a['kind'] == O.KEY_PREP and (a['flags'] & 3) == 1]
    assert kp, 'no query-side KEY_PREP recorded'
    kp = kp[-1]
    import ctypes

    def dev(ptr, n, dtype):                                     # a device array known by its address only -> tensor (device-to-device copy)
        t = torch.empty((n,), dtype=dtype, device='cuda')
        torch.cuda.synchronize()
        hip = ctypes.CDLL('libamdhip64.so')
        hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(int(ptr)), ctypes.c_size_t(t.numel() * t.element_size()), 3)
        return t
    qk = dev(kp['p'][0], HW * 64, torch.float32).view(HW, 64).t()
    qe = dev(kp['p'][1], HW * 64, torch.float32).view(HW, 64).t()
    sec_rows = []
    for s, n in ranges:
        mk = b.rawkey[s:s + n].float().t(); ms_ = b.rawshr[s:s + n].float()
        sim = (-(mk.pow(2).t() @ qe) + 2 * (mk.t() @ (qk * qe)) - (qe * qk.pow(2)).sum(0, keepdim=True)) * ms_[:, None] / 8.0
        T = -(-n // 16)
        sim = torch.cat([sim, torch.full((T * 16 - n, HW), float('-inf'), device='cuda')]).view(T, 16, HW)
        sec_rows.append(sim.topk(2, dim=1)[0][:, 1])          # [T, HW]
    sec = torch.cat(sec_rows).t().contiguous()                 # [HW, G]
    assert sec.shape == (HW, G), (sec.shape, HW, G)
    g2 = torch.full((HWp * Gld + HWp,), float('-inf'), device='cuda')
    g2[:HWp * Gld].view(HWp, Gld)[:HW, :G] = sec
    g2[HWp * Gld:] = tau
    thr = tau[:HW] - tau[:HW].abs() * 1e-6 - 1e-30
    frac = lambda m: float((m[:HW, :G] >= thr[:, None]).float().mean())
    print('tokens', sum(n for _, n in ranges), 'tiles', G, '| (tile, query) pairs at or above the threshold: maxima %.4f, second largest %.4f' % (frac(gmax), frac(g2[:HWp * Gld].view(HWp, Gld))))
    torch.cuda.synchronize()
    a = affs.copy()
    base = t1(a[2:3])
    a2 = affs.copy(); a2['p'][2, 6] = g2.data_ptr() + 4 * HWp * Gld
    new = t1(a2[2:3])
    a3 = affs.copy(); a3['i'][2, 15] = 1; a4 = a2.copy(); a4['i'][2, 15] = 1
    print('candidate pass, register-staged kernel: today %.1f us, fed with the second-largest matrix %.1f us' % (base, new))
    print('candidate pass, LDS-DMA kernel:         today %.1f us, fed with the second-largest matrix %.1f us' % (t1(a3[2:3]), t1(a4[2:3])))
    g3 = torch.full((HWp * Gld + HWp,), float('-inf'), device='cuda'); g3[HWp * Gld:] = tau
    a5 = affs.copy(); a5['p'][2, 6] = g3.data_ptr() + 4 * HWp * Gld
    a6 = affs.copy(); a6['flags'][2] &= ~1                                    # no skipping: every tile recomputed
    g4 = g3.clone(); g4[HWp * Gld:] = float('inf')                            # thresholds nobody reaches: no candidate, no append, no flush
    a7 = affs.copy(); a7['p'][2, 6] = g4.data_ptr() + 4 * HWp * Gld
    print('candidate pass with NO tile at the threshold (floor of the streaming structure) %.1f us; without the skip (every tile recomputed) %.1f us; '
          'thresholds at +inf and no tile %.1f us' % (t1(a5[2:3]), t1(a6[2:3]), t1(a7[2:3])))
    a10 = a7.copy(); a10['flags'][2] &= ~1
    print('every tile recomputed, thresholds at +inf (the matrix work of the pass without any extraction) %.1f us' % t1(a10[2:3]))
    for tpb in (8, 16, 32, 64):
        a8 = affs.copy(); a8['i'][2, 13] = tpb
        a9 = a5.copy(); a9['i'][2, 13] = tpb
        print('  tiles per block %2d: today %.1f us, floor %.1f us' % (tpb, t1(a8[2:3]), t1(a9[2:3])))
    print('score pass %.1f us, selection %.1f us, read-out %.1f us' % (t1(a[0:1]), t1(a[0:2]) - t1(a[0:1]), t1(a[3:4])))
