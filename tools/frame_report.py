"""Launch-by-launch device time of one propagated frame (480p, K objects), in execution order.
Run on the MI355X box:  python tools/frame_report.py [--mem-frame]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--objects', type=int, default=3)
    ap.add_argument('--mem-frame', action='store_true')
    args = ap.parse_args()
    from bench import Recorder
    from cutie_amd import _lib, ops as O
    from cutie_amd.config import default_config
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.model.cutie import CUTIE
    from cutie_amd.utils.synth import SyntheticClip
    from cutie_amd.utils.synth_weights import make_state_dict
    cfg = default_config(use_long_term=True)
    net = CUTIE(cfg).cuda().eval()
    net.load_weights(make_state_dict(0))
    rec = Recorder(_lib.get_executor())
    _lib.set_executor_for_testing(rec)
    clip = SyntheticClip(480, 854, args.objects, 32, seed=1)
    proc = InferenceCore(net, cfg=cfg)
    side = torch.cuda.Stream()
    with torch.inference_mode(), torch.cuda.stream(side):
        proc.step(clip.frame(0).cuda(), clip.first_mask().cuda(), objects=clip.objects)
        last = 15 if args.mem_frame else 12           # mem_every = 5: frame 15 is a memory frame
        for t in range(1, last):
            proc.step(clip.frame(t).cuda())
        torch.cuda.synchronize()
        rec.rec, rec.on = [], True
        proc.step(clip.frame(last).cuda())
        rec.on = False
        torch.cuda.synchronize()
        ops = np.concatenate(rec.rec)
        whole = rec.ex.time_ops(ops, 10) * 1e3
        tot = 0.0
        print(f'{len(ops)} launches; whole frame replayed back-to-back: {whole:.1f} us')
        for n in range(len(ops)):
            one = ops[n:n + 1].copy()
            us = rec.ex.time_ops(one, 10) * 1e3
            tot += us
            i = one['i'][0]
            print(f'{n:4d} {O.KIND_NAMES.get(int(one["kind"][0]), "?"):>14s} {us:7.1f}  i={list(int(v) for v in i[:12])}')
        print(f'sum of isolated launches: {tot:.1f} us')


if __name__ == '__main__':
    main()
