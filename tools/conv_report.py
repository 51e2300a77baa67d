"""Per-launch device time of every conv of one propagated frame (480p, K objects): shape, autotuned tile, us, TFLOP/s.
Run on the MI355X box:  python tools/conv_report.py [--objects 3]"""
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
        for t in range(1, 12):
            proc.step(clip.frame(t).cuda())
        torch.cuda.synchronize()
        rec.rec, rec.on = [], True
        proc.step(clip.frame(12).cuda())
        rec.on = False
        if args.mem_frame:
            rec.rec, rec.on = [], True
            for t in (13, 14, 15):
                rec.rec = []
                proc.step(clip.frame(t).cuda())
            rec.on = False
        torch.cuda.synchronize()
        ops = np.concatenate(rec.rec)
        rows = []
        for n in range(len(ops)):
            if ops['kind'][n] != O.CONV:
                continue
            one = ops[n:n + 1].copy()
            us = rec.ex.time_ops(one, 10) * 1e3
            i = one['i'][0]
            M, cout, k, cin = int(i[0]) * int(i[7]) * int(i[8]), int(i[9]), int(i[11]), int(i[18])
            fl = 2.0 * M * cout * k * k * cin
            rows.append((us, M, cin, cout, k, int(i[13]), int(i[17]), fl / us / 1e6, int(i[19])))
        tot = sum(r[0] for r in rows)
        print(f'{len(rows)} convs, {tot:.1f} us back-to-back (isolated timing), {sum(r[7] * r[0] for r in rows) / tot:.1f} TFLOP/s average')
        print('    us      M   Cin  Cout k s tile splitk  TFLOP/s   cum%')
        cum = 0.0
        for r in sorted(rows, key=lambda r: -r[0]):
            cum += r[0]
            print(f'{r[0]:7.1f} {r[1]:6d} {r[2]:5d} {r[3]:5d} {r[4]} {r[5]} {str(O.TILES.get(r[6], "c1")) + ("x%d" % O.TILE_WK[r[6]] if r[6] in O.TILE_WK else ""):>17s} {r[8]:2d} {r[7]:8.1f} {cum / tot * 100:6.1f}')


if __name__ == '__main__':
    main()
