"""Every conv of a 480p frame x every legal tile of every conv kernel (register-staged 0-44, buffer-load 50-56, LDS-DMA 60-69):
isolated device time (hipEvents on the launch stream, min of REPS x ITERS launches).  Prints, per conv geometry, the best tile of
each kernel family and writes the full table + the per-geometry winners (a tile table in the format of
cutie_amd/tiles_gfx950.json) under gpurun_out/.  Run on the MI355X box:

    python tools/conv_sweep.py [--objects 3] [--reps 3] [--iters 12] [--out gpurun_out/conv_sweep]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def family(t, O):
    if t in O.PC_HALO:
        return 'halo'
    if t in O.PC_TILES:
        return 'pc'
    if t in O.DMA_TILES:
        return 'dma'
    return 'igemm'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--objects', type=int, nargs='+', default=[3])
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--iters', type=int, default=12)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=854)
    ap.add_argument('--out', default='gpurun_out/conv_sweep')
    ap.add_argument('--cold', type=int, default=0, help='MiB copied through the device before every timed launch (evicts code and operands from the L2s: the state a layer finds inside a frame); 0 = warm back-to-back timing')
    ap.add_argument('--families', default='', help='comma list: time only tiles of these families (igemm,dma,pc,halo); default all')
    ap.add_argument('--window', type=int, nargs='*', default=[], help='sweep the convs of the BATCHED image encoder (CUTIE._encode_window, the look-ahead '
                    'window of InferenceCore) for these batch sizes instead of the frame\'s convs; candidates are restricted to the K-order class of '
                    'the one-frame plan\'s tile (ops.korder_class), the only tiles the window plan may use')
    ap.add_argument('--lockstep', type=int, default=0, help='sweep the convs of LOCK-STEP frames of this many clips (cutie_amd/inference/lockstep.py: batch = clips x '
                    'objects, joint encoder windows) instead; candidates are restricted to the K-order class of the tile the plan runs (that of the one-clip plan)')
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
    geoms = {}
    want_class = {}
    with torch.inference_mode():
        for B in args.window:
            clip = SyntheticClip(args.height, args.width, 1, 16, seed=1)
            imgs = [clip.frame(t).cuda().float().contiguous() for t in range(B)]
            from cutie_amd.inference.inference_core import pad_geometry
            H, W, pad = pad_geometry(args.height, args.width, 16)
            g = (args.height, args.width, H, W, pad[0], pad[2])
            net._encode_window(imgs, *g)                  # builds the plan (and applies the table with the class rule)
            net._encode(imgs[0], *g)
            torch.cuda.synchronize()
            rec.rec, rec.on = [], True
            net._encode(imgs[0], *g)
            one_ops = np.concatenate(rec.rec)
            rec.rec = []
            net._encode_window(imgs, *g)
            rec.on = False
            torch.cuda.synchronize()
            ops = np.concatenate(rec.rec)
            c1 = [n for n in range(len(one_ops)) if one_ops['kind'][n] == O.CONV]
            cB = [n for n in range(len(ops)) if ops['kind'][n] == O.CONV]
            assert len(c1) == len(cB)
            for n1, n in zip(c1, cB):
                i = ops['i'][n]
                M, cout, cin = int(i[0]) * int(i[7]) * int(i[8]), int(i[9]), int(i[3]) + int(i[4])
                key = (M, cout, cin, int(i[11]), int(i[13]), int(ops['flags'][n]) & 3, int(i[1]), int(i[2]))
                want_class[key] = O.korder_class(int(one_ops['i'][n1][17]), int(one_ops['i'][n1][19]))
                if key not in geoms:
                    geoms[key] = [ops[n:n + 1].copy(), 0]
                geoms[key][1] += 1
        for K in (args.objects if args.lockstep else []):
            from cutie_amd.inference.lockstep import LockstepCores
            C, NF = args.lockstep, 40
            clips = [SyntheticClip(args.height, args.width, K, NF, seed=1 + c) for c in range(C)]
            frames = [[cl.frame(t).cuda() for t in range(NF)] for cl in clips]
            ls = LockstepCores(net, cfg, C)
            ls.step([f[0] for f in frames], [cl.first_mask().cuda() for cl in clips], [cl.objects for cl in clips])
            hint = lambda t: dict(next_images=[f[t + 1:t + 14] for f in frames])
            for t in range(1, 14):
                ls.step([f[t] for f in frames], **hint(t))
            torch.cuda.synchronize()
            rec.rec, rec.on = [], True
            for t in range(14, 26):                      # whole encoder batches and two memory frames
                ls.step([f[t] for f in frames], **hint(t))
            rec.on = False
            torch.cuda.synchronize()
            ops = np.concatenate(rec.rec)
            for n in range(len(ops)):
                if ops['kind'][n] != O.CONV:
                    continue
                i = ops['i'][n]
                M, cout, cin = int(i[0]) * int(i[7]) * int(i[8]), int(i[9]), int(i[3]) + int(i[4])
                key = (M, cout, cin, int(i[11]), int(i[13]), int(ops['flags'][n]) & 3, int(i[1]), int(i[2]))
                if key not in geoms:
                    geoms[key] = [ops[n:n + 1].copy(), 0]
                    geoms[key][0]['p'][0, 7] = geoms[key][0]['p'][0, 8] = 0
                    geoms[key][0]['i'][0, 21] = 0
                    want_class[key] = O.korder_class(int(i[17]), int(i[19]))       # (enforced by Plan.autotune_convs: the one-clip plan's class)
                geoms[key][1] += 1
        for K in ([] if (args.window or args.lockstep) else args.objects):
            clip = SyntheticClip(args.height, args.width, K, 16, seed=1)
            proc = InferenceCore(net, cfg=cfg)
            proc.step(clip.frame(0).cuda(), clip.first_mask().cuda(), objects=clip.objects)
            for t in range(1, 4):
                proc.step(clip.frame(t).cuda())
            torch.cuda.synchronize()
            rec.rec, rec.on = [], True
            for t in range(4, 10):                       # covers a memory frame (mem_every = 5)
                proc.step(clip.frame(t).cuda())
            rec.on = False
            torch.cuda.synchronize()
            ops = np.concatenate(rec.rec)
            for n in range(len(ops)):
                if ops['kind'][n] != O.CONV:
                    continue
                i = ops['i'][n]
                M, cout, cin = int(i[0]) * int(i[7]) * int(i[8]), int(i[9]), int(i[3]) + int(i[4])
                key = (M, cout, cin, int(i[11]), int(i[13]), int(ops['flags'][n]) & 3, int(i[1]), int(i[2]))
                if key not in geoms:
                    geoms[key] = [ops[n:n + 1].copy(), 0]
                    geoms[key][0]['p'][0, 7] = geoms[key][0]['p'][0, 8] = 0     # GAP accumulation / zero side jobs exist on the LDS-DMA tiles only
                    geoms[key][0]['i'][0, 21] = 0
                geoms[key][1] += 1
    flush, t_flush = None, 0.0
    if args.cold:
        nb = args.cold << 20
        fsrc, fdst = torch.zeros(nb, dtype=torch.uint8, device='cuda'), torch.zeros(nb, dtype=torch.uint8, device='cuda')
        fl = O.OpList()
        fl.copy2d(fsrc, fdst, rows=args.cold, rowbytes=1 << 20, src_stride=1 << 20, dst_stride=1 << 20)
        flush = fl.finalize()
        for _ in range(3):
            rec.ex.run(flush)
        torch.cuda.synchronize()
        t_flush = min(rec.ex.time_ops(flush, 10) for _ in range(5)) * 1e3
        print(f'cold mode: {args.cold} MiB flush = {t_flush:.1f} us before every launch')

    def time_one(one):
        if flush is None:
            return min(rec.ex.time_ops(one, args.iters) for _ in range(args.reps)) * 1e3
        seq = np.concatenate([flush, one])
        return min(rec.ex.time_ops(seq, max(2, args.iters // 2)) for _ in range(args.reps)) * 1e3 - t_flush

    rows, table = [], []
    for key, (one, count) in geoms.items():
        i = one['i'][0]
        M, cout, cin, k = key[0], key[1], key[2], key[3]
        cands = O.tile_candidates(M, cout, cin, int(i[16]), geom=dict(kh=k, stride=int(i[13]), pad=int(i[14]), W=int(i[2]), c2=int(i[4])))
        if int(i[4]) or one['p'][0, 4]:
            cands = [t for t in cands if t != O.COUT1_TILE]
        if args.families:
            keep = set(args.families.split(','))
            cands = [t for t in cands if family(t, O) in keep or t == int(i[17])]
        if key in want_class:
            cands = [t for t in cands if O.korder_class(t) == want_class[key]] or [int(i[17])]
        res = {}
        for t in cands:
            for sk in O.splitk_candidates(M, cout, int(i[16]), t):
                one['i'][0, 17], one['i'][0, 19] = t, sk
                try:
                    us = time_one(one)
                except RuntimeError as e:
                    print('  tile', t, 'failed on', key, str(e)[:120])
                    continue
                res[(t, sk)] = us
        fl = 2.0 * M * cout * k * k * int(i[18])
        best = {}
        for (t, sk), us in res.items():
            f = family(t, O)
            if f not in best or us < best[f][1]:
                best[f] = ((t, sk), us)
        (bt, bsk), bus = min(res.items(), key=lambda kv: kv[1])
        # lock-step geometries: keyed with their K-order class (a batched geometry may be shared by layers of different classes; Plan.autotune_convs)
        table.append([list(key) + ([want_class[key]] if (args.lockstep and isinstance(want_class.get(key), str)) else []), [bt, bsk]])
        rows.append(dict(key=list(key), count=count, gflop=fl / 1e9, best=[bt, bsk, bus],
                         families={f: [v[0][0], v[0][1], v[1]] for f, v in best.items()},
                         all={f'{t}x{sk}': us for (t, sk), us in res.items()}))
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(rows, open(args.out + '.json', 'w'))
    json.dump({'tiles': table}, open(args.out + '_tiles.json', 'w'))
    fam = [f for f in ['igemm', 'dma', 'pc', 'halo'] if any(f in r['families'] for r in rows)]
    tot = {f: 0.0 for f in fam + ['best', 'old']}
    print(f'{"M":>6s} {"Cout":>5s} {"Cin":>5s} k s fl     HxW  n  GFLOP |' + ''.join(f' {f:>14s}' for f in fam) + ' |  best TFLOP/s')
    for r in sorted(rows, key=lambda r: -r['best'][2] * r['count']):
        key = r['key']
        cells = ''
        for f in fam:
            v = r['families'].get(f)
            cells += f' {v[0]:3d}x{v[1]:<2d} {v[2]:7.1f}' if v else ' ' * 15
        old = min([v[2] for f, v in r['families'].items() if f in ('igemm', 'dma')] or [r['best'][2]])
        tot['best'] += r['best'][2] * r['count']
        tot['old'] += old * r['count']
        print(f'{key[0]:6d} {key[1]:5d} {key[2]:5d} {key[3]} {key[4]} {key[5]} {key[6]:4d}x{key[7]:<4d} {r["count"]:2d} {r["gflop"]:6.2f} |{cells} | '
              f'{r["best"][0]:3d} {r["gflop"] / r["best"][2] * 1e3:7.1f}')
    print(f'sum over the recorded frames: best-of-all {tot["best"]:.1f} us, best of the round-2 kernels that are left (igemm/dma) {tot["old"]:.1f} us '
          f'({len(rows)} geometries, {sum(r["count"] for r in rows)} launches)')


if __name__ == '__main__':
    main()
