"""Weights under which the ORACLE is decisive -- so that "identical argmax object ids" can be asserted over whole frames.

Why: no trained checkpoint exists offline, and with the purely random synthetic weights (cutie_amd/utils/synth_weights.py) most
pixels of a propagated frame are nearly tied between classes: an argmax comparison then says nothing (whole-frame
agreement 0.27-0.98, and only ~25 % of bike's pixels have an oracle top-1 / top-2 margin above the tolerance).  An argmax is
scale-invariant, so no gain on the logits helps; the features must separate the classes.  This script fits the LAST layers of the
mask decoder (`mask_decoder.pred`, optionally `mask_decoder.up_8_4.out_conv.*`) of the oracle on synthetic clips whose ground truth is
known (the rectangles of cutie_amd.utils.synth.SyntheticClip move with the texture, 2 px per frame), leaving every other tensor of
the synthetic state dict alone, and stores the changed tensors (fp32, a few KB .. 1.2 MB) as tests/golden/decisive_delta.npz.
The recurrence makes the features depend on the head (masks are fed back through the memory), so a few rounds of
{run the oracle free, collect the decoder inputs, fit the head} are made.

    python -m oracle.make_decisive_weights [--rounds 2] [--steps 400] [--lr 2e-3] [--gain 4] [--train-up]

Committed delta: the defaults above (head only: 1153 numbers; class-balanced loss).  Measured here: oracle margin > 0.33 on
100 / 98.4 / 99.9 / 99.2 % of the pixels of the four bike frames (27.7 / 24.2 / 56.9 % with the plain synthetic weights), both objects
still present in every frame.  --train-up (also fitting up_8_4) tracked the synthetic clips better and bike worse: not used.

Prints, per round, the training loss and the fraction of pixels of every bike frame (and of held-out synthetic frames) whose oracle
margin exceeds 0.33.  TEST INFRASTRUCTURE (see oracle/__init__.py); the recipe is deterministic up to torch's CPU reduction order --
the committed delta, not a re-run, is what the tests load (`oracle.scenarios.decisive_state_dict`)."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd.utils.synth import SyntheticClip                      # noqa: E402
from oracle import scenarios as S                                    # noqa: E402
from oracle.inference import OracleProcessor, DEFAULT_CFG            # noqa: E402
from oracle.net import OracleNet, aggregate                          # noqa: E402
from oracle.weights import make_state_dict                           # noqa: E402

OUT = os.path.join(S.GOLDEN_DIR, 'decisive_delta.npz')


def truth(clip, t):
    """Index mask of frame t: the first-frame rectangles, moved with the texture (shift px per frame, bouncing like SyntheticClip.frame)."""
    tt = t % (2 * clip._period)
    if tt >= clip._period:
        tt = 2 * clip._period - tt
    off = tt * clip.shift
    m = torch.zeros((clip.h, clip.w), dtype=torch.long)
    for i, (y0, y1, x0, x1) in enumerate(clip.rects):
        a, b = max(x0 - off, 0), max(x1 - off, 0)
        if b > a:
            m[y0:y1, a:b] = i + 1
    return m


def collect(onet, clips, frames, mem_every=3):
    """Free-running oracle on every clip; the inputs of OracleNet.segment of every propagated frame + that frame's truth."""
    data = []
    seg = onet.segment
    for clip in clips:
        rec = []

        def spy(ms, ro, sens, update_sensory=True, _rec=rec):
            _rec.append((ms[1].clone(), ms[2].clone(), ro.clone()))
            return seg(ms, ro, sens, update_sensory=update_sensory)

        onet.segment = spy
        proc = OracleProcessor(onet, dict(DEFAULT_CFG, mem_every=mem_every))
        with torch.inference_mode():
            proc.step(clip.frame(0), clip.first_mask(), objects=clip.objects)
            for t in range(1, frames):
                proc.step(clip.frame(t))
        onet.segment = seg
        for t, (f8, f4, ro) in enumerate(rec, start=1):
            data.append((f8, f4, ro, truth(clip, t), (clip.h, clip.w)))
    return data


def head(W, f8, f4, ro, train_up):
    """OracleNet.segment's decoder (net.py:335-352) on detached inputs, with the trainable tensors taken from W."""
    net = head.net
    bs, K = ro.shape[:2]
    f8p = net.conv('mask_decoder.decoder_feat_proc.transforms.0', f8)
    f4p = net.conv('mask_decoder.decoder_feat_proc.transforms.1', f4)
    up = lambda g: F.interpolate(g.flatten(0, 1), scale_factor=2, mode='bilinear', align_corners=False).view(bs, K, -1, g.shape[-2] * 2, g.shape[-1] * 2)
    p8 = net._group_resblock('mask_decoder.up_16_8.out_conv', up(ro) + f8p.unsqueeze(1))
    keep = {}
    if train_up:
        for k in W:
            if k.startswith('mask_decoder.up_8_4'):
                keep[k] = net.W[k]
                net.W[k] = W[k]
    p4 = net._group_resblock('mask_decoder.up_8_4.out_conv', up(p8) + f4p.unsqueeze(1))
    for k, v in keep.items():
        net.W[k] = v
    logits = F.conv2d(F.relu(p4.flatten(0, 1)), W['mask_decoder.pred.weight'], W['mask_decoder.pred.bias'], 1, 1).view(bs, K, *p4.shape[-2:])
    lg = aggregate(torch.sigmoid(logits), dim=1)
    return F.interpolate(lg, scale_factor=4, mode='bilinear', align_corners=False)


def margins(onet, name, hist=False):
    steps, _ = S.scenario_inputs(name)
    proc = OracleProcessor(onet, dict(DEFAULT_CFG, **S.SCENARIOS[name]['cfg']))
    out = []
    with torch.inference_mode():
        for img, mask, objs in steps:
            p = proc.step(img, mask, objects=objs) if mask is not None else proc.step(img)
            top2 = p.topk(2, dim=0)[0]
            out.append(float(((top2[0] - top2[1]) > 0.33).float().mean()))
            if hist:
                print('   ', name, 'argmax histogram', torch.bincount(p.argmax(0).flatten(), minlength=p.shape[0]).tolist())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--clips', type=int, default=6)
    ap.add_argument('--train-up', action='store_true', help='also fit mask_decoder.up_8_4.out_conv (conv1, conv2)')
    ap.add_argument('--lr', type=float, default=2e-3)
    ap.add_argument('--gain', type=float, default=4.0, help='the fitted head is scaled by this before it is stored: sharper sigmoids -> wider margins '
                    '(the argmax of one frame is scale-invariant; through the recurrence the objects shrink a little). 1: 86-92 %% of bike decisive, 3: 96-100 %%, 4: 98-100 %%; at 5 the objects all but vanish')
    ap.add_argument('--out', default=OUT)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    sd = make_state_dict(seed=0)
    names = ['mask_decoder.pred.weight', 'mask_decoder.pred.bias']
    if args.train_up:
        names += [k for k in sd if k.startswith('mask_decoder.up_8_4.out_conv.') and sd[k].is_floating_point()]
    onet = OracleNet(sd)
    head.net = onet
    print('before: bike decisive fraction per frame', [round(v, 3) for v in margins(onet, 'bike')])
    clips = [SyntheticClip(96, 136, 3, args.frames, seed=21 + i) for i in range(args.clips)]
    clips += [SyntheticClip(128, 160, 2, args.frames, seed=41 + i) for i in range(2)]
    W = {k: onet.W[k].clone().requires_grad_(True) for k in names}
    for rnd in range(args.rounds):
        data = collect(onet, clips, args.frames)
        opt = torch.optim.Adam(list(W.values()), lr=args.lr)
        for it in range(args.steps):
            f8, f4, ro, gt, (h0, w0) = data[it % len(data)]
            lg = head(W, f8, f4, ro, args.train_up)                     # [1, K+1, H, W] (padded frame)
            H, Wd = lg.shape[-2:]
            ph, pw = (H - h0) // 2, (Wd - w0) // 2                       # pad_divide_by: symmetric
            # class-balanced: the rectangles cover a few per cent of a frame, and an unweighted fit ends at "background everywhere,
            # confidently" -- decisive, but an argmax test on it would compare constants
            cnt = torch.bincount(gt.flatten(), minlength=lg.shape[1]).float().clamp(min=1)
            loss = F.cross_entropy(lg[:, :, ph:ph + h0, pw:pw + w0], gt.unsqueeze(0), weight=cnt.sum() / (cnt * len(cnt)))
            opt.zero_grad()
            loss.backward()
            opt.step()
            if it % 100 == 0 or it == args.steps - 1:
                print(f'round {rnd} step {it}: loss {float(loss.detach()):.4f}')
        for k in names:
            onet.W[k] = W[k].detach().clone()
        print(f'round {rnd}: bike decisive fraction per frame', [round(v, 3) for v in margins(onet, 'bike')],
              ' small_fifo', [round(v, 3) for v in margins(onet, 'small_fifo')][:8])
    for k in names:
        onet.W[k] = onet.W[k] * args.gain
    print(f'gain {args.gain}: bike decisive fraction per frame', [round(v, 3) for v in margins(onet, 'bike', hist=True)])
    np.savez_compressed(args.out, **{k: onet.W[k].numpy() for k in names})
    print('wrote', args.out, {k: tuple(onet.W[k].shape) for k in names})


if __name__ == '__main__':
    main()
