"""Aggregate frames/s with N clips in flight on ONE GPU (one host thread + HIP stream + CUTIE.fork() per clip).
Run on the MI355X box: python tools/multistream_probe.py"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd.config import default_config
from cutie_amd.inference.inference_core import InferenceCore
from cutie_amd.model.cutie import CUTIE
from cutie_amd.utils.synth import SyntheticClip
from cutie_amd.utils.synth_weights import make_state_dict

cfg = default_config(use_long_term=True)
net = CUTIE(cfg).cuda().eval(); net.load_weights(make_state_dict(0))
NF, STEPS, PRE = 64, 150, 60


def make(i):
    clip = SyntheticClip(480, 854, 3, NF, seed=1 + i)
    return torch.stack([clip.frame(t) for t in range(NF)]).cuda(), clip.first_mask().cuda(), clip.objects


def worker(n, frames, mask, objs, stream, start, done, out, idx):
    with torch.inference_mode(), torch.cuda.stream(stream):
        proc = InferenceCore(n, cfg=cfg)
        proc.step(frames[0], mask, objects=objs)
        for t in range(1, PRE):
            proc.step(frames[t % NF])
        stream.synchronize()
        start.wait()
        for t in range(PRE, PRE + STEPS):
            proc.step(frames[t % NF])
        stream.synchronize()
        out[idx] = time.perf_counter()
        done.wait()


# tune on one stream first (the forks reuse the tile cache)
f0 = make(0)
with torch.inference_mode(), torch.cuda.stream(torch.cuda.Stream()):
    p = InferenceCore(net, cfg=cfg)
    p.step(f0[0][0], f0[1], objects=f0[2])
    for t in range(1, 12):
        p.step(f0[0][t])
    torch.cuda.synchronize()
for N in [int(a) for a in sys.argv[1:]] or (1, 2, 3, 4):
    data = [make(i) for i in range(N)]
    nets = [net] + [net.fork() for _ in range(N - 1)]
    start, done = threading.Barrier(N + 1), threading.Barrier(N + 1)
    out = [0.0] * N
    th = [threading.Thread(target=worker, args=(nets[i], *data[i], torch.cuda.Stream(), start, done, out, i)) for i in range(N)]
    for t in th: t.start()
    start.wait()
    t0 = time.perf_counter()
    done.wait()
    for t in th: t.join()
    el = max(out) - t0
    print(f'{N} clips in flight: {N * STEPS / el:7.1f} frames/s total  ({el / STEPS * 1e3:.2f} ms per frame-step)', flush=True)
