#!/usr/bin/env python
"""bench.py -- frames/sec of the Cutie per-frame hot path (InferenceCore.step) on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "frames/sec (480p, 3 objects)"): synthetic 854x480
clip, 3 objects, long-term memory enabled, eval_config defaults (mem_every=5, top_k=30), bf16 storage / fp32
accumulate, seeded random-init weights (no checkpoint offline).  One step = one propagated frame through
InferenceCore.step with the frame already resident in HBM; the FPS definition mirrors the reference's
cutie/eval_vos.py:126-145,165-167 (time around processor.step only).  With N GPUs each rank runs its own clip
(clip sharding: weak scaling, no collective on the data path; the final gather of per-rank times is RCCL).

One JSON line is printed by rank 0.  Extra objects:
  roofline           dominant kernel family (implicit-GEMM conv on MFMA): algorithmic flops of the conv launches
                     of the timed frames / their device time, measured live with hipEvents on the launch stream
  roofline_affinity  the fused affinity read-out (similarity + top-k + softmax + V read-out) credited with the
                     reference's dense algorithmic flops (256+512K)*N*HW  
  cpu_baseline       the oracle (torch-fp32 restatement of the reference, "port") timed on this box's host cores
                     on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--objects', type=int, default=3)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=854)
    ap.add_argument('--no-long-term', action='store_true')
    ap.add_argument('--preroll', type=int, default=300,
                    help='untimed frames run before the warm-up so the memory bank is in its steady-state size')
    ap.add_argument('--cpu-frames', type=int, default=12, help='frames of the CPU baseline sample (0 = skip)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--full-bank-preroll', type=int, default=2000,
                    help='also report frames/s with the memory bank at its steady-state size: an extra clip run this many frames '
                         'before its timed steps (SURVEY C2: 18-26k tokens, pruning fires near frame 1995); 0 = skip')
    ap.add_argument('--no-lookahead', action='store_true',
                    help='do not pass next_image to step (no overlap of the next frame\'s image encoder on a side stream)')
    ap.add_argument('--window', type=int, default=int(os.environ.get('CUTIE_AMD_WINDOW', '12')),
                    help='frames per batched look-ahead encoder plan (step(next_images=...), InferenceCore.prefetch_window); '
                         '<= 1: one frame ahead (step(next_image=...))')
    ap.add_argument('--repeats', type=int, default=5,
                    help='the timed region of --steps frames is repeated this many times; "value" is the FIRST one (the protocol-conform '
                         'measurement), the spread goes to "repeats"')
    ap.add_argument('--clips-in-flight', type=int, default=4,
                    help='also measure the aggregate frames/s with this many independent clips in flight per GPU '
                         '(one HIP stream + CUTIE.fork() each, see --multi-mode; reported as "multi_clip"; 0 = skip)')
    ap.add_argument('--multi-hw-queues', type=int, default=0,
                    help='> 0 (one GPU): the clips-in-flight leg runs in a child process started with GPU_MAX_HW_QUEUES set to this (HIP maps '
                         'its streams onto 4 hardware queues by default).  An experiment of round 5 with --multi-mode threads '
                         '(profiles/r05_clips_in_flight.txt: no stable gain); 0 = in this process, default environment')
    ap.add_argument('--multi-mode', choices=('threads', 'interleaved', 'lockstep'), default='lockstep',
                    help='several clips per GPU: in LOCK STEP through one launch plan per stage (cutie_amd/inference/lockstep.py), or in flight on a '
                         'stream each, driven by ONE thread that issues a step of every clip in turn, or by one host thread per clip')
    ap.add_argument('--lockstep-groups', type=int, default=3, help='(--multi-mode lockstep) also measure this many lock-step groups of --clips-in-flight clips IN FLIGHT next to each '
                         'other (reported as multi_clip.groups_in_flight; with --multi-only: the leg itself); <= 1: skip')
    ap.add_argument('--multi-only', action='store_true', help='(internal) run the clips-in-flight leg only and print its seconds')
    ap.add_argument('--cpu-interpreter', action='store_true',
                    help='(tests only) no GPU: the launch plans run through the torch interpreter of the descriptors (tests/mock_exec.py) and the ranks '
                         'meet over gloo -- exercises the launcher / sharding / timing protocol of --gpus N on a CPU box, measures nothing')
    ap.add_argument('--device-index', type=int, default=None, help='(internal) GPU of a --multi-only child')
    return ap.parse_args()


def multi_clip_child(args, local):
    """The clips-in-flight leg in a child process of its own (one GPU, no process group): same workload arguments, GPU_MAX_HW_QUEUES
    set before the HIP runtime starts.  Returns (seconds, hardware queues) or raises."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'GROUP_RANK', 'ROLE_RANK', 'TORCHELASTIC_RUN_ID')}
    env['GPU_MAX_HW_QUEUES'] = str(args.multi_hw_queues)
    cmd = [sys.executable, os.path.abspath(__file__), '--multi-only', '--device-index', str(local), '--clips-in-flight', str(args.clips_in_flight),
           '--steps', str(args.steps), '--warmup', str(args.warmup), '--preroll', str(args.preroll), '--objects', str(args.objects),
           '--height', str(args.height), '--width', str(args.width), '--window', str(args.window), '--multi-mode', args.multi_mode]
    if args.no_long_term:
        cmd.append('--no-long-term')
    if args.no_lookahead:
        cmd.append('--no-lookahead')
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError('clips-in-flight child failed: ' + r.stderr[-800:])
    return float(json.loads(r.stdout.strip().splitlines()[-1])['seconds'])


def multi_clip_throughput(net, cfg, args, K, rank, dist, dev):
    """Same workload, C independent clips in flight on this GPU (cutie_amd/parallel.py:run_concurrent's scheme with a
    common start line): returns the seconds the slowest clip needed for multi_clip_steps(args) frames after the pre-roll."""
    import threading
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.utils.synth import SyntheticClip
    C, NF = args.clips_in_flight, 48
    steps = multi_clip_steps(args)
    data = []
    for c in range(C):
        clip = SyntheticClip(args.height, args.width, K, NF, seed=101 + 16 * rank + c)
        data.append((torch.stack([clip.frame(t) for t in range(NF)]).to(dev), clip.first_mask().to(dev), clip.objects))
    views = [net] + [net.fork() for _ in range(C - 1)]
    for v in views:
        v.engine().one_lane = True                          # one stream per clip, as cutie_amd/parallel.py:run_concurrent runs them
    try:
        return _multi_clip_threads(net, cfg, args, K, rank, dist, dev, views, data, steps)
    finally:
        for v in views:                                     # (every view, also when a worker or a barrier failed)
            v.engine().one_lane = False


def _multi_clip_threads(net, cfg, args, K, rank, dist, dev, views, data, steps):
    import threading
    from cutie_amd.inference.inference_core import InferenceCore
    C, NF = args.clips_in_flight, 48
    ready, start, done = threading.Barrier(C + 1), threading.Barrier(C + 1), threading.Barrier(C + 1)
    finish, errors = [0.0] * C, []

    def work(i):
        frames, mask, objs = data[i]
        stream = torch.cuda.Stream(device=dev)
        try:
            with torch.inference_mode(), torch.cuda.stream(stream):
                proc = InferenceCore(views[i], cfg=cfg)
                nxt = make_hint(args, frames, NF)
                proc.step(frames[0], mask, objects=objs, **nxt(0))
                for t in range(1, 1 + args.preroll + args.warmup):
                    proc.step(frames[t % NF], **nxt(t))
                torch.cuda.synchronize()
                ready.wait()                                # pre-roll done on this clip
                start.wait()                                # released once every rank is ready
                for t in range(steps):
                    tt = 1 + args.preroll + args.warmup + t
                    proc.step(frames[tt % NF], **nxt(tt))
                torch.cuda.synchronize()
                finish[i] = time.perf_counter()
        except BaseException as e:
            errors.append(e)
            ready.abort()
            start.abort()
        finally:
            try:
                done.wait()
            except threading.BrokenBarrierError:
                pass

    threads = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(C)]
    for t in threads:
        t.start()
    failed = False
    try:
        ready.wait()
    except threading.BrokenBarrierError:
        failed = True
    if dist is not None:
        dist.barrier()                                      # reached by every rank, also by one whose pre-roll failed
    def give_up():                                           # release the workers parked at `done`, then surface the error
        done.abort()
        for t in threads:
            t.join(timeout=60)
        raise errors[0]

    if failed:
        give_up()
    t0 = time.perf_counter()
    try:
        start.wait()
    except threading.BrokenBarrierError:
        give_up()
    try:
        done.wait()
    except threading.BrokenBarrierError:
        give_up()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return max(finish) - t0


def multi_clip_interleaved(net, cfg, args, K, rank, dist, dev):
    """Same workload, C clips in flight driven by THIS thread in turn (cutie_amd/parallel.py:run_interleaved's scheme): one stream, one
    net.fork() and one frame_context table per clip; returns the seconds for multi_clip_steps(args) frames of every clip."""
    from cutie_amd import frame_context
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.utils.synth import SyntheticClip
    C, NF = args.clips_in_flight, 48
    steps = multi_clip_steps(args)
    views = [net] + [net.fork() for _ in range(C - 1)]
    for v in views:
        v.engine().one_lane = True
    clips = []
    try:
        with torch.inference_mode():
            for c in range(C):
                clip = SyntheticClip(args.height, args.width, K, NF, seed=101 + 16 * rank + c)
                frames = torch.stack([clip.frame(t) for t in range(NF)]).to(dev)
                st, ctx = torch.cuda.Stream(device=dev), frame_context.new_context()
                with frame_context.context(ctx), torch.cuda.stream(st):
                    proc = InferenceCore(views[c], cfg=cfg)
                    nxt = make_hint(args, frames, NF)
                    proc.step(frames[0], clip.first_mask().to(dev), objects=clip.objects, **nxt(0))
                clips.append((proc, frames, nxt, st, ctx))

            def round_(t):
                for proc, frames, nxt, st, ctx in clips:
                    with frame_context.context(ctx), torch.cuda.stream(st):
                        proc.step(frames[t % NF], **nxt(t))

            for t in range(1, 1 + args.preroll + args.warmup):
                round_(t)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            for t in range(steps):
                round_(1 + args.preroll + args.warmup + t)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
    finally:
        for v in views:
            v.engine().one_lane = False


def multi_clip_lockstep(net, cfg, args, K, rank, dist, dev, rec=None):
    """Same workload, C clips advanced in LOCK STEP (cutie_amd/inference/lockstep.py): one launch plan per stage for the C x K objects of
    all clips, one joint encoder window, one memory bank per clip, one look-ahead read-out pass per bank version for the banks of all clips.  Returns (seconds for
    multi_clip_steps(args) frames of every clip, roofline object of the conv launches of the lock-step frames or None)."""
    from cutie_amd import ops as O
    from cutie_amd.inference.lockstep import LockstepCores
    from cutie_amd.utils.synth import SyntheticClip
    C, NF = args.clips_in_flight, 48
    steps = multi_clip_steps(args)
    with torch.inference_mode():
        clips = [SyntheticClip(args.height, args.width, K, NF, seed=101 + 16 * rank + c) for c in range(C)]
        frames = [torch.stack([cl.frame(t) for t in range(NF)]).to(dev) for cl in clips]
        views = []
        for fr in frames:
            v = [fr[i] for i in range(NF)]
            views.append(v + v[:32])
        depth = 0 if args.no_lookahead else 16
        hint = (lambda t: {}) if depth == 0 else (lambda t: {'next_images': [v[(t + 1) % NF:(t + 1) % NF + depth] for v in views]})
        ls = LockstepCores(net, cfg, C)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            ls.step([v[0] for v in views], [cl.first_mask().to(dev) for cl in clips], [cl.objects for cl in clips])
            for t in range(1, 1 + args.preroll + args.warmup):
                ls.step([v[t % NF] for v in views], **hint(t))
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            base = 1 + args.preroll + args.warmup
            for t in range(base, base + steps):
                ls.step([v[t % NF] for v in views], **hint(t))
            issue = time.perf_counter() - t0                # (the host has issued every launch; the device may still be running)
            torch.cuda.synchronize()
            secs = time.perf_counter() - t0
            assert ls.batched_steps == args.preroll + args.warmup + steps, 'every propagated frame of the leg must run as one plan per stage'
            roof = None
            if rec is not None and not args.no_roofline:
                # the conv launches of whole encoder batches / memory cycles of lock-step frames, replayed back to back (hipEvents)
                nrec = 30
                rec.rec, rec.on = [], True
                for t in range(base + steps, base + steps + nrec):
                    ls.step([v[t % NF] for v in views], **hint(t))
                rec.on = False
                torch.cuda.synchronize()
                allops = np.concatenate(rec.rec)
                convs = allops[allops['kind'] == O.CONV]
                conv_t = rec.ex.time_ops(convs, 3) * 1e-3
                enc = [a for a in rec.rec if (a['kind'] == O.STEM).any() and (a['kind'] == O.KEY_PREP).any()]      # the joint encoder windows
                enc_convs = np.concatenate([a[a['kind'] == O.CONV] for a in enc]) if enc else convs[:0]
                enc_t = rec.ex.time_ops(enc_convs, 3) * 1e-3 if len(enc_convs) else 0.0
                others = allops[(allops['kind'] != O.CONV)]
                oth_t = rec.ex.time_ops(others, 3) * 1e-3
                px = (-(-args.height // 16) * 16) * (-(-args.width // 16) * 16) / 414720.0
                alg_f = (60.2 + 56.1 * K + 44.7 * K / cfg.mem_every) * px * 1e9 * C      # SURVEY 8(d), per lock-step frame = C clip frames
                roof = {'bound': 'mfma', 'kernel': 'all conv launches of a lock-step frame (C clips: batch = C x K objects, joint encoder window)',
                        'achieved': round(alg_f / (conv_t / nrec) / 1e12, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(alg_f / (conv_t / nrec) / 1e12 / PEAK_BF16_TFLOPS, 4),
                        'executed_frac': round(conv_flops(convs) / conv_t / 1e12 / PEAK_BF16_TFLOPS, 4),
                        'ms_per_lockstep_frame': round(conv_t / nrec * 1e3, 3), 'ms_per_clip_frame': round(conv_t / nrec / C * 1e3, 3),
                        'launches_per_lockstep_frame': round(len(convs) / nrec, 1), 'launches_per_clip_frame': round(len(convs) / nrec / C, 1),
                        'all_launches_per_clip_frame': round(len(allops) / nrec / C, 1), 'recorded_lockstep_frames': nrec, 'traffic': None,
                        'encoder_convs_ms_per_lockstep_frame': round(enc_t / nrec * 1e3, 3), 'encoder_conv_launches_per_lockstep_frame': round(len(enc_convs) / nrec, 1),
                        'other_kernels_ms_per_lockstep_frame': round(oth_t / nrec * 1e3, 3),
                        'host_issue_ms_per_lockstep_frame': round(issue / steps * 1e3, 3), 'wall_ms_per_lockstep_frame': round(secs / steps * 1e3, 3)}
    return secs, roof


def multi_clip_lockstep_groups(net, cfg, args, K, rank, dist, dev):
    """--lockstep-groups G > 1: G lock-step groups of C clips each IN FLIGHT next to each other -- a stream and a CUTIE.fork() per group, this
    thread issues a lock-step frame of every group in turn (the two schemes of cutie_amd/parallel.py combined).  Returns the seconds for
    multi_clip_steps(args) frames of every clip (G x C clips)."""
    from cutie_amd import frame_context
    from cutie_amd.inference.lockstep import LockstepCores
    from cutie_amd.utils.synth import SyntheticClip
    C, G, NF = args.clips_in_flight, args.lockstep_groups, 48
    steps = multi_clip_steps(args)
    groups = []
    lanes = []
    try:
      with torch.inference_mode():
        for g in range(G):
            view = net if g == 0 else net.fork()
            lanes.append((view.engine(), view.engine().one_lane))
            view.engine().one_lane = True                   # a stream per GROUP: its look-ahead lanes in line (cutie_amd/parallel.py)
            clips = [SyntheticClip(args.height, args.width, K, NF, seed=101 + 16 * rank + g * C + c) for c in range(C)]
            views = []
            for cl in clips:
                fr = torch.stack([cl.frame(t) for t in range(NF)]).to(dev)
                v = [fr[i] for i in range(NF)]
                views.append(v + v[:32])
            st, ctx = torch.cuda.Stream(device=dev), frame_context.new_context()
            with frame_context.context(ctx), torch.cuda.stream(st):
                ls = LockstepCores(view, cfg, C)
                ls.step([v[0] for v in views], [cl.first_mask().to(dev) for cl in clips], [cl.objects for cl in clips])
            groups.append((ls, views, st, ctx))
        hint = lambda views, t: {} if args.no_lookahead else {'next_images': [v[(t + 1) % NF:(t + 1) % NF + 16] for v in views]}

        def round_(t):
            for ls, views, st, ctx in groups:
                with frame_context.context(ctx), torch.cuda.stream(st):
                    ls.step([v[t % NF] for v in views], **hint(views, t))
        for t in range(1, 1 + args.preroll + args.warmup):
            round_(t)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for t in range(steps):
            round_(1 + args.preroll + args.warmup + t)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    finally:
        for eng, was in lanes:
            eng.one_lane = was


def multi_leg(args):
    if args.multi_mode == 'lockstep' and args.lockstep_groups > 1:
        return multi_clip_lockstep_groups
    return {'lockstep': multi_clip_lockstep, 'interleaved': multi_clip_interleaved, 'threads': multi_clip_throughput}[args.multi_mode]


def multi_clip_steps(args):
    """Frames per clip of the multi-clip leg: at least 100 -- with the driver's 20 steps the leg measured the start-up of four host threads
    (690 frames/s next to 1070 for one clip, same box)."""
    return max(args.steps, 100)


def make_hint(args, frames, n):
    """t -> the look-ahead keyword arguments of step(frame t): what a video reader knows about the frames that follow."""
    if args.no_lookahead:
        return lambda t: {}
    if args.window > 1:
        from cutie_amd.inference import inference_core as IC
        depth = args.window + IC.WINDOW_LEAD + 2           # >= WINDOW + WINDOW_LEAD + 1 frames: a full batch can always be formed
        views = [frames[i] for i in range(n)]              # (what a reader holds anyway: no per-step tensor construction in the timed loop)
        views = views + views[:depth + 1]
        return lambda t: {'next_images': views[(t + 1) % n:(t + 1) % n + depth]}
    return lambda t: {'next_image': frames[(t + 1) % n]}


class Recorder:
    """Wraps the executor to capture the descriptor arrays of one frame (for the per-kernel timing)."""

    def __init__(self, ex):
        self.ex, self.rec, self.on = ex, [], False
        self.is_mock = ex.is_mock

    def run(self, arr):
        if self.on:
            self.rec.append(arr.copy())
        self.ex.run(arr)

    def run_cached(self, arr, cache, head=0):             # plans: replayed as HIP graphs by the executor, except while recording
        if self.on:
            return self.run(arr)
        self.ex.run_cached(arr, cache, head)

    def stream(self):
        return self.ex.stream()


def usable_cores():
    """Host cores this process may really use (affinity mask and cgroup quota), capped at 32 for the CPU baseline."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    env = os.environ.get('OMP_NUM_THREADS')
    if env and env.isdigit():
        n = min(n, int(env))
    return max(1, min(n, 32))


def conv_flops(arr):
    from cutie_amd import ops as O
    f = 0.0
    for r in arr:
        if r['kind'] == O.CONV:
            i = r['i']
            M = int(i[0]) * int(i[7]) * int(i[8])
            f += 2.0 * M * int(i[9]) * int(i[11]) * int(i[12]) * int(i[18])
    return f


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks ourselves -- the driver's own command line, one process per GPU
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>`) -- and hand its
    exit code back.  Rank 0 of that job prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # (dmabuf IPC: RCCL between the ranks of one node)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and not args.multi_only:
        sys.exit(relaunch_under_torchrun(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != max(1, args.gpus) and not args.multi_only:
        # (--gpus used to be parsed and dropped; a launcher that starts another number of ranks than the command line asks for is an error)
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; start it as `python bench.py --gpus N` or under '
                 f'`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`')
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    pinned = None
    on_cpu = args.cpu_interpreter
    if on_cpu:                                               # (tests: protocol only -- every leg that measures the device is off)
        args.no_roofline = args.no_graph = True
        args.clips_in_flight = args.cpu_frames = args.full_bank_preroll = 0
        args.repeats = 1
    if world > 1:
        # One rank per GPU, each with its own slice of the host cores: a rank's frame is ~0.8 ms of Python + launch calls, and N ranks
        # hopping over the same cores (or all landing on one NUMA node's first cores) is the one visible risk to clip-shard scaling.
        # LOCAL_WORLD_SIZE ranks share this host; rank r takes the r-th contiguous slice of the cores this job may use.
        try:
            cores = sorted(os.sched_getaffinity(0))
            lws = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
            per = len(cores) // lws
            if per >= 1:
                mine = cores[local * per:(local + 1) * per]
                os.sched_setaffinity(0, mine)
                torch.set_num_threads(max(1, min(per, 4)))
                pinned = [mine[0], mine[-1], len(mine)]
        except (AttributeError, OSError, ValueError):
            pinned = None
        import torch.distributed as dist
        if on_cpu:
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local))
        assert dist.get_world_size() == world
    if args.multi_only:
        local = args.device_index if args.device_index is not None else local
    if on_cpu:
        dev = torch.device('cpu')
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
    else:
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    sync = (lambda: None) if on_cpu else torch.cuda.synchronize

    from cutie_amd import _lib, ops as O
    from cutie_amd.config import default_config
    from cutie_amd.inference import inference_core as IC
    from cutie_amd.inference.inference_core import InferenceCore
    IC.WINDOW = args.window
    from cutie_amd.model.cutie import CUTIE
    from cutie_amd.utils.synth import SyntheticClip

    K = args.objects
    use_lt = not args.no_long_term
    cfg = default_config(use_long_term=use_lt)
    torch.manual_seed(0)
    net = CUTIE(cfg).to(dev).eval()
    from cutie_amd.utils.synth_weights import make_state_dict   # same deterministic synthetic weights as the parity tests
    sd = make_state_dict(seed=0)
    net.load_weights(sd)
    if args.multi_only:                                    # (the child of multi_clip_child: this leg and nothing else)
        secs = multi_leg(args)(net, cfg, args, K, rank, None, dev)
        secs = secs[0] if isinstance(secs, tuple) else secs
        print(json.dumps({'seconds': secs, 'steps_per_clip': multi_clip_steps(args), 'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES')}))
        return
    if on_cpu:
        from mock_exec import MockExecutor
        _lib.set_executor_for_testing(MockExecutor())
    rec = Recorder(_lib.get_executor())
    _lib.set_executor_for_testing(rec)                     # only a recording shim around the HIP executor

    clip = SyntheticClip(args.height, args.width, K, 128, seed=1 + rank)       # one clip per rank
    frames = torch.stack([clip.frame(t) for t in range(128)]).to(dev)           # resident in HBM (630 MB)
    mask = clip.first_mask().to(dev)
    proc = InferenceCore(net, cfg=cfg)
    import contextlib
    side = None if on_cpu else torch.cuda.Stream(device=dev)       # a real (capturable) stream, not the legacy null stream
    with torch.inference_mode(), (contextlib.nullcontext() if on_cpu else torch.cuda.stream(side)):
        la = make_hint(args, frames, 128)                  # the next frames, as a video reader knows them
        proc.step(frames[0], mask, objects=clip.objects, **la(0))
        t_idx = 1
        for _ in range(args.preroll):
            proc.step(frames[t_idx % 128], **la(t_idx))
            t_idx += 1
        sync()
        n_tok_start = sum(b.size() for b in proc.memory.buckets.values())
        from cutie_amd.parallel import timed_steps
        base = t_idx

        def one_step(i):
            proc.step(frames[(base + i) % 128], **la(base + i))

        # W warm-up steps, then exactly K timed steps between barrier + synchronize, MAX over the ranks (cutie_amd/parallel.py)
        rank_secs = []
        elapsed = timed_steps(one_step, args.steps, args.warmup, dev, per_rank=rank_secs)
        t_idx = base + args.warmup + args.steps
        # the same timed region again (no warm-up: the clip simply goes on).  EVERY region is W = 0 + exactly K steps between barrier +
        # synchronize, MAX over the ranks; "value" is the MEDIAN region (VERDICT r05 / ADVICE r04: a 20-step region holds one or two 12-frame
        # encoder batches, single regions scatter by 10 % with their phase; the first region is kept as value_first_region)
        regions = [(elapsed, list(rank_secs))]
        for _ in range(max(0, args.repeats - 1)):
            base = t_idx
            rs = []
            regions.append((timed_steps(one_step, args.steps, 0, dev, per_rank=rs), rs))
            t_idx = base + args.steps
        rep_vals = [round(world * args.steps / r[0], 2) for r in regions]
        med = sorted(range(len(regions)), key=lambda j: regions[j][0])[len(regions) // 2]
        first_elapsed, (elapsed, rank_secs) = elapsed, regions[med]
        n_tok_end = sum(b.size() for b in proc.memory.buckets.values())
        # ---- the same clip WITHOUT the next_image hint: what an unchanged scripting_demo.py / eval loop of the reference gets ----
        no_la = None
        hinted_protocol = None
        if not args.no_lookahead and not on_cpu:
            # the reference's own FPS protocol (cutie/eval_vos.py:126-145) for the HINTED caller as well
            ev_ms, t0p = 0.0, time.perf_counter()
            for i in range(args.steps):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                proc.step(frames[t_idx % 128], **la(t_idx))
                e1.record()
                torch.cuda.synchronize()
                ev_ms += e0.elapsed_time(e1)
                t_idx += 1
            wallp = time.perf_counter() - t0p
            hinted_protocol = {'fps': round(args.steps / ev_ms * 1e3, 2), 'wall_fps': round(args.steps / wallp, 2), 'steps': args.steps,
                               'note': 'step(image, next_images=...) under the reference\'s FPS protocol: synchronize, event, step, event, synchronize; fps = frames / sum of '
                                       'the event intervals ON THE CALLER\'S STREAM -- the look-ahead lanes (window encoder, stacked read-outs) run on other streams and are '
                                       'only caught by the synchronize behind the second event, so for this multi-stream caller wall_fps (the two host synchronisations per '
                                       'frame included) is the honest one of the two'}
        if not args.no_lookahead and not on_cpu:
            base2 = t_idx
            # un-timed steps first, enough of them to use up every frame the hinted steps before have already encoded ahead (a look-ahead
            # window and its lead): with 2 of them a 20-step region still ran on left-over window entries (1023 frames/s instead of ~820)
            drain = 2 if args.window <= 1 else args.window + IC.WINDOW_LEAD + 2
            t_nola = timed_steps(lambda i: proc.step(frames[(base2 + i) % 128]), args.steps, drain, dev)
            t_idx = base2 + drain + args.steps
            assert not getattr(proc, '_window', None), 'the un-hinted region must not find frames encoded ahead'
            no_la = {'value': round(world * args.steps / t_nola, 2), 'ms_per_step': round(t_nola / args.steps * 1e3, 4)}
            # the reference's own FPS protocol (cutie/eval_vos.py:126-145): synchronize, start event, step, end event, synchronize, and
            # FPS = frames / the sum of the event intervals -- the host never runs ahead of the device, every frame starts on an idle GPU
            ev_ms, t0 = 0.0, time.perf_counter()
            for i in range(args.steps):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                proc.step(frames[t_idx % 128])
                e1.record()
                torch.cuda.synchronize()
                ev_ms += e0.elapsed_time(e1)
                t_idx += 1
            wall = time.perf_counter() - t0
            no_la['eval_vos_protocol'] = {'fps': round(args.steps / ev_ms * 1e3, 2), 'wall_fps': round(args.steps / wall, 2), 'steps': args.steps,
                                          'note': 'per frame: synchronize, event, step(image), event, synchronize (cutie/eval_vos.py:126-145); fps = frames / sum of the '
                                                  'event intervals (what the reference logs as FPS), wall_fps includes the two host synchronisations per frame'}
        # ---- full-bank point: long-term memory at its steady-state size ----
        full_bank = None
        if args.full_bank_preroll > 0 and use_lt:
            proc_fb = InferenceCore(net, cfg=cfg)
            proc_fb.step(frames[0], mask, objects=clip.objects, **la(0))
            for t in range(1, args.full_bank_preroll):
                proc_fb.step(frames[t % 128], **la(t))
            torch.cuda.synchronize()
            fb0 = args.full_bank_preroll
            nfb = min(args.steps, 200)
            t_fb = timed_steps(lambda i: proc_fb.step(frames[(fb0 + i) % 128], **la(fb0 + i)), nfb, 5, dev)
            full_bank = {'preroll_frames': args.full_bank_preroll, 'memory_tokens': sum(b.size() for b in proc_fb.memory.buckets.values()),
                         'value': round(world * nfb / t_fb, 2), 'ms_per_step': round(t_fb / nfb * 1e3, 4), 'steps': nfb}
            del proc_fb

        # ---- per-kernel-family timing on the launch stream (hipEvents inside cutie_time_ops) ----
        roof = roof_aff = None
        if not args.no_roofline and rank == 0:
            conv_t = conv_f = aff_t = aff_f = 0.0
            # nrec frames of the timed workload, driven exactly like the timed region (same hints): with the look-ahead window 5 batched
            # encoder plans and `window` memory frames -- their conv launches are replayed back to back and divided by the frames
            nrec = 20 if args.window <= 1 or args.no_lookahead else 5 * args.window      # whole encoder batches and whole memory cycles
            rec.rec, rec.on = [], True
            for _ in range(nrec):
                proc.step(frames[t_idx % 128], **la(t_idx))
                t_idx += 1
            rec.on = False
            torch.cuda.synchronize()
            allops = np.concatenate(rec.rec)
            convs = allops[allops['kind'] == O.CONV]
            # the affinity plan = [score/0, select (+ counter clear, usage ticks), score/1, readout]; since round 5 one plan serves every
            # announced frame up to the next memory frame (one read-out per bank version): frames of a plan = stacked rows / rows per frame
            aff_arrs = [a[a['kind'] != O.USAGE_TICK] for a in rec.rec if (a['kind'] == O.AFF_SCORE).any()]
            affs_all = np.concatenate(aff_arrs)

            def plan_frames(a):
                ii = a[a['kind'] == O.AFF_SCORE]['i'][0]
                return int(ii[1]) // int(ii[16]) if int(ii[16]) > 0 else 1

            def plan_tokens(a):
                ii = a[a['kind'] == O.AFF_SCORE]['i'][0]
                return sum(int(ii[4 + 2 * r]) for r in range(int(ii[2])))
            conv_t += rec.ex.time_ops(convs, 3) * 1e-3
            conv_f += conv_flops(convs)
            aff_t += rec.ex.time_ops(affs_all, 3) * 1e-3
            HW = (proc.memory.H * proc.memory.W)
            kb = {len(b.objects) for b in proc.memory.buckets.values()}
            assert len(proc.memory.buckets) == 1 and len(kb) == 1, 'the bench clip has one bucket'
            frames_read = sum(plan_frames(a) for a in aff_arrs)
            for a in aff_arrs:                             # the reference's dense flops of every frame a plan reads, at that plan's bank size
                aff_f += (256 + 512 * K) * float(plan_tokens(a)) * HW * plan_frames(a)
            n_conv = round(int((allops['kind'] == O.CONV).sum()) / nrec, 1)
            # the affinity matmul on its own (score pass 0 = the S = A.B^T tiles + tile maxima; valid in isolation): ALL pass-0 launches of
            # the recorded frames replayed back to back (mfma_util = their issued split-bf16 flops / their time), and the stages of the
            # most common plan by prefix differences of [score/0, select (+ counter clear, usage ticks), score/1, readout]
            aff_parts = None
            pass0 = np.concatenate([a[(a['kind'] == O.AFF_SCORE) & (a['i'][:, 11] == 0)] for a in aff_arrs])
            if len(pass0):
                issued = sum(3 * 2.0 * 128 * (int(r['i'][9]) * 16) * int(r['i'][1]) for r in pass0)     # 3 split-bf16 terms, padded tiles x stacked rows
                dense32 = sum(2.0 * 128 * sum(int(r['i'][4 + 2 * q]) for q in range(int(r['i'][2]))) * int(r['i'][0]) *
                              (int(r['i'][1]) // int(r['i'][16]) if int(r['i'][16]) > 0 else 1) for r in pass0)
                t_reps = [rec.ex.time_ops(pass0, 5) * 1e-3 for _ in range(3)]
                t_mm = sum(t_reps) / len(t_reps)             # MEAN of the replays (the minimum flattered the figure)
                real = sum(3 * 2.0 * 128 * sum(int(r['i'][4 + 2 * q]) for q in range(int(r['i'][2]))) * int(r['i'][0]) *
                           (int(r['i'][1]) // int(r['i'][16]) if int(r['i'][16]) > 0 else 1) for r in pass0)     # split-bf16 flops of the REAL tokens x real query rows
                by_f = {}
                for a in aff_arrs:
                    by_f.setdefault(plan_frames(a), []).append(a)
                common_f = max(by_f, key=lambda f_: len(by_f[f_]) * f_)
                affs = by_f[common_f][-1]
                s0 = next(k for k in range(len(affs)) if int(affs['kind'][k]) == O.AFF_SCORE)
                pre = [0.0] * (1 - s0) + [min(rec.ex.time_ops(affs[:k], 10) for _ in range(3)) * 1e3 for k in range(1, s0 + 5)]
                ii = affs[s0]['i']
                aff_parts = {'kernel': 'aff_score_kernel mode 0 (S = A.B^T on v_mfma_f32_16x16x32_bf16, 3 split terms)',
                             'launches': int(len(pass0)), 'frames_read': frames_read, 'recorded_frames': nrec,
                             'frames_per_launch': {str(f_): len(v) for f_, v in sorted(by_f.items())},
                             'us': round(t_mm * 1e6 / len(pass0), 2), 'us_per_frame': round(t_mm * 1e6 / frames_read, 2),
                             'mfma_issued_tflops': round(issued / t_mm / 1e12, 1),
                             'mfma_util': round(issued / t_mm / 1e12 / PEAK_BF16_TFLOPS, 4),
                             'mfma_util_unpadded': round(real / t_mm / 1e12 / PEAK_BF16_TFLOPS, 4),
                             'mfma_util_best_replay': round(issued / min(t_reps) / 1e12 / PEAK_BF16_TFLOPS, 4),
                             'mfma_util_note': 'issued split-bf16 MFMA flops of ALL score-pass launches of the recorded frames (padded 16-token tiles x padded query '
                                               'rows) / their device time (MEAN of three back-to-back replays, hipEvents) / 2.5 PFLOP/s; _unpadded prices the real '
                                               'tokens x real query rows only',
                             'fp32_equivalent_tflops': round(dense32 / t_mm / 1e12, 1),
                             'stage_plan': {'frames': common_f, 'tokens': plan_tokens(affs), 'queries': int(ii[0]) * common_f},
                             'stage_us': {'memset': round(pre[0], 1), 'score0': round(pre[1] - pre[0], 1),
                                          'select': round(pre[2] - pre[1], 1), 'score1': round(pre[3] - pre[2], 1),
                                          'readout': round(pre[4] - pre[3], 1)},
                             'stage_us_per_frame': {'score0': round((pre[1] - pre[0]) / common_f, 1), 'select': round((pre[2] - pre[1]) / common_f, 1),
                                                    'score1': round((pre[3] - pre[2]) / common_f, 1), 'readout': round((pre[4] - pre[3]) / common_f, 1)}}
            # device-time breakdown of the last recorded frame by op kind (back-to-back replays, hipEvents), and the
            # whole frame replayed as ONE HIP graph (no host involvement): shows how much of the step is launch gaps
            names = O.KIND_NAMES
            breakdown = {}
            aff_kinds = (O.AFF_SCORE, O.AFF_SELECT, O.AFF_READOUT)
            # [launches, device us] PER FRAME, averaged over the nrec recorded frames (back-to-back replays of each kind)
            if not args.no_breakdown:
                breakdown['AFFINITY(score x2+select+readout)'] = [round(len(affs_all) / nrec, 2), round(rec.ex.time_ops(affs_all, 3) * 1e3 / nrec, 1)]
            for kind in ([] if args.no_breakdown else sorted(set(int(k) for k in allops['kind']) - set(aff_kinds))):
                sel = allops[allops['kind'] == kind]
                if os.environ.get('BENCH_DEBUG'):
                    print('replay kind', kind, names.get(kind), len(sel), file=sys.stderr, flush=True)
                breakdown[names.get(kind, str(kind))] = [round(len(sel) / nrec, 2), round(rec.ex.time_ops(sel, 3) * 1e3 / nrec, 1)]
            lib = rec.ex.lib
            # one frame without hints (all of its launches in one list, its own image encoder included) as ONE HIP graph: un-hinted
            # steps first until nothing encoded ahead by the hinted frames above is left (round 4 recorded a frame that found its encoder
            # output in the look-ahead window: 0.733 ms for a graph without an encoder)
            for _ in range(2 if args.window <= 1 or args.no_lookahead else args.window + IC.WINDOW_LEAD + 2):
                proc.step(frames[t_idx % 128])
                t_idx += 1
            assert not getattr(proc, '_window', None) and proc._prefetched is None, 'the graph frame must run its own encoder'
            while (proc.curr_ti + 1 - proc.last_mem_ti) >= proc.mem_every:     # record a plain (non-memory) frame
                proc.step(frames[t_idx % 128])
                t_idx += 1
            rec.rec, rec.on = [], True
            proc.step(frames[t_idx % 128])
            t_idx += 1
            rec.on = False
            torch.cuda.synchronize()
            oneframe = np.concatenate(rec.rec)
            assert int((oneframe['kind'] == O.STEM).sum()) >= 1, 'the graph frame has no image encoder'
            graph_launches = int(len(oneframe))
            g = None if args.no_graph else lib.cutie_graph_capture(oneframe.ctypes.data, len(oneframe), rec.ex.stream())
            graph_ms = None
            if g:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    lib.cutie_graph_launch(g, rec.ex.stream())
                torch.cuda.synchronize()
                graph_ms = (time.perf_counter() - t1) / 20 * 1e3
                lib.cutie_graph_destroy(g)
            # HBM traffic of the same kernel family from the committed rocprofv3 PMC passes (tools/profile_round.sh: FETCH_SIZE x2
            # per the gfx950 correction + WRITE_SIZE, bytes per launch); null when no profile summary is present
            traffic = traffic_src = None
            headline = (args.height, args.width, K, use_lt) == (480, 854, 3, True)     # the committed PMC passes are of the headline workload only
            try:
                import glob
                summ = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_summary.json')))
                if summ and headline:
                    sj = json.load(open(summ[-1]))
                    pmc = sj['pmc']
                    traffic_src = {'file': 'profiles/' + os.path.basename(summ[-1]), 'tree': sj.get('tree', 'unknown (summary predates the field)')}
                    fam = [v for k, v in pmc.items() if k.startswith(('conv_igemm_kernel', 'conv_dma_kernel', 'conv_pc_kernel', 'conv_cout1'))
                           and 'fetch_MB_per_dispatch_x2_gfx950_corrected' in v and 'write_MB_per_dispatch' in v]
                    nd = sum(v['dispatches'] for v in fam)
                    traffic = round(sum((v['fetch_MB_per_dispatch_x2_gfx950_corrected'] + v['write_MB_per_dispatch']) * v['dispatches']
                                        for v in fam) / nd * 1e6) if nd else None
            except Exception:
                traffic = None
            # algorithmic flops per frame: SURVEY.md section 8(d) (60.2 G shared + 56.1 G x K per frame + 44.7 G x K per memory frame, at
            # 414 720 padded pixels); the executed count is ~5 % higher (the composed [pixel | x] projections of the transformer blocks,
            # K = 512) -- `frac` is priced with the ALGORITHMIC flops
            px = (-(-args.height // 16) * 16) * (-(-args.width // 16) * 16) / 414720.0
            alg_f = (60.2 + 56.1 * K + 44.7 * K / cfg.mem_every) * px * 1e9
            ms_conv = conv_t / nrec * 1e3
            roof = {'bound': 'mfma', 'kernel': 'conv_pc_kernel<*> + conv_dma_kernel<*> + conv_igemm_kernel<*> + conv_cout1 (all conv launches of a frame)',
                    'achieved': round(alg_f / (conv_t / nrec) / 1e12, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(alg_f / (conv_t / nrec) / 1e12 / PEAK_BF16_TFLOPS, 4),
                    'algorithmic_gflop_per_frame': round(alg_f / 1e9, 1),
                    'executed_frac': round(conv_f / conv_t / 1e12 / PEAK_BF16_TFLOPS, 4),
                    'hbm_frac': None if traffic is None else round(traffic * n_conv / (conv_t / nrec) / 1e9 / PEAK_HBM_GBS, 4),
                    'traffic': traffic,
                    'traffic_unit': 'HBM bytes per conv launch (rocprofv3 PMC passes of tools/profile_round.sh, measured on the tree named in traffic_source)',
                    'traffic_source': traffic_src,
                    'gflop_per_frame': round(conv_f / nrec / 1e9, 1), 'ms_per_frame': round(conv_t / nrec * 1e3, 3),
                    'launches_per_frame': n_conv}
            roof_aff = {'bound': 'mfma', 'kernel': 'aff_score x2 + aff_select + aff_readout',
                        'achieved': round(aff_f / aff_t / 1e12, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(aff_f / aff_t / 1e12 / PEAK_BF16_TFLOPS, 4), 'traffic': None,
                        'algorithmic_gflop_per_frame': round(aff_f / nrec / 1e9, 1), 'ms_per_frame': round(aff_t / nrec * 1e3, 4),
                        'launches_per_frame': round(len(affs_all) / nrec, 2), 'frames_read_per_recorded_frame': round(frames_read / nrec, 2),
                        'memory_tokens': n_tok_end, 'matmul': aff_parts,
                        'note': 'algorithmic = dense (256+512K)*N*HW of the reference; the kernels do the top-k readout sparsely'}

    tmax = elapsed                                           # already the MAX over the ranks

    multi = None
    if args.clips_in_flight > 1:
        # an extra leg: a failure here is reported in the line, it must not take the headline measurement with it
        in_child = world == 1 and args.multi_hw_queues > 0
        multi_roof = None
        try:
            if in_child:
                leg = multi_clip_child(args, local)
            elif args.multi_mode == 'lockstep':
                leg, multi_roof = multi_clip_lockstep(net, cfg, args, K, rank, dist, dev, rec if rank == 0 else None)
            else:
                leg = multi_leg(args)(net, cfg, args, K, rank, dist, dev)
            leg_err = None
        except Exception as e:
            import traceback
            traceback.print_exc()
            leg, leg_err = 0.0, f'{type(e).__name__}: {e}'
        mt = torch.tensor([leg, 0.0 if leg_err is None else 1.0], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(mt, op=dist.ReduceOp.MAX)
        if float(mt[1].item()) > 0:
            multi = {'clips_in_flight_per_gpu': args.clips_in_flight, 'error': leg_err or 'failed on another rank'}
        else:
            mt = float(mt[0].item())
            msteps = multi_clip_steps(args)
            multi = {'clips_in_flight_per_gpu': args.clips_in_flight, 'value': round(world * args.clips_in_flight * msteps / mt, 2),
                     'unit': 'frames/s', 'steps_per_clip': msteps, 'ms_per_step': round(mt / msteps * 1e3, 4),
                     'hw_queues': (args.multi_hw_queues if in_child else os.environ.get('GPU_MAX_HW_QUEUES', 'default (4)')),
                     'mode': args.multi_mode,
                     'host': {'lockstep': 'ONE launch plan per stage for the C x K objects of all clips (cutie_amd/inference/lockstep.py: joint encoder window, one bank per clip, '
                                          'one look-ahead read-out pass over the banks of all clips); per clip bit-identical to its own InferenceCore run',
                              'interleaved': 'one thread issues a step of every clip in turn (cutie_amd/parallel.py:run_interleaved)',
                              'threads': 'one host thread per clip (cutie_amd/parallel.py:run_concurrent)'}[args.multi_mode],
                     'note': ('same workload, C independent clips per GPU advanced in lock step' if args.multi_mode == 'lockstep' else
                              'same workload, independent clips in flight on one GPU (ONE HIP stream + CUTIE.fork() per clip' + ('; measured in a child process started with GPU_MAX_HW_QUEUES=%d: '
                              'HIP maps streams onto 4 hardware queues by default and clips that share one serialise' % args.multi_hw_queues if in_child else '') + ')')
                             + '; "value" above stays one clip per GPU, default environment'}
            if multi_roof is not None:
                multi['roofline'] = multi_roof
            if args.multi_mode == 'lockstep' and args.lockstep_groups > 1 and world == 1 and not in_child:
                try:
                    tg = multi_clip_lockstep_groups(net, cfg, args, K, rank, None, dev)
                    ncl = args.lockstep_groups * args.clips_in_flight
                    multi['groups_in_flight'] = {'groups': args.lockstep_groups, 'clips_per_group': args.clips_in_flight, 'clips': ncl,
                                                 'value': round(ncl * msteps / tg, 2), 'unit': 'frames/s',
                                                 'note': 'lock-step groups in flight next to each other (a stream + CUTIE.fork() per group, one thread issues a lock-step '
                                                         'frame of every group in turn: cutie_amd/parallel.py run_batched(in_flight=...))'}
                except Exception as e:
                    multi['groups_in_flight'] = {'error': f'{type(e).__name__}: {e}'}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle.inference import OracleProcessor, DEFAULT_CFG
        from oracle.net import OracleNet
        ncores = usable_cores()
        torch.set_num_threads(ncores)
        onet = OracleNet(sd)
        ocfg = dict(DEFAULT_CFG)
        ocfg['use_long_term'] = use_lt
        oproc = OracleProcessor(onet, ocfg)
        # bank of the same size class as the timed device region (which starts after `preroll` frames, ~7 memory frames in the bank):
        # memorise every frame for n_fill frames (mem_every = 1), then back to the configured cadence for the timed sample
        n_fill = max(0, min(9, round(n_tok_start / max(1, (args.height // 16) * (-(-args.width // 16)))) - 1)) if use_lt else 4
        with torch.inference_mode():
            oproc.step(clip.frame(0), clip.first_mask(), objects=clip.objects)
            keep_every, oproc.mem_every = oproc.mem_every, 1
            for tt in range(1, 1 + n_fill):
                oproc.step(clip.frame(tt))
            oproc.mem_every = keep_every
            oproc.step(clip.frame(1 + n_fill))             # warm-up at the full bank
            c0 = time.perf_counter()
            n_cpu = 0
            for tt in range(2 + n_fill, 2 + n_fill + args.cpu_frames):
                oproc.step(clip.frame(tt))
                n_cpu += 1
                if time.perf_counter() - c0 > 30.0:            # bounded sample
                    break
            cpu_t = time.perf_counter() - c0
            args.cpu_frames = n_cpu
        ratio = None
        try:
            ratio = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'cpu_reference_ratio.json')))
        except Exception:
            pass
        cpu = {'value': round(args.cpu_frames / cpu_t, 3), 'unit': 'frames/s', 'cores': torch.get_num_threads(),
               'kind': 'port',
               'port_over_reference': None if ratio is None else ratio['oracle_over_reference'],
               'port_over_reference_note': None if ratio is None else
               f"the live reference (/root/reference) and this port timed on the same {ratio['cores']} cores of the build container, same workload "
               f"(oracle/time_reference.py): reference {ratio['reference_fps']} frames/s, port {ratio['oracle_fps']} frames/s",
               'sample': f'{args.cpu_frames} propagated frames of the same {args.width}x{args.height} {K}-object clip, oracle (torch fp32 '
                         f'restatement of the reference) on host cores, timed with {1 + n_fill} memory frames in the bank '
                         f'({(1 + n_fill) * (args.height // 16) * (-(-args.width // 16))} tokens: the size class of the device run, '
                         f'{n_tok_start} tokens at the start of its timed region; filled by memorising every frame, cadence {keep_every} while timed)'}

    if rank == 0:
        fps = world * args.steps / tmax
        out = {
            'metric': 'frames/sec (480p, 3 objects)', 'value': round(fps, 2), 'unit': 'frames/s',
            'n_gpus': (dist.get_world_size() if dist is not None else 1),      # the ranks the process group really has (one GPU each)
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(tmax / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'synthetic {args.width}x{args.height} {K}-object clip, long_term={use_lt}, one clip per GPU '
                                   f'(SURVEY 8d C2/C3), eval_config defaults (mem_every=5, top_k=30), random-init weights',
                       'preroll_frames': args.preroll, 'memory_tokens_start': n_tok_start, 'memory_tokens_end': n_tok_end,
                       'parallelism': f'clip-shard x{world}', 'accumulate': 'fp32',
                       'host_cores_of_rank0': pinned,
                       'lookahead': 'off' if args.no_lookahead else
                                    (f'step(next_images=...): the image encoder runs over a window of {args.window} upcoming frames as one batched plan on a '
                                     'third stream (tiles of the same K-order class: bit-identical features)' if args.window > 1 else
                                     'step(next_image=...): the next frame\'s image encoder runs on a side stream') +
                                    '; when the current frame does not write the memory bank, the next frame\'s affinity read-out runs ahead on a side stream'},
        }
        gs = getattr(rec.ex, 'graph_stats', None)
        if gs is not None:
            out['plans_eager_vs_graph_replay'] = list(gs)      # whole run (pre-roll included): plans issued launch by launch | as one HIP graph
        rv = sorted(rep_vals)
        out['value_first_region'] = round(world * args.steps / first_elapsed, 2)
        out['per_rank_fps'] = [round(args.steps / s_, 2) for s_ in rank_secs]      # every rank's own clip in the region "value" is taken from
        out['repeats'] = {'values': rep_vals, 'median': rv[len(rv) // 2], 'min': rv[0], 'max': rv[-1],
                          'mean_fps_all_regions': round(len(rep_vals) / sum(1.0 / v for v in rep_vals), 2),
                          'note': f'{len(rep_vals)} consecutive timed regions on this box, each EXACTLY {args.steps} steps between barrier + synchronize (the first one behind '
                                  f'the {args.warmup} warm-up steps); "value" = the MEDIAN region, value_first_region = the first.  A region of {args.steps} frames holds '
                                  f'{args.steps // max(1, args.window)}-{-(-args.steps // max(1, args.window)) + 1} batched encoder plans of {args.window} frames each, so short regions scatter with the phase of the batches; '
                                  'mean_fps_all_regions = all frames / all time'}
        if hinted_protocol is not None:
            out['eval_vos_protocol'] = hinted_protocol
        if no_la is not None:
            out['value_no_lookahead'] = no_la['value']
            out['no_lookahead'] = dict(no_la, note='same clip, step(image) without the next_image hint (an unchanged scripting_demo.py)')
        if full_bank is not None:
            out['full_bank'] = full_bank
        if roof is not None:
            out['roofline'] = roof
            out['roofline_affinity'] = roof_aff
            out['device_us_by_kind'] = breakdown
            out['frame_as_one_hip_graph_ms'] = None if graph_ms is None else round(graph_ms, 3)
            out['frame_as_one_hip_graph_launches'] = graph_launches
        if multi is not None:
            out['multi_clip'] = multi
        if cpu is not None:
            out['cpu_baseline'] = cpu
            out['speedup_vs_cpu'] = round(fps / cpu['value'], 1)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
