"""Clip sharding over the GPUs of one node.

The per-frame state of a clip is a strict recurrence in time, so a clip never spans GPUs; independent clips are
distributed round-robin, one process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" in the CPU tests).  There is no collective on the per-frame path: the only communication is the final
gather of per-clip results to rank 0 -- a direct (non-ring) gather, since the payload is small and latency-bound.

Inside one GPU, several clips can be *in flight* at once: ONE HIP stream + ``CUTIE.fork()`` per clip.  A single clip is a chain of
dependent small launches per frame and leaves compute units idle (kernel-boundary bubbles, layers with < 256 workgroups); independent
chains interleave on the hardware queues.  Clips of one geometry and object count can instead advance in LOCK STEP through one launch plan per
stage (``run_batched``, cutie_amd/inference/lockstep.py): fewer, larger launches -- the better of the two on the MI355X (round 6).  A clip in flight keeps the batching of its look-ahead lanes (one encoder plan per 12 frames,
one read-out per bank version) but runs them on its own stream (``Engine.one_lane``): next to other clips its extra streams only
compete for the hardware queues.  Two drivers:
* ``run_interleaved`` -- ONE host thread issues a step of every clip in turn (generator clips).  ``step`` never waits for the device,
  so one thread keeps four streams fed: **1811 / 1826 / 1820 frames/s with four clips in flight against 1163 for one clip** (MI355X,
  480p / 3 objects, round 5, `profiles/r05_clips_in_flight.txt`; 2 clips 1525, 3: 1718, 8: 1769) -- bound by the host's issue time
  per frame (~0.55 ms);
* ``run_concurrent`` -- one host thread per clip (clips that also read / write files).  Two threads: 1525-1565; three or four threads
  hand the interpreter lock around at every launch call and vary between 770 and 1865 from run to run of one command, hence its
  default of two.
"""
import queue
import threading
import time
from typing import Callable, Dict, List, Sequence

import torch
import torch.distributed as dist


def shard_clips(num_clips: int, rank: int, world_size: int) -> List[int]:
    """Clip c runs on rank c mod world_size (the reference creates one InferenceCore per video, eval_vos.py:97)."""
    return [c for c in range(num_clips) if c % world_size == rank]


def timed_steps(step: Callable[[int], None], steps: int, warmup: int, device, per_rank: list = None) -> float:
    """The timing protocol of bench.py (one clip per rank, weak scaling): `warmup` untimed calls of step(i), then EXACTLY `steps`
    timed calls bracketed by a barrier + device synchronisation on both sides; returns the MAX over the ranks of the elapsed
    seconds (all-reduce: RCCL on the GPUs, gloo in the CPU tests) -- the whole job is as slow as its slowest rank.
    per_rank (a list): receives every rank's own elapsed seconds, in rank order (all-gather)."""
    dev = torch.device(device)
    on_gpu = dev.type == 'cuda'
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def fence():
        if on_gpu:
            torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(dev)

    for i in range(warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    if on_gpu:
        torch.cuda.synchronize(dev)
    own = time.perf_counter() - t0                        # this rank's clip alone (before it waits for the others)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if on_gpu else 'cpu')
    if per_rank is not None:
        if multi:
            mine = torch.tensor([own], dtype=torch.float64, device=dev if on_gpu else 'cpu')
            parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(parts, mine)
            per_rank[:] = [float(p.item()) for p in parts]
        else:
            per_rank[:] = [own]
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_concurrent(net, clip_ids: Sequence[int], run_clip: Callable, *, streams: int = 2) -> Dict[int, Dict]:
    """Run ``run_clip(net_view, clip_id) -> result`` for every clip with up to ``streams`` clips in flight on the GPU of
    ``net``.  Every worker owns a host thread, a HIP stream and a ``net.fork()`` (shared weights, private launch plans and
    activation buffers); clips are pulled from a common queue.  The first clip runs alone when the conv tiles are not
    tuned yet, so the autotuner times kernels on a quiet device.  Results are bit-identical to running the clips one
    after another (every kernel is deterministic and no state is shared between views)."""
    clip_ids = list(clip_ids)
    results: Dict[int, Dict] = {}
    if not clip_ids:
        return results
    on_gpu = net.device.type == 'cuda'
    n = max(1, min(streams, len(clip_ids)))
    errors: List[BaseException] = []
    jobs: 'queue.Queue' = queue.Queue()
    for c in clip_ids:
        jobs.put(c)
    tuned = threading.Event()
    if net.engine().tile_cache or not on_gpu:
        tuned.set()

    def work(view, first):
        stream = torch.cuda.Stream(device=net.device) if on_gpu else None
        try:
            with torch.inference_mode():
                while True:
                    if not first:
                        tuned.wait()
                    try:
                        c = jobs.get_nowait()
                    except queue.Empty:
                        return
                    if stream is not None:
                        with torch.cuda.stream(stream):
                            r = run_clip(view, c)
                        stream.synchronize()
                    else:
                        r = run_clip(view, c)
                    results[c] = r
                    tuned.set()
                    first = False
        except BaseException as e:          # surfaced in the caller's thread
            errors.append(e)
            tuned.set()

    views = [net] + [net.fork() for _ in range(n - 1)]
    # One stream per clip (see the module docstring): the other clips are what runs next to a clip's chain, and n x 4 streams only share
    # the device's hardware queues (HIP maps streams onto $GPU_MAX_HW_QUEUES = 4 of them by default).
    lanes = [(v.engine(), v.engine().one_lane) for v in views] if n > 1 else []
    for eng, _ in lanes:
        eng.one_lane = True
    try:
        threads = [threading.Thread(target=work, args=(views[i], i == 0), daemon=True) for i in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        for eng, was in lanes:
            eng.one_lane = was
    if errors:
        raise errors[0]
    return results


def run_interleaved(net, clip_ids: Sequence[int], run_clip: Callable, *, streams: int = 4) -> Dict[int, Dict]:
    """Like ``run_concurrent`` with ONE host thread: ``run_clip(net_view, clip_id)`` is a GENERATOR function that yields after every
    ``step`` and returns the clip's result; up to ``streams`` clips are in flight, each on its own HIP stream and ``net.fork()``, and the
    calling thread issues one step of each in turn.  ``step`` never waits for the device, so one thread keeps several streams fed; there
    is no interpreter lock to hand around (three or four Python threads do that at every launch call, and their throughput varies by
    2 x from run to run: profiles/r05_clips_in_flight.txt).  Bound by the host's issue time per frame.  Results are bit-identical to
    running the clips one after another.  A generator that ends should have fetched what it returns (``.cpu()``): its slot is
    re-used by the next clip at once."""
    import contextlib
    from . import frame_context
    clip_ids = list(clip_ids)
    results: Dict[int, Dict] = {}
    if not clip_ids:
        return results
    on_gpu = net.device.type == 'cuda'
    n = max(1, min(streams, len(clip_ids)))
    views = [net] + [net.fork() for _ in range(n - 1)]
    lanes = [(v.engine(), v.engine().one_lane) for v in views] if n > 1 else []
    for eng, _ in lanes:
        eng.one_lane = True
    pending = list(reversed(clip_ids))
    slots = [dict(view=v, stream=torch.cuda.Stream(device=net.device) if on_gpu else None, ctx=frame_context.new_context(), gen=None, clip=None)
             for v in views]

    def enter(slot):
        return torch.cuda.stream(slot['stream']) if slot['stream'] is not None else contextlib.nullcontext()

    def advance(slot):
        """One step of the slot's clip (starting the next clip of the queue when the slot is free); False when there is nothing left."""
        if slot['gen'] is None:
            if not pending:
                return False
            slot['clip'] = pending.pop()
            slot['gen'] = run_clip(slot['view'], slot['clip'])
        with frame_context.context(slot['ctx']), enter(slot):
            try:
                next(slot['gen'])
            except StopIteration as e:
                results[slot['clip']] = e.value
                slot['gen'] = None
        return True

    try:
        with torch.inference_mode():
            if on_gpu and not net.engine().tile_cache:
                # the conv tiles are not tuned yet: the first clip runs alone, so the autotuner times kernels on a quiet device
                while advance(slots[0]) and slots[0]['gen'] is not None:
                    pass
            busy = True
            while busy:
                busy = False
                for slot in slots:
                    busy = advance(slot) or busy
        if on_gpu:
            torch.cuda.synchronize(net.device)
    finally:
        for eng, was in lanes:
            eng.one_lane = was
    return results


def run_batched(net, cfg, clips: Sequence[Dict], *, lockstep: int = 4, in_flight: int = 1, lookahead: int = 16, on_frame: Callable = None) -> Dict[int, List]:
    """Several clips per GPU in LOCK STEP (cutie_amd/inference/lockstep.py): groups of ``lockstep`` clips advance together through ONE launch
    plan per stage -- batch = clips x objects for the pixel fusion, the object transformer, the decoder and the mask encoder, one joint
    encoder window, one memory bank and one look-ahead read-out lane per clip.  Per clip the results are bit-identical to its own
    ``InferenceCore`` run; next to ``run_interleaved`` this divides the launches and the host's issue time per frame by the group size and
    gives every convolution of the per-object path ``lockstep`` x the rows.  ``in_flight`` > 1 keeps that many GROUPS in flight next to
    each other (``run_interleaved`` over groups: a stream and a ``CUTIE.fork()`` per group, the calling thread issues a lock-step frame of
    every group in turn) -- the two schemes multiply.  MI355X, 480p, 3 objects, long-term memory (round 6, one box each): 4 clips ~1800 frames/s
    in one group against ~1530 interleaved one by one; 12 clips as 3 groups of 4 in flight ~2020.

    clips: [{'frames': sequence of [3, H, W] tensors on the device, 'mask': the first frame's mask, 'objects': its object ids}, ...];
    the clips of a group should share geometry, object count and length (a group whose clips differ still runs -- clip by clip where the
    states differ, see LockstepCores).  on_frame(clip_index, t, prob, core) receives every result (default: the uint8 object-id masks are
    collected).  Returns {clip index: [per-frame results]}."""
    from . import frame_context
    from .inference.lockstep import LockstepCores
    results: Dict[int, List] = {i: [] for i in range(len(clips))}
    if on_frame is None:
        def on_frame(i, t, prob, core):
            results[i].append(core.output_prob_to_mask(prob, dtype=torch.uint8))
    groups = [list(range(g0, min(len(clips), g0 + max(1, lockstep)))) for g0 in range(0, len(clips), max(1, lockstep))]

    def run_group(view, gi):
        """Generator: one lock-step frame of group gi per `next`."""
        group = groups[gi]
        frames = [clips[i]['frames'] for i in group]
        T = min(len(f) for f in frames)
        ls = LockstepCores(view, cfg, len(group))
        for t in range(T):
            hint = {}
            if lookahead > 0 and 0 < t < T - 1:
                hint = dict(next_images=[[f[j] for j in range(t + 1, min(T, t + 1 + lookahead))] for f in frames])
            end = all(t == len(f) - 1 for f in frames)
            if t == 0:
                probs = ls.step([f[0] for f in frames], [clips[i]['mask'] for i in group], [clips[i]['objects'] for i in group], end=end)
            else:
                probs = ls.step([f[t] for f in frames], end=end, **hint)
            for c, i in enumerate(group):
                on_frame(i, t, probs[c], ls.cores[c])
            yield
        for c, i in enumerate(group):                       # clips longer than the shortest of their group: the rest on their own
            f = frames[c]
            with frame_context.context(ls._ctx[c]):
                for t in range(T, len(f)):
                    on_frame(i, t, ls.cores[c].step(f[t], end=(t == len(f) - 1)), ls.cores[c])
                    yield
        return None

    if in_flight > 1 and len(groups) > 1:
        run_interleaved(net, list(range(len(groups))), run_group, streams=in_flight)
    else:
        with torch.inference_mode():
            for gi in range(len(groups)):
                for _ in run_group(net, gi):
                    pass
    return results


def _comm_device() -> torch.device:
    """Where point-to-point payloads of the default process group must live: the rank's GPU under RCCL ("nccl"), host memory under gloo."""
    if dist.is_initialized() and dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def run_sharded(clip_ids: Sequence[int], run_clip: Callable[[int], Dict], *, gather_masks: bool = False,
                owner_of: Callable[[int], int] = None):
    """Run ``run_clip(clip_id) -> {'frames': int, 'seconds': float, 'masks': uint8 tensor [T,H,W] (optional)}`` for
    this rank's share of ``clip_ids`` and gather the results on rank 0.  Returns {clip_id: result} on rank 0, None elsewhere.
    owner_of(i) -> rank of the i-th clip (default: round robin, i mod world size); rank 0 need not own any clip."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if owner_of is None:
        owner_of = lambda i: i % world
    owners = [int(owner_of(i)) % world for i in range(len(clip_ids))]
    mine = [c for i, c in enumerate(clip_ids) if owners[i] == rank]
    local = {}
    for c in mine:
        r = run_clip(c)
        if not gather_masks:
            r = {k: v for k, v in r.items() if k != 'masks'}
        local[c] = r
    if world == 1:
        return local
    # scalars via all_gather_object (tiny); masks via tensor gather so they travel device-to-device over xGMI
    stats = {c: {k: v for k, v in r.items() if k != 'masks'} for c, r in local.items()}
    gathered = [None] * world
    dist.all_gather_object(gathered, stats)
    result = {}
    for g in gathered:
        result.update(g)
    if gather_masks:
        dev = _comm_device()                 # (NOT the device of a local result: rank 0 may own no clip at all)
        for i, c in enumerate(clip_ids):
            owner = owners[i]
            if rank == owner and owner != 0:
                m = local[c]['masks'].to(dev).contiguous()
                meta = torch.tensor(list(m.shape), dtype=torch.int64, device=dev)
                dist.send(meta, dst=0)
                dist.send(m, dst=0)
            elif rank == 0:
                if owner == 0:
                    result[c]['masks'] = local[c]['masks']
                else:
                    meta = torch.zeros(3, dtype=torch.int64, device=dev)
                    dist.recv(meta, src=owner)
                    m = torch.empty(tuple(int(v) for v in meta.tolist()), dtype=torch.uint8, device=dev)
                    dist.recv(m, src=owner)
                    result[c]['masks'] = m
    return result if rank == 0 else None
