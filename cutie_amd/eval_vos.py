"""Dataset-level driver: the loop of the reference cutie/eval_vos.py:85-171 without hydra (one InferenceCore per video,
first-mask alignment, ``end`` on the last frame, FPS = frames / sum of device-event time around ``step``), on top of
``VOSTestDataset`` / ``VideoReader`` / ``ResultSaver``.

    python -m cutie_amd.eval_vos --images DIR/JPEGImages --masks DIR/Annotations --output OUT [--weights ckpt.pth]
        [--size 480] [--use-all-masks] [--long-term] [--dataset d17-val] [--visualize] [--clips-in-flight 2] [--lockstep 4]
        [--model small] [--flip-aug] [--save-scores]      (multi-scale testing: one run per --size with --save-scores, then
                                           python -m cutie_amd.merge_multi_scale --list OUT_a OUT_b --output MERGED)

With several GPUs launch it under torch.distributed.run: videos are sharded over the ranks (cutie_amd/parallel.py)."""
import argparse
import logging
import os
import time
from collections import deque
from os import path
from typing import Dict

import torch

from .config import default_config
from .inference.data.prefetch import ReadAhead
from .inference.data.vos_test_dataset import VOSTestDataset
from .inference.inference_core import InferenceCore
from .inference.utils.results_utils import ResultSaver, make_zip

log = logging.getLogger()


def process_video(network, cfg, vid_reader, mask_output_root, *, dataset='generic', save_all=True, visualize=False,
                  visualize_output_root=None, lookahead=True, save_scores=False, score_output_root=None,
                  read_workers=4) -> Dict:
    """One video through a fresh InferenceCore (eval_vos.py:97-151).  Returns {'frames', 'seconds'} (time around step)."""
    processor = InferenceCore(network, cfg=cfg)
    saver = ResultSaver(mask_output_root, vid_reader.vid_name, dataset=dataset, object_manager=processor.object_manager,
                        use_long_id=vid_reader.use_long_id, palette=vid_reader.get_palette(), visualize=visualize,
                        visualize_output_root=visualize_output_root, processor=processor, save_scores=save_scores,
                        score_output_root=score_output_root)
    dev = network.device
    on_gpu = dev.type == 'cuda'
    n = len(vid_reader)
    total, frames, first_mask_loaded = 0.0, 0, False
    try:
        loader = iter(ReadAhead(vid_reader, workers=read_workers))      # decode runs ahead on threads (eval_vos.py:92)
        # the frames of the following steps, already on the device: step(next_images=...) runs the image encoder over a window of them
        # (InferenceCore.prefetch_window); the SAME tensors are handed to the later steps (the look-ahead matches frames by storage)
        from .inference import inference_core as IC
        depth = (IC.WINDOW + IC.WINDOW_LEAD + 1) if lookahead else 1
        ahead = deque()

        def fill():
            while len(ahead) < depth:
                d = next(loader, None)
                if d is None:
                    break
                d['rgb'] = d['rgb'].to(dev)
                ahead.append(d)

        fill()
        for ti in range(n):
            data = ahead.popleft()
            fill()
            image = data['rgb']
            next_images = [d['rgb'] for d in ahead] if (lookahead and ahead) else None
            mask = data.get('mask')
            mask = mask.to(dev) if mask is not None else None
            valid = data.get('valid_labels')
            valid = valid.tolist() if valid is not None else None
            info = data['info']
            if not first_mask_loaded:
                if mask is None:
                    continue                                  # nothing to do before the first mask
                first_mask_loaded = True
            if on_gpu:
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                t0 = time.perf_counter()
            prob = processor.step(image, mask, valid, end=(ti == n - 1), next_images=next_images)
            if on_gpu:
                e1.record()
                torch.cuda.synchronize()
                total += e0.elapsed_time(e1) / 1000
            else:
                total += time.perf_counter() - t0
            frames += 1
            if save_all or info['save']:
                saver.process(prob, info['frame'], resize_needed=info['resize_needed'], shape=info['shape'],
                              last_frame=(ti == n - 1), path_to_image=info['path_to_image'])
    finally:
        saver.end()
    return {'frames': frames, 'seconds': total}


def lockstep_key(vid_reader):
    """What the videos of a lock-step group must share (cutie_amd/inference/lockstep.py): frame size as the model sees it, the number of
    objects of the first mask, the mask schedule.  None: the video cannot join a group (no mask on its first frame)."""
    d0 = vid_reader[0]
    if d0.get('mask') is None:
        return None
    return (tuple(d0['rgb'].shape[-2:]), int(len(d0['valid_labels'])), bool(vid_reader.use_all_mask))


def process_videos_lockstep(network, cfg, vid_readers, mask_output_root, *, dataset='generic', save_all=True, visualize=False,
                            visualize_output_root=None, lookahead=True, save_scores=False, score_output_root=None, read_workers=4) -> Dict[int, Dict]:
    """``process_video`` for a GROUP of videos advanced in lock step (``LockstepCores``: one launch plan per stage for the objects of all
    videos; per video the results of its own ``InferenceCore``).  The videos should share ``lockstep_key``; they may differ in length -- the
    group runs as long as its shortest video, the others finish on their own cores.  Returns {index in vid_readers: {'frames', 'seconds'}}
    (the seconds of a lock-step frame are split evenly over its videos)."""
    from .inference import inference_core as IC
    from .inference.lockstep import LockstepCores
    from . import frame_context
    C = len(vid_readers)
    ls = LockstepCores(network, cfg, C)
    dev = network.device
    on_gpu = dev.type == 'cuda'
    savers = [ResultSaver(mask_output_root, rd.vid_name, dataset=dataset, object_manager=ls.cores[c].object_manager, use_long_id=rd.use_long_id,
                          palette=rd.get_palette(), visualize=visualize, visualize_output_root=visualize_output_root, processor=ls.cores[c],
                          save_scores=save_scores, score_output_root=score_output_root) for c, rd in enumerate(vid_readers)]
    lens = [len(rd) for rd in vid_readers]
    T = min(lens)
    depth = 16 if lookahead else 1
    stats = {c: {'frames': 0, 'seconds': 0.0} for c in range(C)}

    def timed(fn):
        if on_gpu:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            torch.cuda.synchronize()
            return out, e0.elapsed_time(e1) / 1000
        t0 = time.perf_counter()
        out = fn()
        return out, time.perf_counter() - t0

    try:
        loaders = [iter(ReadAhead(rd, workers=read_workers)) for rd in vid_readers]
        ahead = [deque() for _ in range(C)]

        def fill(c):
            while len(ahead[c]) < depth:
                d = next(loaders[c], None)
                if d is None:
                    break
                d['rgb'] = d['rgb'].to(dev)
                ahead[c].append(d)

        for c in range(C):
            fill(c)
        for ti in range(T):
            data = [ahead[c].popleft() for c in range(C)]
            for c in range(C):
                fill(c)
            masks = [d['mask'].to(dev) if d.get('mask') is not None else None for d in data]
            valid = [d['valid_labels'].tolist() if d.get('valid_labels') is not None else None for d in data]
            any_mask = any(m is not None for m in masks)
            hint = None
            if lookahead and not any_mask and all(len(a) > 0 for a in ahead) and ti < T - 1:
                n = min(min(len(a) for a in ahead), T - 1 - ti)
                hint = [[d['rgb'] for d in list(a)[:n]] for a in ahead]
            end = all(ti == n_ - 1 for n_ in lens)
            probs, secs = timed(lambda: ls.step([d['rgb'] for d in data], masks if any_mask else None, valid if any_mask else None,
                                                end=end, next_images=hint))
            for c in range(C):
                stats[c]['frames'] += 1
                stats[c]['seconds'] += secs / C
                info = data[c]['info']
                if save_all or info['save']:
                    savers[c].process(probs[c], info['frame'], resize_needed=info['resize_needed'], shape=info['shape'],
                                      last_frame=(ti == lens[c] - 1), path_to_image=info['path_to_image'])
        for c in range(C):                                   # the videos longer than the shortest one: on their own cores
            core = ls.cores[c]
            with frame_context.context(ls._ctx[c]):
                for ti in range(T, lens[c]):
                    d = ahead[c].popleft()
                    fill(c)
                    nxt = [x['rgb'] for x in ahead[c]] if (lookahead and ahead[c]) else None
                    mask = d['mask'].to(dev) if d.get('mask') is not None else None
                    val = d['valid_labels'].tolist() if d.get('valid_labels') is not None else None
                    prob, secs = timed(lambda: core.step(d['rgb'], mask, val, end=(ti == lens[c] - 1), next_images=nxt))
                    stats[c]['frames'] += 1
                    stats[c]['seconds'] += secs
                    info = d['info']
                    if save_all or info['save']:
                        savers[c].process(prob, info['frame'], resize_needed=info['resize_needed'], shape=info['shape'],
                                          last_frame=(ti == lens[c] - 1), path_to_image=info['path_to_image'])
    finally:
        for s in savers:
            s.end()
    return stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', required=True)
    ap.add_argument('--masks', required=True)
    ap.add_argument('--output', required=True)
    ap.add_argument('--weights')
    ap.add_argument('--dataset', default='generic')
    ap.add_argument('--size', type=int, default=-1)
    ap.add_argument('--subset')
    ap.add_argument('--use-all-masks', action='store_true')
    ap.add_argument('--long-term', action='store_true')
    ap.add_argument('--visualize', action='store_true')
    ap.add_argument('--clips-in-flight', type=int, default=1)
    ap.add_argument('--lockstep', type=int, default=1, help='advance up to this many videos of one frame size / object count in LOCK STEP through one launch plan '
                                                            'per stage (cutie_amd/inference/lockstep.py; same results per video); the others run as --clips-in-flight says')
    ap.add_argument('--read-workers', type=int, default=4, help='decode threads per clip (0 = inline)')
    ap.add_argument('--flip-aug', action='store_true')
    ap.add_argument('--model', default='base', choices=['base', 'small'], help='cutie/config/model/{base,small}.yaml')
    ap.add_argument('--save-scores', action='store_true')
    args = ap.parse_args()
    from .model.cutie import CUTIE
    from .parallel import run_concurrent, shard_clips
    import torch.distributed as dist
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group(backend='nccl')
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    cfg = default_config(model=args.model, use_long_term=args.long_term, flip_aug=args.flip_aug, save_scores=args.save_scores)
    net = CUTIE(cfg).cuda().eval()
    if args.weights:
        net.load_weights(torch.load(args.weights, map_location='cpu'))
    meta = VOSTestDataset(args.images, args.masks, use_all_masks=args.use_all_masks, size=args.size, subset=args.subset)
    readers = list(meta.get_datasets())
    mine = shard_clips(len(readers), rank, world)
    mask_root = path.join(args.output, 'Annotations')
    run = lambda view, c: process_video(view, cfg, readers[c], mask_root, dataset=args.dataset, visualize=args.visualize,
                                        visualize_output_root=path.join(args.output, 'Visualizations'),
                                        save_scores=args.save_scores, score_output_root=path.join(args.output, 'Scores'),
                                        read_workers=args.read_workers)
    res = {}
    with torch.inference_mode():
        if args.lockstep > 1 and not args.flip_aug:
            # groups of videos that can advance together (same frame size, object count and mask schedule), longest first so that the
            # videos of a group have similar lengths; what is left over runs one by one below
            by_key = {}
            for c in mine:
                k = lockstep_key(readers[c])
                if k is not None:
                    by_key.setdefault(k, []).append(c)
            grouped = []
            for k, cs in by_key.items():
                cs.sort(key=lambda c: -len(readers[c]))
                for g0 in range(0, len(cs) - 1, args.lockstep):
                    grp = cs[g0:g0 + args.lockstep]
                    if len(grp) >= 2:
                        grouped.append(grp)
            for grp in grouped:
                st = process_videos_lockstep(net, cfg, [readers[c] for c in grp], mask_root, dataset=args.dataset, visualize=args.visualize,
                                             visualize_output_root=path.join(args.output, 'Visualizations'), save_scores=args.save_scores,
                                             score_output_root=path.join(args.output, 'Scores'), read_workers=args.read_workers)
                for j, c in enumerate(grp):
                    res[c] = st[j]
            mine = [c for c in mine if c not in res]
        res.update(run_concurrent(net, mine, run, streams=max(1, args.clips_in_flight)))
    frames, secs = sum(r['frames'] for r in res.values()), sum(r['seconds'] for r in res.values())
    print(f'rank {rank}: {frames} frames, {secs:.2f} s in step, FPS {frames / max(secs, 1e-9):.1f}')
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        make_zip(args.dataset, args.output, 'cutie_amd', mask_root)


if __name__ == '__main__':
    main()
