"""Mask propagation through one video given a few annotated frames: the workflow of the reference scripts/process_video.py
(:22-211) on the HIP path.

1. every mask in ``--mask_dir`` (``<frame number, 7 digits>.png``; palette 'P', 'L' or RGB long ids) is committed to
   PERMANENT memory first: ``step(frame, one_hot[1:], idx_mask=False, force_permanent=True)`` (:94-120);
2. the video then runs from frame 0; frames that have a mask are stepped with it, the others propagate (:139-176); every
   output goes through ``ResultSaver`` (fused argmax + id remap on the device, PNG encoding on a writer thread);
3. ``--mem_cleanup_ratio r``: when used / total device memory exceeds r the non-permanent memory is cleared (:214-228).

Frame source: a directory of images (sorted by name) or, when OpenCV is importable, a video file.  Differences from the
reference, on purpose: frames from a directory are RGB (the reference hands OpenCV's BGR arrays to the network unchanged,
:105,162); there is no CPU / MPS device switch -- the product has no CPU path; defaults are the reference's video_config.yaml
(long-term memory on, mem_every 10, max_internal_size 480).

    python -m cutie_amd.process_video -v FRAMES_DIR -m MASK_DIR -o OUT [--weights ckpt.pth] [--mem_every 10]
        [--max_internal_size 480] [--mem_cleanup_ratio 0.9] [--num_objects N] [--model small]"""
import os
from argparse import ArgumentParser
from os import path
from typing import Callable, Dict, Iterator, Optional, Tuple

import numpy as np
import torch
from PIL import Image

from .config import default_config
from .inference.data.prefetch import ReadAhead
from .inference.inference_core import InferenceCore
from .inference.utils.results_utils import ResultSaver

IMAGE_EXT = ('.jpg', '.jpeg', '.png', '.bmp')


def video_config(**overrides):
    """cutie/config/video_config.yaml: long-term memory on, mem_every 10, max_internal_size 480."""
    base = dict(use_long_term=True, mem_every=10, max_internal_size=480)
    base.update(overrides)
    return default_config(**base)


class FrameSource:
    """Random access + sequential reading of frames as float [3,H,W] in [0,1] (gui/interactive_utils.py:11-15)."""

    def __init__(self, video: str):
        self.cap = None
        if path.isdir(video):
            self.names = sorted(n for n in os.listdir(video) if n.lower().endswith(IMAGE_EXT))
            self.root = video
            self.count = len(self.names)
        else:
            try:
                import cv2
            except ImportError as e:
                raise RuntimeError(f'{video} is not a directory of frames and OpenCV is not available to decode a video file') from e
            self.cv2 = cv2
            self.cap = cv2.VideoCapture(video)
            if not self.cap.isOpened():
                raise RuntimeError(f'Unable to open video {video}!')
            self.count = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))

    def __len__(self):
        return self.count

    def read(self, index: int) -> Optional[torch.Tensor]:
        if self.cap is None:
            if not 0 <= index < self.count:
                return None
            arr = np.array(Image.open(path.join(self.root, self.names[index])).convert('RGB'))
        else:
            self.cap.set(self.cv2.CAP_PROP_POS_FRAMES, index)
            ok, arr = self.cap.read()
            if arr is None:
                return None
            arr = arr[:, :, ::-1].copy()                          # BGR -> RGB
        return torch.from_numpy(arr).permute(2, 0, 1).float() / 255

    def __iter__(self) -> Iterator[torch.Tensor]:
        # a directory decodes ahead on threads; a cv2.VideoCapture is a sequential, stateful decoder -> inline
        for f in ReadAhead(self, workers=4 if self.cap is None else 0, length=self.count, getitem=self.read):
            if f is None:
                return
            yield f

    def close(self):
        if self.cap is not None:
            self.cap.release()


def one_hot_planes(mask_np: np.ndarray, num_objects: int, device) -> torch.Tensor:
    """index mask -> float one-hot [num_objects, H, W] without the background plane
    (gui/interactive_utils.py:24-26 followed by ``[1:]``); ids above num_objects are an error there too."""
    m = torch.from_numpy(np.ascontiguousarray(mask_np)).long().to(device)
    if int(m.max()) > num_objects:
        raise RuntimeError(f'mask holds id {int(m.max())} but num_objects is {num_objects}')
    ids = torch.arange(1, num_objects + 1, device=device).view(-1, 1, 1)
    return (m.unsqueeze(0) == ids).float()


def check_to_clear_non_permanent_memory(processor: InferenceCore, mem_cleanup_ratio: float,
                                        mem_get_info: Callable[[], Tuple[int, int]] = None) -> bool:
    """scripts/process_video.py:214-228.  Returns True when a cleanup was triggered."""
    if not (0 < mem_cleanup_ratio <= 1):
        return False
    free, total = (mem_get_info or torch.cuda.mem_get_info)()
    ratio = (total - free) / total
    if ratio > mem_cleanup_ratio:
        print(f'GPU cleanup triggered: {ratio} > {mem_cleanup_ratio}')
        processor.clear_non_permanent_memory()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        return True
    return False


def process_video(network, cfg, video: str, mask_dir: str, output_dir: str, *, num_objects: int = -1,
                  mem_cleanup_ratio: float = -1, mem_get_info=None, lookahead: bool = True) -> Dict:
    dev = network.device
    src = FrameSource(video)
    mask_names = sorted(n for n in os.listdir(mask_dir) if n.lower().endswith('.png'))
    if not mask_names:
        raise RuntimeError('No mask frames found!')
    first = Image.open(path.join(mask_dir, mask_names[0]))
    if first.mode == 'P':
        use_long_id, palette = False, first.getpalette()
    elif first.mode == 'RGB':
        use_long_id, palette = True, None
    elif first.mode == 'L':
        use_long_id, palette = False, None
    else:
        raise RuntimeError(f'Unknown mode {first.mode} in {mask_names[0]}.')

    def index_mask(name):
        arr = np.array(Image.open(path.join(mask_dir, name)))
        if use_long_id:
            arr = arr.astype(np.int64)
            arr = arr[..., 0] + 256 * arr[..., 1] + 65536 * arr[..., 2]
        return arr

    if num_objects is None or num_objects < 1:
        num_objects = len(np.unique(index_mask(mask_names[0]))) - 1
    processor = InferenceCore(network, cfg=cfg)
    on_gpu = dev.type == 'cuda'
    with torch.inference_mode():
        # 1. commit the annotated frames to permanent memory
        for name in mask_names:
            frame = src.read(int(name[:-4]))
            if frame is None:
                break
            processor.step(frame.to(dev), one_hot_planes(index_mask(name), num_objects, dev), idx_mask=False,
                           force_permanent=True)
        # 2. the whole video
        saver = ResultSaver(output_dir, '', dataset='', object_manager=processor.object_manager, use_long_id=use_long_id,
                            palette=palette, processor=processor)
        total, n, cleanups = 0.0, 0, 0
        try:
            it = iter(src)
            nxt = next(it, None)
            nxt = nxt.to(dev) if nxt is not None else None
            while nxt is not None:
                frame, nxt = nxt, next(it, None)
                nxt = nxt.to(dev) if nxt is not None else None
                name = f'{n:07d}.png'
                mask = one_hot_planes(index_mask(name), num_objects, dev) if path.exists(path.join(mask_dir, name)) else None
                if on_gpu:
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                hint = nxt if lookahead else None
                prob = processor.step(frame, mask, idx_mask=False, next_image=hint) if mask is not None \
                    else processor.step(frame, next_image=hint)
                if on_gpu:
                    e1.record()
                    torch.cuda.synchronize()
                    total += e0.elapsed_time(e1) / 1000
                saver.process(prob, name, resize_needed=False, shape=None, last_frame=(n == len(src) - 1), path_to_image=None)
                cleanups += check_to_clear_non_permanent_memory(processor, mem_cleanup_ratio, mem_get_info)
                n += 1
        finally:
            saver.end()
            src.close()
    return {'frames': n, 'seconds': total, 'cleanups': cleanups, 'num_objects': num_objects, 'processor': processor}


def main():
    ap = ArgumentParser()
    ap.add_argument('-v', '--video', required=True, help='directory of frames (or a video file when OpenCV is available)')
    ap.add_argument('-m', '--mask_dir', required=True, help='masks named <frame number, 7 digits>.png')
    ap.add_argument('-o', '--output_dir', required=True)
    ap.add_argument('--weights')
    ap.add_argument('--num_objects', type=int, default=-1)
    ap.add_argument('--mem_every', type=int, default=10)
    ap.add_argument('--max_internal_size', type=int, default=480)
    ap.add_argument('--mem_cleanup_ratio', type=float, default=-1)
    ap.add_argument('--model', default='base', choices=['base', 'small'], help='cutie/config/model/{base,small}.yaml')
    args = ap.parse_args()
    from .model.cutie import CUTIE
    cfg = video_config(model=args.model, mem_every=args.mem_every, max_internal_size=args.max_internal_size)
    net = CUTIE(cfg).cuda().eval()
    if args.weights:
        net.load_weights(torch.load(args.weights, map_location='cpu'))
    else:
        print('No model weights loaded. Are you sure about this?')
    r = process_video(net, cfg, args.video, args.mask_dir, args.output_dir, num_objects=args.num_objects,
                      mem_cleanup_ratio=args.mem_cleanup_ratio)
    print(f'Total processing time: {r["seconds"]}\nTotal processed frames: {r["frames"]}\n'
          f'FPS: {r["frames"] / max(r["seconds"], 1e-9)}\nMax allocated memory (MB): {torch.cuda.max_memory_allocated() / 2 ** 20}')


if __name__ == '__main__':
    main()
