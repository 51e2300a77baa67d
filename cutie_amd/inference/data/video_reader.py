"""Frame-by-frame reader of one video folder; same surface and semantics as the reference
cutie/inference/data/video_reader.py:14-165 (JPEG frames, PNG masks; palette / RGB long-id / greyscale masks; optional
resize of the shorter side to ``size``: bilinear + antialias for frames, nearest for masks; ``to_save`` / ``use_all_masks`` /
``start`` / ``end`` / ``reverse`` / ``enabled_frame_list``).  torchvision is not required: ToTensor and Resize are spelled
with numpy / torch.nn.functional (identical arithmetic: F.interpolate(..., antialias=True) is what torchvision calls).

Structure (ours): the frame list, the mask format and the two decoders are separate helpers; ``__getitem__`` only assembles the
record.  Behaviour is pinned by recordings of the executed reference class (tests/golden/io, tests/test_io_fixtures_cpu.py)."""
import os
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

# PIL mode of the first mask -> (ids are 24-bit RGB triples, the palette is kept)
_MASK_FORMATS = {'P': (False, True), 'RGB': (True, False), 'L': (False, False)}


def _stem(name: str) -> str:
    return name[:-4]                                       # "00012.jpg" -> "00012" (the reference cuts four characters as well)


def _shorter_side_to(h: int, w: int, size: int):
    """torchvision.transforms.Resize(int): the shorter side becomes ``size``, aspect ratio kept (the longer side truncated)."""
    return (size, int(size * w / h)) if h <= w else (int(size * h / w), size)


def _frame_window(names: List[str], start: int, end: int, reverse: bool) -> List[str]:
    """``start`` / ``end`` as Python slice bounds where they are >= 0 (a negative value means "open"), then the optional reversal."""
    lo = start if start >= 0 else None
    hi = end if end >= 0 else None
    picked = names[lo:hi] if (lo is not None or hi is not None) else list(names)
    return picked[::-1] if reverse else picked


def _decode_rgb(img_path: str):
    """-> (float32 [3, h, w] in 0..1 -- ToTensor --, (h, w))"""
    img = Image.open(img_path).convert('RGB')
    arr = np.asarray(img, dtype=np.uint8).copy()
    return torch.from_numpy(arr).permute(2, 0, 1).float().div_(255.0), (img.height, img.width)


def _decode_mask(mask_path: str, long_ids: bool, size: Optional[int]):
    """-> int64 [h, w] object ids; ``size``: nearest-neighbour resize of the shorter side first."""
    m = Image.open(mask_path)
    if size is not None:
        nh, nw = _shorter_side_to(m.height, m.width, size)
        m = m.resize((nw, nh), Image.NEAREST)
    ids = torch.from_numpy(np.array(m)).long()
    if long_ids:
        assert ids.dim() == 3, 'RGB masks should have 3 dimensions'
        return ids[..., 0] + (ids[..., 1] << 8) + (ids[..., 2] << 16)
    assert ids.dim() == 2, 'Single channel masks should have 2 dimensions'
    return ids


class VideoReader(torch.utils.data.Dataset):
    def __init__(self, vid_name: str, image_dir: str, mask_dir: str, *, size: int = -1, to_save: Optional[List[str]] = None,
                 use_all_masks: bool = False, size_dir: Optional[str] = None, start: int = -1, end: int = -1,
                 reverse: bool = False, object_name: str = None, enabled_frame_list: Optional[List[str]] = None):
        # caller-visible attributes (names as in the reference)
        self.vid_name, self.object_name = vid_name, object_name
        self.image_dir, self.mask_dir = image_dir, mask_dir
        self.size_dir = image_dir if size_dir is None else size_dir
        self.size, self.to_save, self.use_all_mask = size, to_save, use_all_masks
        self.enabled_frame_list = enabled_frame_list

        every = sorted(os.listdir(image_dir))
        if enabled_frame_list is not None:
            allowed = set(enabled_frame_list)
            every = [f for f in every if _stem(f) in allowed]
        self._time_index = {f: t for t, f in enumerate(every)}      # position in the un-windowed video (info['time_index'])
        self.frames = _frame_window(every, start, end, reverse)

        self.first_mask_frame = min(os.listdir(mask_dir))            # (= sorted(...)[0])
        mode = Image.open(os.path.join(mask_dir, self.first_mask_frame)).mode
        if mode not in _MASK_FORMATS:
            raise NotImplementedError(f'Unknown mode {mode} in {self.first_mask_frame}.')
        self.use_long_id, keep_palette = _MASK_FORMATS[mode]
        self.palette = Image.open(os.path.join(mask_dir, self.first_mask_frame)).getpalette() if keep_palette else None

    def _wants_mask(self, frame: str) -> bool:
        return self.use_all_mask or _stem(frame) == _stem(self.first_mask_frame)

    def __getitem__(self, idx):
        frame = self.frames[idx]
        im_path = os.path.join(self.image_dir, frame)
        rgb, in_hw = _decode_rgb(im_path)
        if self.size_dir == self.image_dir:
            out_hw = in_hw
        else:
            ref = Image.open(os.path.join(self.size_dir, frame))
            out_hw = (ref.height, ref.width)
        shrink = in_hw != out_hw or (self.size > 0 and min(in_hw) > self.size)
        if shrink:
            rgb = F.interpolate(rgb[None], size=_shorter_side_to(*in_hw, self.size), mode='bilinear', align_corners=False, antialias=True)[0]
        data = {}
        mask_path = os.path.join(self.mask_dir, _stem(frame) + '.png')
        if self._wants_mask(frame) and os.path.exists(mask_path):
            ids = _decode_mask(mask_path, self.use_long_id, self.size if shrink else None)
            present = torch.unique(ids)
            data['mask'], data['valid_labels'] = ids, present[present != 0]
        data['rgb'] = rgb
        data['info'] = {'frame': frame, 'save': self.to_save is None or _stem(frame) in self.to_save, 'shape': out_hw,
                        'resize_needed': shrink, 'time_index': self._time_index[frame], 'path_to_image': im_path}
        return data

    def get_palette(self):
        return self.palette

    def __len__(self):
        return len(self.frames)
