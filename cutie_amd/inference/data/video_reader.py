"""Frame-by-frame reader of one video folder; same surface and semantics as the reference
cutie/inference/data/video_reader.py:14-165 (JPEG frames, PNG masks; palette / RGB long-id / greyscale masks; optional
resize of the shorter side to ``size``: bilinear + antialias for frames, nearest for masks; ``to_save`` / ``use_all_masks`` /
``start`` / ``end`` / ``reverse`` / ``enabled_frame_list``).  torchvision is not required: ToTensor and Resize are spelled
with numpy / torch.nn.functional (identical arithmetic: F.interpolate(..., antialias=True) is what torchvision calls)."""
import copy
import os
from os import path
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


def _resized_hw(h: int, w: int, size: int):
    """torchvision.transforms.Resize(int): the shorter side becomes ``size``, aspect ratio kept."""
    if h <= w:
        return size, int(size * w / h)
    return int(size * h / w), size


class VideoReader(torch.utils.data.Dataset):
    def __init__(self, vid_name: str, image_dir: str, mask_dir: str, *, size: int = -1, to_save: Optional[List[str]] = None,
                 use_all_masks: bool = False, size_dir: Optional[str] = None, start: int = -1, end: int = -1,
                 reverse: bool = False, object_name: str = None, enabled_frame_list: Optional[List[str]] = None):
        self.vid_name, self.image_dir, self.mask_dir = vid_name, image_dir, mask_dir
        self.to_save, self.use_all_mask, self.object_name = to_save, use_all_masks, object_name
        self.enabled_frame_list = enabled_frame_list
        self.size_dir = self.image_dir if size_dir is None else size_dir
        self.frames = sorted(os.listdir(self.image_dir))
        if enabled_frame_list is not None:
            self.frames = [f for f in self.frames if f[:-4] in enabled_frame_list]
        self._all_frames = copy.deepcopy(self.frames)
        if start >= 0:
            self.frames = self.frames[start:end] if end >= 0 else self.frames[start:]
        elif end >= 0:
            self.frames = self.frames[:end]
        if reverse:
            self.frames = list(reversed(self.frames))
        # 3-channel long ids or 1-channel (0..255) short ids?
        self.first_mask_frame = sorted(os.listdir(self.mask_dir))[0]
        first_mask = Image.open(path.join(self.mask_dir, self.first_mask_frame))
        if first_mask.mode == 'P':
            self.use_long_id, self.palette = False, first_mask.getpalette()
        elif first_mask.mode == 'RGB':
            self.use_long_id, self.palette = True, None
        elif first_mask.mode == 'L':
            self.use_long_id, self.palette = False, None
        else:
            raise NotImplementedError(f'Unknown mode {first_mask.mode} in {self.first_mask_frame}.')
        self.size = size

    def __getitem__(self, idx):
        frame = self.frames[idx]
        info = {'frame': frame, 'save': (self.to_save is None) or (frame[:-4] in self.to_save)}
        data = {}
        im_path = path.join(self.image_dir, frame)
        img = Image.open(im_path).convert('RGB')
        input_shape = (img.height, img.width)
        if self.image_dir == self.size_dir:
            output_shape = input_shape
        else:
            size_im = Image.open(path.join(self.size_dir, frame))
            output_shape = (size_im.height, size_im.width)
        resize_needed = (input_shape != output_shape) or ((self.size > 0) and (min(input_shape) > self.size))
        rgb = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)   # ToTensor
        if resize_needed:
            nh, nw = _resized_hw(*input_shape, self.size)
            rgb = F.interpolate(rgb.unsqueeze(0), size=(nh, nw), mode='bilinear', align_corners=False, antialias=True)[0]
        load_mask = self.use_all_mask or (frame[:-4] == self.first_mask_frame[:-4])
        if load_mask:
            mask_path = path.join(self.mask_dir, frame[:-4] + '.png')
            if path.exists(mask_path):
                mask = Image.open(mask_path)
                if resize_needed:
                    nh, nw = _resized_hw(mask.height, mask.width, self.size)
                    mask = mask.resize((nw, nh), Image.NEAREST)
                mask = torch.from_numpy(np.array(mask)).long()
                if self.use_long_id:
                    assert mask.dim() == 3, 'RGB masks should have 3 dimensions'
                    mask = mask[:, :, 0] + mask[:, :, 1] * 256 + mask[:, :, 2] * 256 * 256
                else:
                    assert mask.dim() == 2, 'Single channel masks should have 2 dimensions'
                valid = torch.unique(mask)
                data['mask'] = mask
                data['valid_labels'] = valid[valid != 0]
        info['shape'] = output_shape
        info['resize_needed'] = resize_needed
        info['time_index'] = self._all_frames.index(frame)
        info['path_to_image'] = im_path
        data['rgb'] = rgb
        data['info'] = info
        return data

    def get_palette(self):
        return self.palette

    def __len__(self):
        return len(self.frames)
