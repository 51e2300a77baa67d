"""DAVIS / YouTubeVOS / MOSE style test set = one sub-folder per video; same surface as the reference
cutie/inference/data/vos_test_dataset.py:9-65 (``subset`` txt, YouTubeVOS ``req_frames_json``)."""
import json
import os
from os import path
from typing import Iterable, Optional

from .video_reader import VideoReader


class VOSTestDataset:
    def __init__(self, image_dir: str, mask_dir: str, *, use_all_masks: bool, req_frames_json: Optional[str] = None,
                 size: int = -1, size_dir: Optional[str] = None, subset: Optional[str] = None):
        self.image_dir, self.mask_dir, self.use_all_masks = image_dir, mask_dir, use_all_masks
        self.size, self.size_dir = size, size_dir
        if subset is None:
            self.vid_list = sorted(os.listdir(self.mask_dir))
        else:
            with open(subset) as f:                           # DAVIS-2017 style txt
                self.vid_list = sorted(line.strip() for line in f)
        self.req_frame_list = {}
        if req_frames_json is not None:                       # YouTubeVOS meta.json: frames required for evaluation
            with open(req_frames_json) as f:
                meta = json.load(f)['videos']
            for vid in self.vid_list:
                req = []
                for value in meta[vid]['objects'].values():
                    req.extend(value['frames'])
                self.req_frame_list[vid] = list(set(req))

    def get_datasets(self) -> Iterable[VideoReader]:
        for video in self.vid_list:
            yield VideoReader(video, path.join(self.image_dir, video), path.join(self.mask_dir, video), size=self.size,
                              to_save=self.req_frame_list.get(video, None), use_all_masks=self.use_all_masks,
                              size_dir=path.join(self.size_dir, video) if self.size_dir is not None else None)

    def __len__(self):
        return len(self.vid_list)
