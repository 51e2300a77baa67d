"""DAVIS / YouTubeVOS / MOSE style test set = one sub-folder per video; same surface as the reference
cutie/inference/data/vos_test_dataset.py:9-65 (``subset`` txt, YouTubeVOS ``req_frames_json``)."""
import json
import os
from typing import Dict, Iterable, List, Optional

from .video_reader import VideoReader


def _video_names(mask_dir: str, subset: Optional[str]) -> List[str]:
    """Every annotated video, or the ones a DAVIS-2017 style txt file names (one per line); sorted either way."""
    if subset is None:
        return sorted(os.listdir(mask_dir))
    with open(subset) as f:
        return sorted(line.strip() for line in f)


def _required_frames(meta_json: Optional[str], videos: List[str]) -> Dict[str, List[str]]:
    """YouTubeVOS ``meta.json``: the frames the evaluation server asks for = the union of the frames of a video's objects."""
    if meta_json is None:
        return {}
    with open(meta_json) as f:
        per_video = json.load(f)['videos']
    return {v: sorted({fr for obj in per_video[v]['objects'].values() for fr in obj['frames']}) for v in videos}


class VOSTestDataset:
    def __init__(self, image_dir: str, mask_dir: str, *, use_all_masks: bool, req_frames_json: Optional[str] = None,
                 size: int = -1, size_dir: Optional[str] = None, subset: Optional[str] = None):
        self.image_dir, self.mask_dir, self.size_dir = image_dir, mask_dir, size_dir
        self.use_all_masks, self.size = use_all_masks, size
        self.vid_list = _video_names(mask_dir, subset)
        self.req_frame_list = _required_frames(req_frames_json, self.vid_list)

    def reader(self, video: str) -> VideoReader:
        sub = lambda root: os.path.join(root, video)       # noqa: E731
        return VideoReader(video, sub(self.image_dir), sub(self.mask_dir), size=self.size, use_all_masks=self.use_all_masks,
                           to_save=self.req_frame_list.get(video), size_dir=None if self.size_dir is None else sub(self.size_dir))

    def get_datasets(self) -> Iterable[VideoReader]:
        return (self.reader(v) for v in self.vid_list)

    def __len__(self):
        return len(self.vid_list)
