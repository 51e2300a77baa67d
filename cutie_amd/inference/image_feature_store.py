"""Cache of the per-frame encoder outputs, keyed by frame index; surface of cutie/inference/image_feature_store.py:7-49
(``get_features`` / ``get_key`` compute on first use, ``delete``, ``len``, a warning at destruction if entries were left).

Here ONE fused launch plan produces (ms_features, pix_feat) and (key, shrinkage, selection) together: CUTIE.encode_image
leaves the key outputs with the features and CUTIE.transform_key picks them up, so the two getters share one record.
The look-ahead encoder of InferenceCore (side stream) deposits its record through ``_store`` directly."""
import warnings

from .. import frame_context
from typing import Dict, Iterable, NamedTuple, Tuple

import torch


class _Record(NamedTuple):
    ms_features: Iterable[torch.Tensor]
    pix_feat: torch.Tensor
    key: torch.Tensor
    shrinkage: torch.Tensor
    selection: torch.Tensor


class ImageFeatureStore:
    def __init__(self, network, no_warning: bool = False):
        self.network, self.no_warning = network, no_warning
        self._store: Dict[int, Tuple] = {}

    def _record(self, index: int, image: torch.Tensor) -> Tuple:
        rec = self._store.get(index)
        if rec is None:
            net = self.network
            geometry = frame_context.recall('geometry', image)  # InferenceCore: un-padded frame + pad geometry (padding is
            if geometry is None:                                 # fused into the first kernel)
                ms, pix = net.encode_image(image)
            else:
                ms, pix = net._encode_image_raw(image, *geometry)
            rec = self._store[index] = _Record(ms, pix, *net.transform_key(ms[0]))
        return rec

    # kept for callers that pre-compute a frame (the reference names it the same way)
    def _encode_feature(self, index: int, image: torch.Tensor) -> None:
        self._store.pop(index, None)
        self._record(index, image)

    def get_features(self, index: int, image: torch.Tensor) -> (Iterable[torch.Tensor], torch.Tensor):
        rec = self._record(index, image)
        return rec[0], rec[1]

    def get_key(self, index: int, image: torch.Tensor) -> (torch.Tensor, torch.Tensor, torch.Tensor):
        rec = self._record(index, image)
        return rec[2], rec[3], rec[4]

    def delete(self, index: int) -> None:
        self._store.pop(index, None)

    def __len__(self) -> int:
        return len(self._store)

    def __del__(self):
        if self._store and not self.no_warning:
            warnings.warn(f'Leaking {self._store.keys()} in the image feature store')
