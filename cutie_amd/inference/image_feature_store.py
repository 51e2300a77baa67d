"""Per-frame feature cache; same surface as cutie/inference/image_feature_store.py:7-49.

One fused launch plan produces (ms_features, pix_feat) and (key, shrinkage, selection) together
(CUTIE.encode_image caches the key outputs, CUTIE.transform_key picks them up).
"""
import warnings
from typing import Iterable

import torch


class ImageFeatureStore:
    def __init__(self, network, no_warning: bool = False):
        self.network = network
        self._store = {}
        self.no_warning = no_warning

    def _encode_feature(self, index: int, image: torch.Tensor) -> None:
        raw = getattr(image, '_cutie_raw', None)
        if raw is not None:
            # InferenceCore hands the un-padded frame + pad geometry; padding is fused into the first kernel
            ms_features, pix_feat = self.network._encode_image_raw(image, *raw)
        else:
            ms_features, pix_feat = self.network.encode_image(image)
        key, shrinkage, selection = self.network.transform_key(ms_features[0])
        self._store[index] = (ms_features, pix_feat, key, shrinkage, selection)

    def get_features(self, index: int, image: torch.Tensor) -> (Iterable[torch.Tensor], torch.Tensor):
        if index not in self._store:
            self._encode_feature(index, image)
        return self._store[index][:2]

    def get_key(self, index: int, image: torch.Tensor) -> (torch.Tensor, torch.Tensor, torch.Tensor):
        if index not in self._store:
            self._encode_feature(index, image)
        return self._store[index][2:]

    def delete(self, index: int) -> None:
        self._store.pop(index, None)

    def __len__(self):
        return len(self._store)

    def __del__(self):
        if len(self._store) > 0 and not self.no_warning:
            warnings.warn(f'Leaking {self._store.keys()} in the image feature store')
