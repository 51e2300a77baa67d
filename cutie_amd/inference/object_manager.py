"""Object id <-> tensor-slot bookkeeping; same surface as cutie/inference/object_manager.py:7-149.

Object ids are immutable; "tmp ids" are 1-based positions of the objects in the channel dimension of every
per-object tensor (0 = background) and are compacted when objects are deleted.
"""
from typing import Dict, List, Union

import torch

from .object_info import ObjectInfo


class ObjectManager:
    def __init__(self):
        self.obj_to_tmp_id: Dict[ObjectInfo, int] = {}
        self.tmp_id_to_obj: Dict[int, ObjectInfo] = {}
        self.obj_id_to_obj: Dict[int, ObjectInfo] = {}
        self.all_historical_object_ids: List[int] = []

    def _reindex(self) -> None:
        self.obj_id_to_obj = {o.id: o for o in self.obj_to_tmp_id}

    def add_new_objects(self, objects: Union[List[ObjectInfo], ObjectInfo, List[int]]) -> (List[int], List[int]):
        if not isinstance(objects, list):
            objects = [objects]
        tmp_ids, obj_ids = [], []
        for obj in objects:
            info = ObjectInfo(id=obj) if isinstance(obj, int) else obj
            if info not in self.obj_to_tmp_id:
                info = ObjectInfo(id=info.id)
                slot = len(self.obj_to_tmp_id) + 1
                self.obj_to_tmp_id[info] = slot
                self.tmp_id_to_obj[slot] = info
                self.all_historical_object_ids.append(info.id)
            tmp_ids.append(self.obj_to_tmp_id[info])
            obj_ids.append(info.id)
        self._reindex()
        assert tmp_ids == sorted(tmp_ids), 'objects must arrive in tmp-id order (reference object_manager.py:53)'
        return tmp_ids, obj_ids

    def delete_objects(self, obj_ids_to_remove: Union[int, List[int]]) -> None:
        if isinstance(obj_ids_to_remove, int):
            obj_ids_to_remove = [obj_ids_to_remove]
        survivors = [self.tmp_id_to_obj[t] for t in range(1, len(self.obj_to_tmp_id) + 1)
                     if self.tmp_id_to_obj[t].id not in obj_ids_to_remove]
        self.obj_to_tmp_id = {o: i + 1 for i, o in enumerate(survivors)}
        self.tmp_id_to_obj = {i + 1: o for i, o in enumerate(survivors)}
        self._reindex()

    def purge_inactive_objects(self, max_missed_detection_count: int) -> (bool, List[int], List[int]):
        dead = [o.id for o in self.obj_to_tmp_id if o.poke_count > max_missed_detection_count]
        keep_tmp = [t for o, t in self.obj_to_tmp_id.items() if o.id not in dead]
        keep_obj = [o.id for o in self.obj_to_tmp_id if o.id not in dead]
        if dead:
            self.delete_objects(dead)
        return len(dead) > 0, keep_tmp, keep_obj

    def tmp_to_obj_cls(self, mask) -> torch.Tensor:
        lut = torch.zeros(len(self.tmp_id_to_obj) + 1, dtype=mask.dtype, device=mask.device)
        for t, o in self.tmp_id_to_obj.items():
            lut[t] = o.id
        return lut[mask.long()]

    def get_tmp_to_obj_mapping(self) -> Dict[int, int]:
        """{object id: tmp id} (what object_manager.py:106-108 is meant to return; the reference unpacks the (tmp id, object)
        items the other way round and raises AttributeError, so nothing there can depend on it)."""
        return {obj.id: tmp_id for tmp_id, obj in self.tmp_id_to_obj.items()}

    def realize_dict(self, obj_dict, dim=1) -> torch.Tensor:
        out = []
        for _, obj in self.tmp_id_to_obj.items():
            if obj.id not in obj_dict:
                raise NotImplementedError
            out.append(obj_dict[obj.id])
        return torch.stack(out, dim=dim)

    def make_one_hot(self, cls_mask) -> torch.Tensor:
        out = [cls_mask == obj.id for _, obj in self.tmp_id_to_obj.items()]
        if not out:
            return torch.zeros((0, *cls_mask.shape), dtype=torch.bool, device=cls_mask.device)
        return torch.stack(out, dim=0)

    @property
    def all_obj_ids(self) -> List[int]:
        return [o.id for o in self.obj_to_tmp_id]

    @property
    def num_obj(self) -> int:
        return len(self.obj_to_tmp_id)

    def has_all(self, objects: List[int]) -> bool:
        return all(o in self.obj_to_tmp_id for o in objects)

    def find_object_by_id(self, obj_id) -> ObjectInfo:
        return self.obj_id_to_obj[obj_id]

    def find_tmp_by_id(self, obj_id) -> int:
        return self.obj_to_tmp_id[self.obj_id_to_obj[obj_id]]
