"""Contiguous HBM memory bank replacing the reference's ``KeyValueMemoryStore``
(cutie/inference/kv_memory_store.py:19-352).

The reference grows every key/value tensor with ``torch.cat`` (an O(N) realloc+copy per memory frame) and keeps
working / long-term memory in separate stores that are concatenated again on every read.  Here each bucket
(objects first seen in the same frame share keys, kv_memory_store.py:27-31) owns ONE pre-allocated slab per
array, laid out by physical token slot:

        [ long-term region (L slots) | permanent region (P slots) | working region (Wc slots) | 16 slack ]

so the affinity kernels read at most three contiguous slot ranges and a token's slot indexes every array
(similarity operands, raw key/shrinkage/selection for consolidation, usage counters, per-object values).
FIFO eviction is a ring overwrite; long-term consolidation compacts the working region with one strided copy.
Sized for 288 GB HBM: a 480p bucket with long-term memory is ~26k slots x (0.9 + 0.5 K) KB ~ 60 MB at K=3.
"""
from typing import Dict, List

import numpy as np
import torch


BF16, F32 = torch.bfloat16, torch.float32
LIFE_EPS_BITS = int(np.float32(1e-7).view(np.int32))      # new tokens start with life = 1e-7 (kv_memory_store.py:134)


class Bucket:
    def __init__(self, bucket_id, objects, HW, CK, CV, device, *, use_long_term, work_cap, long_cap):
        self.id = bucket_id
        self.objects: List[int] = list(objects)
        self.HW, self.CK, self.CV, self.device = HW, CK, CV, device
        self.lt = use_long_term
        self.L = long_cap if use_long_term else 0
        self.P = HW                      # grows on demand (force_permanent)
        self.Wc = work_cap
        self.n_long = self.n_perm = self.n_work = 0
        self.ring = 0                    # FIFO write offset (frames) inside the working region
        self.values: Dict[int, torch.Tensor] = {}
        self._alloc()

    # ---- layout ---------------------------------------------------------------------------
    @property
    def slots(self):
        return self.L + self.P + self.Wc + 16

    @property
    def perm_start(self):
        return self.L

    @property
    def work_start(self):
        return self.L + self.P

    def _new(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def _alloc(self):
        n = self.slots
        self.Ahi, self.Alo = self._new((n, 128), BF16), self._new((n, 128), BF16)
        self.scale = self._new((n,), F32)
        if self.lt:
            self.rawkey, self.rawsel = self._new((n, self.CK), F32), self._new((n, self.CK), F32)
            self.rawshr = self._new((n,), F32)
            self.use, self.life = self._new((n,), F32), self._new((n,), F32)
        for o in self.objects:
            self.values[o] = self._new((n, self.CV), BF16)
        self._vptrs = None

    def arrays(self):
        """(tensor, row bytes) of every per-slot array."""
        out = [(self.Ahi, 256), (self.Alo, 256), (self.scale, 4)]
        if self.lt:
            out += [(self.rawkey, 4 * self.CK), (self.rawsel, 4 * self.CK), (self.rawshr, 4), (self.use, 4), (self.life, 4)]
        out += [(v, 2 * self.CV) for v in self.values.values()]
        return out

    def vptrs(self):
        if self._vptrs is None:
            self._vptrs = torch.tensor([self.values[o].data_ptr() for o in self.objects], dtype=torch.int64).to(self.device)
        return self._vptrs

    def ranges(self):
        return [(0, self.n_long), (self.perm_start, self.n_perm), (self.work_start, self.n_work)]

    def size(self):
        return self.n_long + self.n_perm + self.n_work

    def work_order(self):
        """Slot offsets (in frames of HW tokens) of the working tokens in time order, oldest first."""
        nf = self.n_work // self.HW
        if self.lt or nf == 0:
            return list(range(nf))
        frames = self.Wc // self.HW
        if nf < frames:                                  # ring not full yet: written 0, 1, ... in order
            return list(range(nf))
        first = self.ring % frames                       # full ring: the next overwrite hits the oldest frame
        return [(first + i) % frames for i in range(frames)]

    def reserve(self, L=None, P=None, Wc=None):
        """Re-allocate the slabs with new region capacities (rare: permanent memory growing through ``force_permanent`` commits,
        or the memory settings changed by ``update_config``, e.g. the GUI's working / long-term memory sliders).  Long-term and
        permanent tokens are copied as they are; the working tokens are re-written in time order, and when the new working region
        is smaller than what is stored only the newest tokens survive (what the reference's FIFO trim does on the next insertion,
        kv_memory_store.py:206-207)."""
        L = self.L if L is None else L
        P = self.P if P is None else P
        Wc = self.Wc if Wc is None else Wc
        assert L >= self.n_long and P >= self.n_perm, (L, self.n_long, P, self.n_perm)
        if (L, P, Wc) == (self.L, self.P, self.Wc):
            return
        HW = self.HW
        order = self.work_order()
        keep_frames = min(len(order), Wc // HW) if not self.lt else len(order)
        if self.lt:
            assert Wc >= self.n_work, 'long-term mode never drops working tokens outside consolidation'
        order = order[len(order) - keep_frames:]
        old = [(t, rb) for t, rb in self.arrays()]
        oldL, oldP = self.L, self.P
        names = ['Ahi', 'Alo', 'scale'] + (['rawkey', 'rawsel', 'rawshr', 'use', 'life'] if self.lt else [])
        objs = list(self.values.keys())
        self.L, self.P, self.Wc = L, P, Wc
        n = self.slots
        new = []
        for t, _ in old:
            nt = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=self.device)
            nt[:self.n_long] = t[:self.n_long]
            nt[L:L + self.n_perm] = t[oldL:oldL + self.n_perm]
            ws_old, ws_new = oldL + oldP, L + P
            for j, f in enumerate(order):
                nt[ws_new + j * HW:ws_new + (j + 1) * HW] = t[ws_old + f * HW:ws_old + (f + 1) * HW]
            new.append(nt)
        for name, nt in zip(names, new[:len(names)]):
            setattr(self, name, nt)
        for o, nt in zip(objs, new[len(names):]):
            self.values[o] = nt
        self.n_work = keep_frames * HW
        self.ring = keep_frames                          # time order again: the next ring write lands behind the newest frame
        self._vptrs = None
        self._aff_plan = None

    def grow_perm(self, need):
        """A larger permanent region (GUI 'commit' / force_permanent)."""
        P = self.P
        while P < need:
            P *= 2
        self.reserve(P=P)

    def remove_objects(self, keep):
        self.objects = [o for o in self.objects if o in keep]
        self.values = {o: v for o, v in self.values.items() if o in keep}
        self._vptrs = None


class StoreView:
    """Read-only facade with the reference's gauge API (gui/main_controller.py:494-516 reads
    ``memory.work_mem.perm_size(0)`` etc.)."""

    def __init__(self, manager, long_term: bool):
        self._m, self._long = manager, long_term

    @property
    def buckets(self) -> Dict[int, List[int]]:
        if self._long:
            return {b.id: b.objects for b in self._m.buckets.values() if b.n_long > 0}
        return {b.id: b.objects for b in self._m.buckets.values()}

    def size(self, bucket_id: int) -> int:
        b = self._m.buckets.get(bucket_id)
        if b is None:
            return 0
        return b.n_long if self._long else b.n_perm + b.n_work

    def perm_size(self, bucket_id: int) -> int:
        b = self._m.buckets.get(bucket_id)
        return 0 if (b is None or self._long) else b.n_perm

    def non_perm_size(self, bucket_id: int) -> int:
        return self.size(bucket_id) - self.perm_size(bucket_id)

    def engaged(self, bucket_id=None) -> bool:
        if bucket_id is None:
            return len(self.buckets) > 0
        return bucket_id in self.buckets

    @property
    def num_objects(self) -> int:
        return sum(len(v) for v in self.buckets.values())

    def __contains__(self, obj):
        return any(obj in v for v in self.buckets.values())
