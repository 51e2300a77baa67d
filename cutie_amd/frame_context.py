"""Per-process registry of what the HIP path knows about a tensor it handed to the caller.

The facade methods of ``CUTIE`` keep the reference's tensor-in / tensor-out signatures, but some results have companions the next
method needs: the similarity operands of a key (split-bf16 ``B`` matrices), the bf16 shadow of the fp32 sensory state.  They used
to ride on the tensors as Python attributes, which breaks as soon as a caller re-wraps a tensor (``key = key.clone()``,
``key = key * 1``).  Here they are looked up by storage address instead -- any view of the buffer finds them -- and every consumer
has a slow path that recomputes the companion when nothing is registered (a third-party caller of the six ``CUTIE`` methods).

An entry holds references to its payload tensors, so the address it is keyed by cannot be recycled for another tensor while the
entry lives; the table is a small LRU (frames in flight x a handful of entries) PER HOST THREAD: a clip is driven by one thread
(cutie_amd/parallel.py), and a table shared by several clips in flight let a fast clip evict the entries a slow one was about to
look up.  (An entry made on one thread and looked up on another is simply not found: slow path.)"""
import threading
from collections import OrderedDict

_MAX = 96
_local = threading.local()


def _table():
    t = getattr(_local, 'table', None)
    if t is None:
        t = _local.table = OrderedDict()          # (kind, data_ptr) -> (payload, pinned tensors)
        _local.capped = {}                        # kind -> OrderedDict of this kind's keys, oldest first (kinds remembered with a cap)
    return t


def new_context():
    """An empty table of its own for `context`: one per clip when ONE host thread drives several clips in turn
    (cutie_amd/parallel.py:run_interleaved) -- the per-kind caps are sized for the frames in flight of one clip."""
    return (OrderedDict(), {})


class context:
    """with context(ctx): the calling thread's table is `ctx` (from `new_context`) inside the block."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        _table()
        self.old = (_local.table, _local.capped)
        _local.table, _local.capped = self.ctx
        return self

    def __exit__(self, *exc):
        _local.table, _local.capped = self.old
        return False


def remember(kind, tensor, payload, pin=(), cap=None):
    """Attach `payload` to the storage address of `tensor` (a later entry for the same address replaces it).  cap: keep at most this
    many entries of `kind` (large payloads -- whole feature maps -- of which only the last frames can still be asked for)."""
    key = (kind, tensor.data_ptr())
    t = _table()
    t.pop(key, None)
    if cap is not None:
        ko = _local.capped.get(kind)
        if ko is None:
            ko = _local.capped[kind] = OrderedDict()
        ko.pop(key, None)
        ko[key] = None
        while len(ko) > cap:                      # (the order of a kind's keys, kept beside the table: no scan of the table per call)
            t.pop(ko.popitem(last=False)[0], None)
    t[key] = (payload, (tensor,) + tuple(pin))
    while len(t) > _MAX:
        old = t.popitem(last=False)[0]
        ko = _local.capped.get(old[0])
        if ko is not None:
            ko.pop(old, None)


def recall(kind, tensor):
    """Payload registered for this tensor's storage address (None if unknown: the caller takes its slow path)."""
    hit = _table().get((kind, tensor.data_ptr()))
    return None if hit is None else hit[0]


def forget(kind, tensor):
    key = (kind, tensor.data_ptr())
    if _table().pop(key, None) is not None:
        ko = _local.capped.get(kind)
        if ko is not None:
            ko.pop(key, None)
