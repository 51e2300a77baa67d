"""Ordered read-ahead over an indexable frame source: decode (PIL JPEG/PNG, resize) runs on a small thread pool while the GPU
works on earlier frames.  This is what ``DataLoader(vid_reader, batch_size=None, num_workers=4)`` does in the reference's
eval loop (cutie/eval_vos.py:92) -- threads instead of worker processes: PIL and torch release the GIL while decoding /
resizing, nothing has to be pickled, and the items arrive in index order with at most ``depth`` decoded frames alive.
A model at several hundred frames/s is otherwise limited by single-threaded JPEG decode (3-5 ms per 480p frame)."""
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterator, Optional


class ReadAhead:
    def __init__(self, source, *, workers: int = 4, depth: int = 8, length: Optional[int] = None,
                 getitem: Optional[Callable] = None):
        """source[i] for i in range(len(source)) (or ``getitem(i)`` / ``length``).  workers <= 0: read inline."""
        self.get = getitem or source.__getitem__
        self.n = len(source) if length is None else length
        self.workers, self.depth = workers, max(1, depth)

    def __len__(self):
        return self.n

    def __iter__(self) -> Iterator:
        if self.workers <= 0:
            for i in range(self.n):
                yield self.get(i)
            return
        pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix='cutie-read')
        pending = deque()
        try:
            nxt = 0
            while nxt < self.n and len(pending) < self.depth:
                pending.append(pool.submit(self.get, nxt))
                nxt += 1
            while pending:
                item = pending.popleft().result()              # re-raises a reader error at the frame it belongs to
                if nxt < self.n:
                    pending.append(pool.submit(self.get, nxt))
                    nxt += 1
                yield item
        finally:
            for f in pending:
                f.cancel()
            pool.shutdown(wait=True)
