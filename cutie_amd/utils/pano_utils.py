"""Long object ids -> RGB colours for the panoptic-style PNG writer (cutie/utils/pano_utils.py:5-30).

The reference does NOT write a long id back as R + 256 G + 65536 B: every object gets a colour drawn from numpy's global random
stream (``np.random.randint(255, 256**3)``, redrawn while taken), remembered per converter, and the colour's bytes are its RGB,
least significant first.  The product keeps that behaviour -- including the use of the global stream, so a caller that seeds numpy gets
the reference's colours (tests/test_io_fixtures_cpu.py does exactly that)."""
from threading import Lock

import numpy as np


class ID2RGBConverter:
    def __init__(self):
        self.colour_of = {}                     # object id -> drawn 24-bit colour id
        self.taken = set()
        self.lock = Lock()

    @staticmethod
    def _bytes_of(colour: int) -> np.ndarray:
        return np.array([colour & 255, (colour >> 8) & 255, (colour >> 16) & 255], dtype=np.uint8)

    def convert(self, obj: int):
        """-> (colour id, uint8 [3]) of this object, drawn on first use."""
        with self.lock:
            colour = self.colour_of.get(obj)
            while colour is None:
                cand = int(np.random.randint(255, 256 ** 3))
                if cand not in self.taken:
                    colour = self.colour_of[obj] = cand
                    self.taken.add(cand)
        return colour, self._bytes_of(colour)
