"""Deterministic synthetic clips (numpy PCG64 -> identical on every box / torch version).

Used by bench.py, the parity tests and the golden-vector generator.  The clip is a
smooth random texture translated 2 px/frame (so propagation is non-degenerate), with
K rectangular objects in the first-frame index mask.
"""
import numpy as np
import torch


def _smooth_texture(rng, h, w, cell=16):
    gh, gw = h // cell + 2, w // cell + 2
    grid = rng.random((gh, gw, 3), dtype=np.float32)
    ys = (np.arange(h, dtype=np.float32) + 0.5) / cell
    xs = (np.arange(w, dtype=np.float32) + 0.5) / cell
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    a = grid[y0][:, x0]
    b = grid[y0][:, x0 + 1]
    c = grid[y0 + 1][:, x0]
    d = grid[y0 + 1][:, x0 + 1]
    tex = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    tex = 0.8 * tex + 0.2 * rng.random((h, w, 3), dtype=np.float32)
    return np.clip(tex, 0.0, 1.0).astype(np.float32)


def default_rects(h, w, k):
    """K rectangles scaled from the 480x854 layout of SURVEY.md section 8d (C2)."""
    base = [(100, 300, 100, 300), (150, 400, 400, 600), (50, 200, 650, 800),
            (320, 460, 250, 380), (250, 420, 620, 840)]
    out = []
    for i in range(k):
        y0, y1, x0, x1 = base[i % len(base)]
        out.append((int(y0 * h / 480), max(int(y1 * h / 480), int(y0 * h / 480) + 2),
                    int(x0 * w / 854), max(int(x1 * w / 854), int(x0 * w / 854) + 2)))
    return out


class SyntheticClip:
    def __init__(self, h=480, w=854, num_objects=3, num_frames=500, seed=1, shift=2):
        self.h, self.w, self.k, self.n, self.shift = h, w, num_objects, num_frames, shift
        rng = np.random.Generator(np.random.PCG64(seed))
        period = 64                                    # texture wraps every `period` frames
        self._tex = _smooth_texture(rng, h, w + shift * period)
        self._period = period
        self.rects = default_rects(h, w, num_objects)

    def frame(self, t) -> torch.Tensor:
        """[3,h,w] float32 in [0,1]"""
        tt = t % (2 * self._period)
        if tt >= self._period:                         # bounce back so the motion is continuous
            tt = 2 * self._period - tt
        off = tt * self.shift
        img = self._tex[:, off:off + self.w]
        return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))

    def first_mask(self) -> torch.Tensor:
        """[h,w] int64 index mask, ids 1..K"""
        m = np.zeros((self.h, self.w), dtype=np.int64)
        for i, (y0, y1, x0, x1) in enumerate(self.rects):
            m[y0:y1, x0:x1] = i + 1
        return torch.from_numpy(m)

    @property
    def objects(self):
        return list(range(1, self.k + 1))
