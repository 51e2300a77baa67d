"""Multi-scale / flip test-time merge: sum the uint8 score dumps of several runs (``ResultSaver(save_scores=True)`` ->
``<run>/Scores/<video>/<frame>.npz``), argmax, map tmp ids back to object ids, write palette PNGs and the benchmark zip.
Same command line and behaviour as the reference scripts/merge_multi_scale.py:27-139 (``--dataset D|Y``, ``--list`` or
``--pattern``, ``--output``, ``--num_proc``); the score container is .npz (hickle is not available here; .hkl is read when
hickle can be imported).  Host-side tool: no GPU involved.

    python -m cutie_amd.merge_multi_scale --dataset D --list run_480 run_600 --output merged"""
import glob
import os
import shutil
from argparse import ArgumentParser
from collections import defaultdict
from multiprocessing import Pool
from os import path

import numpy as np
from PIL import Image

from .inference.utils.results_utils import davis_palette


def youtube_palette() -> bytes:
    """YouTubeVOS uses the same VOC colour map for its 8-bit PNGs."""
    return davis_palette


def load_scores(fn: str) -> np.ndarray:
    if fn.endswith('.npz'):
        with np.load(fn) as z:
            return z['prob']
    import hickle                                                 # reference dumps
    return hickle.load(fn)


def load_backward(vid_dir: str):
    """{object id: tmp id}, or None when no run wrote it (merge_multi_scale.py:28-32)."""
    fn = path.join(vid_dir, 'backward.npz')
    if path.exists(fn):
        with np.load(fn) as z:
            return {int(o): int(t) for o, t in zip(z['obj_ids'], z['tmp_ids'])}
    fn = path.join(vid_dir, 'backward.hkl')
    if path.exists(fn):
        import hickle
        return hickle.load(fn)
    return None


def merge_video(vid: str, options, out_path: str, dataset: str, palette: bytes) -> int:
    """Frames are enumerated from the first run; runs that lack a frame are skipped for it (:34-52)."""
    backward = None
    for o in options:
        if path.exists(path.join(o, vid)):
            backward = load_backward(path.join(o, vid))
            break
    frames = sorted(f for f in os.listdir(path.join(options[0], vid)) if 'backward' not in f)
    out_dir = path.join(out_path, 'Annotations', vid) if 'Y' in dataset else path.join(out_path, vid)
    os.makedirs(out_dir, exist_ok=True)
    for f in frames:
        total = None
        for o in options:
            fn = path.join(o, vid, f)
            if not path.exists(fn):
                continue
            r = load_scores(fn)
            total = r.astype(np.float32) if total is None else total + r
        idx = np.argmax(total, axis=0)
        if backward is not None:
            out = np.zeros(idx.shape, dtype=np.uint8)
            for obj_id, tmp_id in backward.items():
                out[idx == tmp_id] = obj_id
        else:
            out = idx.astype(np.uint8)
        img = Image.fromarray(out)
        img.putpalette(palette)
        img.save(path.join(out_dir, f[:-4] + '.png'))
    return len(frames)


def _job(a):
    return merge_video(*a)


def merge(options, out_path: str, dataset: str = 'D', num_proc: int = 4) -> int:
    if 'D' in dataset:
        palette = davis_palette
    elif 'Y' in dataset:
        palette = youtube_palette()
    else:
        raise NotImplementedError(dataset)
    options = [path.join(o, 'Scores') for o in options]
    count = defaultdict(int)
    for o in options:
        for vid in sorted(os.listdir(o)):
            count[vid] += 1
    hist = defaultdict(int)
    for v in count.values():
        hist[v] += 1
    for k, v in sorted(hist.items()):
        print('Videos with count %d: %d' % (k, v))
    vids = sorted(count)
    print('Total number of videos: ', len(vids))
    # the first run defines the frame list, so a video must exist there (the reference fails the same way)
    jobs = [(v, options, out_path, dataset, palette) for v in vids]
    if num_proc > 1 and len(jobs) > 1:
        with Pool(processes=num_proc) as pool:
            n = sum(pool.imap_unordered(_job, jobs))
    else:
        n = sum(_job(j) for j in jobs)
    if 'D' in dataset:
        shutil.make_archive(out_path, 'zip', out_path)
    if 'Y' in dataset:
        shutil.make_archive(path.join(out_path, path.basename(out_path)), 'zip', out_path, 'Annotations')
    return n


def main():
    ap = ArgumentParser()
    ap.add_argument('--dataset', default='Y', help='D for DAVIS, Y for YouTubeVOS')
    ap.add_argument('--list', nargs='+')
    ap.add_argument('--pattern', default=None, help='glob pattern, in place of --list')
    ap.add_argument('--output', required=True)
    ap.add_argument('--num_proc', default=4, type=int)
    args = ap.parse_args()
    if args.pattern is None:
        options = args.list
    else:
        assert args.list is None, 'cannot specify both list and pattern'
        options = sorted(glob.glob(args.pattern))
    merge(options, args.output, args.dataset, args.num_proc)


if __name__ == '__main__':
    main()
