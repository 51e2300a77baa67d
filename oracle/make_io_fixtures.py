"""Behaviour fixtures of the reference's VideoReader and ResultSaver, recorded by EXECUTING THE
UNMODIFIED REFERENCE classes (cutie/inference/data/video_reader.py:14-165, cutie/inference/utils/results_utils.py:30-256) on a tiny
generated video folder that is committed with the recordings:

    tests/golden/io/video/...            the input folder (JPEG frames, palette / RGB long-id / greyscale annotations)
    tests/golden/io/video_reader.npz     what VideoReader returns for a set of option combinations
    tests/golden/io/result_saver.json    the PNG / JPEG files ResultSaver writes (bytes + decoded arrays) for a set of calls

tests/test_io_fixtures_cpu.py runs the product classes on the same folder with the same calls and compares.

Two imports of the reference are not installable in this image and are shimmed HERE, in this script only:
  * torchvision.transforms: ToTensor is restated (uint8 HWC -> float CHW / 255, torchvision/transforms/functional.py to_tensor);
    Resize(size: int) is restated with torchvision's semantics (transforms/functional.py resize, _compute_resized_output_size): the
    short side becomes `size`, the long side int(size * long / short); a float TENSOR goes through
    torch.nn.functional.interpolate(mode='bilinear', align_corners=False, antialias=True) (functional_tensor.resize), a PIL image
    through Image.resize(NEAREST) (functional_pil.resize; PIL ignores `antialias`); an input whose short side already equals `size`
    is returned as it is.  The reader resizes the image as a tensor (after ToTensor) and the mask as a PIL image
    (video_reader.py:93-98,118-135);
  * pycocotools.mask: imported by results_utils.py for the BURST json writer only; a stub that raises if used.
hickle (save_scores) is absent as well: that branch is not recorded.

Run in the build container only:  python oracle/make_io_fixtures.py        TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import base64
import enum
import io
import json
import os
import shutil
import sys
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
from oracle.make_golden import GOLDEN, import_reference          # noqa: E402

IO = os.path.join(GOLDEN, 'io')
VIDEO = os.path.join(IO, 'video')
H, W, N = 32, 48, 5

# the option combinations of VideoReader that are recorded: name -> (annotation dir, kwargs)
READER_CASES = {
    'default': ('Annotations', {}),
    'start_end': ('Annotations', dict(start=1, end=4)),
    'start_only': ('Annotations', dict(start=2)),
    'end_only': ('Annotations', dict(end=2)),
    'reverse_to_save': ('Annotations', dict(reverse=True, to_save=['00003'])),
    'enabled_frames': ('Annotations', dict(enabled_frame_list=['00000', '00002', '00004'])),
    'all_masks': ('Annotations', dict(use_all_masks=True)),
    'long_id': ('Annotations_long', {}),
    'greyscale': ('Annotations_L', {}),
    # the `size` path (video_reader.py:93-98,118-135): antialiased bilinear image resize, nearest mask resize
    'resize20': ('Annotations', dict(size=20)),
    'resize20_all_masks': ('Annotations', dict(size=20, use_all_masks=True)),
    'resize27_long_id': ('Annotations_long', dict(size=27)),
    'size_not_smaller': ('Annotations', dict(size=32)),                # min side == size: no resize
    'size_dir': ('Annotations', dict(size=20, size_dir='JPEGImages_half/v')),
}


def install_shims():
    tv = types.ModuleType('torchvision')
    tr = types.ModuleType('torchvision.transforms')

    class InterpolationMode(enum.Enum):
        NEAREST = 'nearest'
        BILINEAR = 'bilinear'

    class ToTensor:
        def __call__(self, pic):
            a = np.array(pic, dtype=np.uint8, copy=True)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)

    class Resize:
        def __init__(self, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=True):
            self.size, self.interpolation, self.antialias = size, interpolation, antialias

        def __call__(self, x):
            assert isinstance(self.size, int)
            if isinstance(x, torch.Tensor):
                h, w = x.shape[-2:]
            else:
                w, h = x.size
            short, long = (w, h) if w <= h else (h, w)
            if short == self.size:
                return x
            new_short, new_long = self.size, int(self.size * long / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
            if isinstance(x, torch.Tensor):
                assert x.is_floating_point() and self.interpolation == InterpolationMode.BILINEAR
                return torch.nn.functional.interpolate(x.unsqueeze(0), size=[nh, nw], mode='bilinear', align_corners=False,
                                                       antialias=bool(self.antialias))[0]
            assert self.interpolation == InterpolationMode.NEAREST
            return x.resize((nw, nh), Image.NEAREST)

    tr.ToTensor, tr.Resize, tr.InterpolationMode = ToTensor, Resize, InterpolationMode
    tv.transforms = tr
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tr
    pc = types.ModuleType('pycocotools')
    pm = types.ModuleType('pycocotools.mask')

    def _unavailable(*a, **k):
        raise NotImplementedError('pycocotools is not installed in this image')

    pm.encode = pm.decode = _unavailable
    pc.mask = pm
    sys.modules['pycocotools'] = pc
    sys.modules['pycocotools.mask'] = pm


def make_video():
    """Deterministic 5-frame clip: smooth colour fields (JPEG-friendly), two moving boxes; annotations of frame 0 and frame 3."""
    if os.path.exists(VIDEO):
        shutil.rmtree(VIDEO)
    for d in ('JPEGImages/v', 'JPEGImages_half/v', 'Annotations/v', 'Annotations_long/v', 'Annotations_L/v'):
        os.makedirs(os.path.join(VIDEO, d))
    yy, xx = np.mgrid[0:H, 0:W]
    pal = []
    for i in range(256):                                           # any fixed palette will do: the reader hands it through
        pal += [(37 * i) % 256, (91 * i + 50) % 256, (173 * i + 11) % 256]
    for t in range(N):
        img = np.stack([(xx * 5 + t * 9) % 256, (yy * 7 + t * 5) % 256, ((xx + yy) * 3 + 40 * t) % 256], -1).astype(np.uint8)
        ids = np.zeros((H, W), dtype=np.uint8)
        ids[4 + t:14 + t, 5 + 2 * t:20 + 2 * t] = 1
        ids[16:30, 22 - t:40 - t] = 2
        img[ids == 1] = (220, 40, 40)
        img[ids == 2] = (40, 60, 230)
        Image.fromarray(img).save(os.path.join(VIDEO, 'JPEGImages/v', f'{t:05d}.jpg'), quality=92)
        Image.fromarray(img[::2, ::2]).save(os.path.join(VIDEO, 'JPEGImages_half/v', f'{t:05d}.jpg'), quality=92)      # (size_dir)
        if t in (0, 3):
            p = Image.fromarray(ids, mode='P')
            p.putpalette(pal)
            p.save(os.path.join(VIDEO, 'Annotations/v', f'{t:05d}.png'))
        if t == 0:
            Image.fromarray(ids * 100, mode='L').save(os.path.join(VIDEO, 'Annotations_L/v', f'{t:05d}.png'))
            long_ids = np.where(ids == 1, 70000, np.where(ids == 2, 300, 0)).astype(np.int64)
            rgb = np.stack([long_ids & 255, (long_ids >> 8) & 255, (long_ids >> 16) & 255], -1).astype(np.uint8)
            Image.fromarray(rgb).save(os.path.join(VIDEO, 'Annotations_long/v', f'{t:05d}.png'))


def saver_probs():
    """Probability stacks [K+1, h, w] for the writer cases (tmp id 1 -> object 2, tmp id 2 -> object 5 / the long ids)."""
    g = torch.Generator().manual_seed(11)
    full = torch.rand((3, H, W), generator=g)
    full[1, 4:14, 5:20] += 1.0
    full[2, 16:30, 22:40] += 1.0
    small = torch.rand((3, H // 2, W // 2), generator=g)           # the resize_needed path: bilinear up-sampling to (H, W) first
    small[1, 2:7, 2:10] += 1.0
    small[2, 8:15, 11:20] += 1.0
    return full, small


# ResultSaver calls that are recorded: name -> (constructor kwargs (beyond output_root / video_name / object_manager), objects,
#                                               process kwargs, which probability stack)
SAVER_CASES = {
    'palette': (dict(dataset='d17-val', use_long_id=False, palette='reader'), [2, 5], {}, 'full'),
    'no_palette': (dict(dataset='generic', use_long_id=False, palette=None), [2, 5], {}, 'full'),
    'long_id': (dict(dataset='generic', use_long_id=True), [300, 70000], {}, 'full'),
    'resized': (dict(dataset='d17-val', use_long_id=False, palette='reader'), [2, 5], dict(resize_needed=True, shape=(H, W)), 'small'),
    'visualize': (dict(dataset='d17-val', use_long_id=False, palette='reader', visualize=True), [2, 5], dict(path_to_image='frame0'), 'full'),
    'visualize_default_colors': (dict(dataset='generic', use_long_id=False, palette=None, visualize=True), [2, 5],
                                 dict(path_to_image='frame0'), 'full'),
    'no_mask_file': (dict(dataset='generic', use_long_id=False, save_mask=False), [2, 5], {}, 'full'),
}


def run_saver_case(ResultSaver, ObjectManager, VideoReader, name, out_root):
    """Shared by the recorder (reference classes) and the test (product classes): returns {relative file: bytes}."""
    kw, objects, pkw, which = SAVER_CASES[name]
    kw, pkw = dict(kw), dict(pkw)
    if kw.get('palette') == 'reader':
        kw['palette'] = VideoReader('v', os.path.join(VIDEO, 'JPEGImages/v'), os.path.join(VIDEO, 'Annotations/v')).get_palette()
    if kw.get('visualize'):
        kw['visualize_output_root'] = os.path.join(out_root, 'vis')
    if pkw.get('path_to_image') == 'frame0':
        pkw['path_to_image'] = os.path.join(VIDEO, 'JPEGImages/v', '00000.jpg')
    om = ObjectManager()
    om.add_new_objects(objects)
    np.random.seed(5)                                              # the long-id writer draws its colours from numpy's global stream
    full, small = saver_probs()
    saver = ResultSaver(os.path.join(out_root, 'masks'), 'v', object_manager=om, **kw)
    saver.process(full if which == 'full' else small, '00000.jpg', **pkw)
    saver.end()
    files = {}
    for base, _, names in os.walk(out_root):
        for n in names:
            p = os.path.join(base, n)
            files[os.path.relpath(p, out_root)] = open(p, 'rb').read()
    return files


def read_case(VideoReader, name):
    """Shared by the recorder and the test: everything a VideoReader hands out for one option combination, as flat arrays."""
    ann, kw = READER_CASES[name]
    kw = dict(kw)
    if 'size_dir' in kw:
        kw['size_dir'] = os.path.join(VIDEO, kw['size_dir'])
    rd = VideoReader('v', os.path.join(VIDEO, 'JPEGImages/v'), os.path.join(VIDEO, ann, 'v'), **kw)
    out = {'len': np.array(len(rd)), 'use_long_id': np.array(bool(rd.use_long_id)),
           'palette': np.array(rd.get_palette() if rd.get_palette() is not None else [], dtype=np.int64)}
    for i in range(len(rd)):
        d = rd[i]
        info = d['info']
        out[f'{i}.rgb'] = d['rgb'].numpy()
        out[f'{i}.frame'] = np.array(info['frame'])
        out[f'{i}.save'] = np.array(bool(info['save']))
        out[f'{i}.shape'] = np.array(list(info['shape']), dtype=np.int64)
        out[f'{i}.resize_needed'] = np.array(bool(info['resize_needed']))
        out[f'{i}.time_index'] = np.array(int(info['time_index']))
        out[f'{i}.path_to_image'] = np.array(os.path.relpath(info['path_to_image'], VIDEO))
        out[f'{i}.has_mask'] = np.array('mask' in d)
        if 'mask' in d:
            out[f'{i}.mask'] = d['mask'].numpy()
            out[f'{i}.mask_dtype'] = np.array(str(d['mask'].dtype))
            out[f'{i}.valid_labels'] = d['valid_labels'].numpy()
    return out


def main():
    import tempfile
    install_shims()
    import_reference()
    from cutie.inference.data.video_reader import VideoReader
    from cutie.inference.object_manager import ObjectManager
    from cutie.inference.utils.results_utils import ResultSaver
    make_video()
    flat = {}
    for name in READER_CASES:
        for k, v in read_case(VideoReader, name).items():
            flat[f'{name}/{k}'] = v
    np.savez_compressed(os.path.join(IO, 'video_reader.npz'), **flat)
    rec = {}
    for name in SAVER_CASES:
        with tempfile.TemporaryDirectory() as tmp:
            files = run_saver_case(ResultSaver, ObjectManager, VideoReader, name, tmp)
        rec[name] = {}
        for rel, data in sorted(files.items()):
            img = Image.open(io.BytesIO(data))
            arr = np.array(img)
            rec[name][rel] = {'bytes_b64': base64.b64encode(data).decode('ascii'), 'mode': img.mode, 'shape': list(arr.shape),
                              'pixels_b64': base64.b64encode(arr.tobytes()).decode('ascii'), 'dtype': str(arr.dtype),
                              'palette': img.getpalette() if img.mode == 'P' else None}
        print(name, {k: (v['mode'], v['shape']) for k, v in rec[name].items()})
    json.dump(rec, open(os.path.join(IO, 'result_saver.json'), 'w'), indent=1, sort_keys=True)
    print('reader cases', len(READER_CASES), 'arrays', len(flat))


if __name__ == '__main__':
    main()
