"""``ResultSaver``: probabilities -> object-id masks -> palette PNGs on a writer thread; same surface as the reference
cutie/inference/utils/results_utils.py:30-256 (``process`` / ``end`` / ``make_zip``).

MI355X-side difference: argmax + tmp-id -> object-id remap run as ONE kernel (PROB_TO_ID) that writes uint8 (int32 for
long ids), so the device-to-host copy is H*W bytes instead of the (K+1)*H*W*4 bytes of probabilities, and the host thread
only encodes the PNG.  ``save_scores`` (multi-scale testing, :94,195-209): probabilities are quantised to uint8 (x255,
truncation) ON THE DEVICE, so the copy is 4x smaller too; the reference stores them with hickle (HDF5, lzf), which is not in
this image, so the container here is ``<frame>.npz`` (key ``prob``) and ``backward.npz`` (keys ``obj_ids`` / ``tmp_ids``) --
``cutie_amd.merge_multi_scale`` reads these (and ``.hkl`` when hickle is importable).  Not supported (raise): the BURST
json writer (``init_json``: its RLE masks need pycocotools, which is not in this image).
Long ids (RGB masks): as in the reference (:171-178) every object gets a random colour (utils/pano_utils.ID2RGBConverter), NOT the
inverse of VideoReader's R + 256 G + 65536 B decoding.  Behaviour recorded from the executed reference: tests/golden/io/."""
import logging
import os
import shutil
from dataclasses import dataclass
from os import path
from queue import Queue
from threading import Thread
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from ...utils.pano_utils import ID2RGBConverter

log = logging.getLogger()


def voc_palette(n: int = 256) -> np.ndarray:
    """The PASCAL-VOC / DAVIS colour map [n,3] (bit-interleaving construction)."""
    pal = np.zeros((n, 3), dtype=np.uint8)
    for i in range(n):
        c, r, g, b = i, 0, 0, 0
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal[i] = (r, g, b)
    return pal


davis_palette_np = voc_palette()
davis_palette = davis_palette_np.tobytes()


@dataclass
class _Job:
    saver: 'ResultSaver'
    mask: torch.Tensor            # CPU uint8 / int32 [H,W], object ids
    frame_name: str
    path_to_image: Optional[str]
    all_obj_ids: list
    prob: Optional[torch.Tensor] = None      # CPU uint8 [K+1,H,W] (save_scores)
    last_frame: bool = False
    tmp_to_obj: Optional[dict] = None        # {tmp_id: object id} at the time of the frame


class ResultSaver:
    def __init__(self, output_root, video_name, *, dataset, object_manager, use_long_id, palette=None, save_mask=True,
                 save_scores=False, score_output_root=None, visualize_output_root=None, visualize=False, init_json=None,
                 processor=None):
        """``processor`` (optional, not in the reference): the InferenceCore whose fused PROB_TO_ID kernel does argmax+remap;
        without it a plain torch argmax + lookup is used (e.g. for probabilities that did not come from an InferenceCore)."""
        if save_scores and score_output_root is None:
            raise ValueError('save_scores needs score_output_root')
        self.save_scores, self.score_output_root = save_scores, score_output_root
        if init_json is not None or 'burst' in dataset.lower():
            raise NotImplementedError('the BURST json writer is not supported')
        self.output_root, self.video_name, self.dataset = output_root, video_name, dataset.lower()
        self.use_long_id, self.palette, self.object_manager = use_long_id, palette, object_manager
        self.save_mask, self.visualize, self.visualize_output_root = save_mask, visualize, visualize_output_root
        self.processor = processor
        if self.visualize:
            self.colors = np.array(self.palette, dtype=np.uint8).reshape(-1, 3) if self.palette is not None else davis_palette_np
        self.need_remapping = True
        self.id2rgb_converter = ID2RGBConverter()
        self.queue: Queue = Queue(maxsize=10)
        self.thread = Thread(target=_writer, args=(self.queue,), daemon=True)
        self.thread.start()

    def process(self, prob: torch.Tensor, frame_name: str, resize_needed: bool = False, shape: Optional[Tuple[int, int]] = None,
                last_frame: bool = False, path_to_image: str = None):
        if resize_needed:
            prob = F.interpolate(prob.unsqueeze(1), shape, mode='bilinear', align_corners=False)[:, 0]
        out_dtype = torch.int32 if self.use_long_id else torch.uint8
        if self.processor is not None:
            mask = self.processor.output_prob_to_mask(prob, dtype=out_dtype)          # argmax + remap, one kernel
        else:
            idx = torch.argmax(prob, dim=0)
            lut = torch.zeros(prob.shape[0], dtype=torch.long, device=idx.device)
            for tmp_id, obj in self.object_manager.tmp_id_to_obj.items():
                if tmp_id < lut.shape[0]:
                    lut[tmp_id] = obj.id
            mask = lut[idx].to(out_dtype)
        q = (prob * 255).to(torch.uint8).cpu() if self.save_scores else None       # == numpy astype(uint8) of prob*255
        self.queue.put(_Job(self, mask.cpu(), frame_name, path_to_image, [o.id for o in self.object_manager.obj_to_tmp_id],
                            prob=q, last_frame=last_frame,
                            tmp_to_obj={t: o.id for t, o in self.object_manager.tmp_id_to_obj.items()} if last_frame else None))

    def end(self):
        self.queue.put(None)
        self.queue.join()
        self.thread.join()


def _writer(queue: Queue):
    while True:
        job = queue.get()
        if job is None:
            queue.task_done()
            break
        try:
            s = job.saver
            out_mask = job.mask.numpy()
            rgb_mask = None
            if s.save_mask:
                if s.use_long_id:
                    m = out_mask.astype(np.uint32)
                    rgb_mask = np.zeros((*m.shape[-2:], 3), dtype=np.uint8)
                    for oid in job.all_obj_ids:
                        rgb_mask[m == oid] = s.id2rgb_converter.convert(oid)[1]
                    out_img = Image.fromarray(rgb_mask)
                else:
                    out_img = Image.fromarray(out_mask.astype(np.uint8))
                    if s.palette is not None:
                        out_img.putpalette(s.palette)
                out_dir = path.join(s.output_root, s.video_name)
                os.makedirs(out_dir, exist_ok=True)
                out_img.save(path.join(out_dir, job.frame_name[:-4] + '.png'))
            if s.save_scores:
                sc_dir = path.join(s.score_output_root, s.video_name)
                os.makedirs(sc_dir, exist_ok=True)
                if job.last_frame:                                 # the reference's backward.hkl: {object id: tmp id}
                    ids = sorted(job.tmp_to_obj.items())
                    np.savez(path.join(sc_dir, 'backward.npz'), obj_ids=np.array([o for _, o in ids], dtype=np.int64),
                             tmp_ids=np.array([t for t, _ in ids], dtype=np.int64))
                np.savez_compressed(path.join(sc_dir, job.frame_name[:-4] + '.npz'), prob=job.prob.numpy())
            if s.visualize:
                if job.path_to_image is None:
                    raise ValueError('Cannot visualize without path_to_image')
                image_np = np.array(Image.open(job.path_to_image).convert('RGB'))
                if rgb_mask is None:
                    rgb_mask = np.zeros((*out_mask.shape, 3), dtype=np.uint8)
                    for oid in job.all_obj_ids:
                        rgb_mask[out_mask == oid] = s.colors[oid % len(s.colors)]
                alpha = ((out_mask == 0).astype(np.float32) * 0.5 + 0.5)[:, :, None]
                blend = (image_np * alpha + rgb_mask * (1 - alpha)).astype(np.uint8)
                vis_dir = path.join(s.visualize_output_root, s.video_name)
                os.makedirs(vis_dir, exist_ok=True)
                Image.fromarray(blend).save(path.join(vis_dir, job.frame_name[:-4] + '.jpg'))
        except Exception as e:                                 # keep the queue draining; surface the problem
            log.error(f'result writer failed on {job.frame_name}: {e}')
        queue.task_done()


def make_zip(dataset, run_dir, exp_id, mask_output_root):
    """results_utils.py:233-256: the archive layouts the benchmark servers expect."""
    if dataset.startswith('y') or dataset == 'lvos-test':
        log.info(f'Making zip for {dataset}...')
        shutil.make_archive(path.join(run_dir, f'{exp_id}_{dataset}'), 'zip', run_dir, 'Annotations')
    elif dataset in ('d17-test-dev', 'mose-val'):
        log.info(f'Making zip for {dataset}...')
        shutil.make_archive(path.join(run_dir, f'{exp_id}_{dataset}'), 'zip', mask_output_root)
    else:
        log.info(f'Not making zip for {dataset}.')
