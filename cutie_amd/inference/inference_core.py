"""``InferenceCore`` -- per-frame scheduler; same surface and frame logic as the reference
cutie/inference/inference_core.py:18-345, with every tensor op of the frame issued as HIP launch plans.

Kept from the reference (file:line): memory-frame / segmentation / sensory-update schedule (:238-243),
stagger_ti (:37-41), mask merge semantics (:259-300), last_mask = prob[1:] (:302), memorise (:308-315),
unpad (:320), optional internal resize (:206-228, :321-326), delete_objects (:330-335),
output_prob_to_mask (:337-345).
flip_aug (the reference's batch of [frame, flipped frame], :142-143,162-165,234-235,303-305) runs as a second lane with its
own memory bank; chunk_size > 0 groups the objects in the memory read-out like the reference.
"""
import contextlib
import logging
import os
from typing import List, Optional

import numpy as np
import torch

from .. import _lib, frame_context, ops as O
from ..model import plans
from .image_feature_store import ImageFeatureStore
from .memory_manager import MemoryManager
from .object_manager import ObjectManager

log = logging.getLogger()
F32 = torch.float32
AHEAD_AFFINITY = os.environ.get('CUTIE_AMD_AHEAD_AFFINITY', '1') not in ('', '0')     # look-ahead of the affinity read-out (see prefetch)
# memorising on the side stream when no look-ahead hint is given (step / _join_pending): opt-in -- measured neutral on the MI355X (726 against
# 730 fps without hints, profiles/r03_host.md: by the time the host has issued the next frame's encoder the side stream is almost done)
DEFER_MEM = os.environ.get('CUTIE_AMD_DEFER_MEM', '0') not in ('', '0')
# look-ahead WINDOW of the image encoder (step(next_images=...)): frames per batched encoder plan (<= 1: one frame at a time, as with
# next_image), and how many already-encoded frames may be left ahead when the next batch is started
# memory frames in two parts when the next frame is encoded already: its affinity read-out overlaps the summarizer (A/B switch)
MEM_SPLIT = os.environ.get('CUTIE_AMD_MEM_SPLIT', '1') not in ('', '0')
WAIT_TRACE = None                                          # a list: step() brackets its wait for the look-ahead with timing events (diagnostic)
# one affinity read-out per BANK VERSION: the look-ahead read-out covers up to this many announced frames at once -- all frames up to and
# including the next memory frame that sit in one encoder batch of the window (MemoryManager.prefetch_affinity_batch; <= 1: frame by frame)
AFF_BATCH = int(os.environ.get('CUTIE_AMD_AFF_BATCH', '8'))
AFF_FIRST_ALONE = os.environ.get('CUTIE_AMD_AFF_FIRST_ALONE', '1') not in ('', '0')     # (A/B switch: a memory frame's successor read on its own, see _ahead_affinity)
# with AFF_FIRST_ALONE: the stacked pass for the rest of the memory cycle is not queued behind the successor's read-out at the memory frame, but by
# the NEXT step's look-ahead (it then starts behind the memory frame's own tail -- sensory deep update, summarizer -- instead of next to it); A/B switch
AFF_REST_LATER = os.environ.get('CUTIE_AMD_AFF_REST_LATER', '0') not in ('', '0')
WINDOW = int(os.environ.get('CUTIE_AMD_WINDOW', '12'))
WINDOW_LEAD = int(os.environ.get('CUTIE_AMD_WINDOW_LEAD', '3'))


def pad_geometry(h, w, d=16):
    """tensor_utils.pad_divide_by (cutie/utils/tensor_utils.py:7-22): symmetric zero pad to a multiple of d."""
    new_h = h + d - h % d if h % d > 0 else h
    new_w = w + d - w % d if w % d > 0 else w
    lh, uh = int((new_h - h) / 2), int(new_h - h) - int((new_h - h) / 2)
    lw, uw = int((new_w - w) / 2), int(new_w - w) - int((new_w - w) / 2)
    return new_h, new_w, (lw, uw, lh, uh)


def unpad(img: torch.Tensor, pad) -> torch.Tensor:
    """cutie/utils/tensor_utils.py:25-44 (3-D case): a view, no copy."""
    if pad[2] + pad[3] > 0:
        img = img[:, pad[2]:img.shape[1] - pad[3], :]
    if pad[0] + pad[1] > 0:
        img = img[:, :, pad[0]:img.shape[2] - pad[1]]
    return img


class InferenceCore:
    def __init__(self, network, cfg, *, image_feature_store: ImageFeatureStore = None, _lane_of: 'InferenceCore' = None):
        self.network = network
        self.cfg = cfg
        self.mem_every = cfg.mem_every
        stagger_updates = cfg.stagger_updates
        self.chunk_size = cfg.chunk_size
        self.save_aux = cfg.save_aux
        self.max_internal_size = cfg.max_internal_size
        # flip_aug: the reference pushes [frame, flipped frame] through everything as a batch of two; the two batch elements
        # interact only through the averaged prediction, so the flipped one is a second lane (= a nested core sharing the
        # object manager, with its own memory bank / sensory state / feature store) instead of a batch dimension in every plan
        self.flip_aug = bool(cfg.flip_aug) and _lane_of is None
        self._lane_of_other = _lane_of
        # chunk_size: in the mask encoder and the decoder (big_modules.py:141-180,267-302) the reference's object chunks are
        # equivalent to the batched form (the HIP plans always batch; 288 GB of HBM).  In MemoryManager.read
        # (memory_manager.py:169-186) they are not: fusion and object transformer see only the objects of a chunk, which is
        # replicated there (pinned by the `small_chunk` golden scenario).  Not replicated: with flip_aug the reference's chunked
        # decoder views the dim-0 concatenation of per-chunk logits as [2, K, H, W] (big_modules.py:300-302), mixing the two lanes.
        self.curr_ti = -1
        self.last_mem_ti = 0
        if stagger_updates >= self.mem_every:
            self.stagger_ti = set(range(1, self.mem_every + 1))
        else:
            self.stagger_ti = set(np.round(np.linspace(1, self.mem_every, stagger_updates)).astype(int))
        self.object_manager = ObjectManager() if _lane_of is None else _lane_of.object_manager
        self.memory = MemoryManager(cfg=cfg, object_manager=self.object_manager)
        self._flip = InferenceCore(network, cfg, _lane_of=self) if self.flip_aug else None
        self.image_feature_store = ImageFeatureStore(self.network) if image_feature_store is None else image_feature_store
        self.last_mask = None
        self.pad = (0, 0, 0, 0)
        self._enc_stream = None        # side stream of the look-ahead image encoder (prefetch) / of a deferred _add_memory
        self._prefetched = None        # (key of the source frame, prepared image, features, event)
        self._pending_mem = None       # event of an _add_memory still running on the side stream (see step)
        self._pending_refs = None      # its inputs: kept referenced until it is joined (pooled buffers are recycled by reference count)
        self._win_stream = None        # third stream: the batched image encoder of the look-ahead window (prefetch_window)
        self._window = {}              # frame key -> (prepared image, record of CUTIE._encode_window, event, source image, geometry)
        self._hinted = False           # this step came with look-ahead hints (CUTIE.segment forks its last launch only then)
        self._prefetched_rec = None    # the encoder record behind _prefetched (window path): lets _add_memory start that frame's read-out
        self._prefetched_group = None  # window entries of the announced frames, next frame first (candidates of a batched read-out)

    def _engine_stream(self, name, dev):
        """The look-ahead streams belong to the ENGINE (= one set of plan buffers / arenas), not to the processor: two processors that
        drive the same network one after the other (clip after clip, a flip lane) then order their look-ahead work on the shared plan
        buffers by stream order.  (CUTIE.fork() gives a concurrent clip its own engine, hence its own streams.)"""
        eng = self.network.engine()
        if plans.ONE_LANE or eng.one_lane:
            # several clips in flight on one GPU (cutie_amd/parallel.py): the look-ahead lanes of a clip keep their batching (one
            # encoder plan per 12 frames, one read-out per bank version) but run on the clip's own stream -- the other clips are what
            # fills the device next to it, and a clip that brings four streams of its own only competes for the hardware queues
            return torch.cuda.current_stream(dev)
        st = eng.__dict__.setdefault('_streams', {})
        if name not in st:
            st[name] = torch.cuda.Stream(device=dev)
        return st[name]

    def _side_stream(self, dev):
        if self._enc_stream is None:
            self._enc_stream = self._engine_stream('side', dev)
        return self._enc_stream

    def _join_pending(self):
        """Memorising frame t (mask encoder + bank insertion, ~25 % of such a frame) does not feed the frame's returned probabilities;
        when the caller gives no look-ahead hint it is queued on the side stream so that the NEXT frame's image encoder (which does not
        read the bank) overlaps with it.  Everything that reads or writes the bank, the sensory state or the object table joins here."""
        ev = self._pending_mem
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._pending_mem = self._pending_refs = None

    # ---- look-ahead image encoder (no counterpart in the reference) ---------------------------------------------
    @staticmethod
    def _frame_key(image):
        """Identity of an announced frame: its storage (address / shape / strides / dtype) AND the version counter of the tensor -- a
        caller that announces frames and then overwrites one in place (a reused decode buffer) changes the counter, the frame no longer
        matches what was encoded ahead and goes through its own encoder (matched by address alone the OLD frame's features
        were returned).  The counter is shared by all views of a tensor: overwriting any part of a stacked clip tensor drops every
        pending look-ahead of its frames (lost overlap, never a wrong result).  Tensors created under torch.inference_mode carry no
        counter; for those the storage identity is all there is -- do not modify them between the announcement and their step."""
        try:
            version = image._version
        except RuntimeError:                                    # inference tensor: no version counter
            version = None
        return (image.data_ptr(), tuple(image.shape), tuple(image.stride()), image.dtype, version)

    def _prepare_image(self, image):
        """Frame -> (f32 contiguous device tensor carrying the pad geometry, geometry).  Zero-padding to /16 is fused into
        the first kernel of each plan: the raw frame + geometry are handed over instead of a padded copy."""
        h0, w0 = image.shape[-2:]
        H, W, pad = pad_geometry(h0, w0, 16)
        dev = self.network.device
        if image.dtype != F32 or image.device != dev:          # (a frame that is already what the first kernel reads costs no torch call)
            image = image.to(device=dev, dtype=F32)
        if not image.is_contiguous():
            image = image.contiguous()
        frame_context.remember('geometry', image, (h0, w0, H, W, pad[0], pad[2]))     # un-padded frame + pad geometry (see ImageFeatureStore)
        return image, (h0, w0, H, W, pad)

    def prefetch(self, image: torch.Tensor, *, affinity: bool = False) -> None:
        """Optional look-ahead: start the image encoder (ResNet-50 + key projection, ~30 % of a frame and independent of
        the memory state) of the frame that will be passed to the NEXT ``step`` on a side stream, so that it overlaps with
        the read-out / transformer / decoder of the current frame.  ``step(next_image=...)`` calls this.  The hint is matched
        by storage (address / shape / strides), so pass the same tensor (or view) to the next ``step`` and do not modify it
        in between; if the next ``step`` receives a different tensor the result is simply dropped.  Results are bit-identical to the unpipelined
        order (same kernels, same inputs).  ``affinity=True`` (set by ``step`` when the current frame does not write the memory bank)
        also runs the next frame's affinity read-out ahead: it depends on that frame's key and on the bank only."""
        if self.max_internal_size > 0 and min(image.shape[-2:]) > self.max_internal_size:
            return                                             # the GUI resize path stays unpipelined
        dev = self.network.device
        if dev.type != 'cuda':
            return
        src_key = self._frame_key(image)
        main = torch.cuda.current_stream(dev)
        prepared, (h0, w0, H, W, pad) = self._prepare_image(image)      # (conversion, if any, runs on the caller's stream)
        geometry = (h0, w0, H, W, pad[0], pad[2])
        enc = self._side_stream(dev)
        enc.wait_stream(main)                                  # frame conversion + any earlier encoder run on the main stream
        pool = self.network.engine().pool
        pool.offset = 1                                        # these are the NEXT frame's tensors: its slot of the frame-slot pool
        try:
            with torch.cuda.stream(enc):
                ms_features, pix_feat = self.network._encode_image_raw(prepared, *geometry)
                key, shrinkage, selection = self.network.transform_key(ms_features[0])
                ro = self.memory.prefetch_affinity(key, selection, self.network) if affinity else None
                ev = torch.cuda.Event()
                ev.record(enc)
        finally:
            pool.offset = 0
        feats = (ms_features, pix_feat, key, shrinkage, selection)
        ahead = [v[0] for v in (ro or {}).values()]
        for t in list(ms_features) + [pix_feat, key, shrinkage, selection] + ahead + list(self.network._key_cache[1].values()):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)                          # allocated on the side stream, consumed on the main one
        # (the source tensor is kept referenced until the next step: its address cannot be recycled for another frame)
        self._prefetched = (src_key, prepared, feats, ev, image, geometry)

    def prefetch_window(self, images, *, affinity: bool = False) -> None:
        """Look-ahead over SEVERAL frames (no counterpart in the reference): ``images`` are the frames of the following ``step`` calls,
        in order (a list of tensors or one stacked tensor; a video reader / ``bench.py`` knows them).  The image encoder is object- and
        memory-independent, and at 480p one frame gives its stride-16 layers 1620 rows -- a third of one round of workgroups on 256 CUs.
        So whenever no more than WINDOW_LEAD already-encoded frames are left ahead, the next WINDOW frames go through ONE plan on a third
        stream (CUTIE._encode_window: every conv with WINDOW x the rows, tiles of the same K-order class, so every frame's features are
        bit-identical to the one-frame plan's).  The next frame's record then takes the place of ``prefetch``'s result, including the
        look-ahead of its affinity read-out on the side stream (``affinity=True``).  Frames are matched by storage like ``prefetch``;
        wrong hints only lose the overlap."""
        n = len(images)
        if n == 0:
            return
        first = images[0]
        dev = self.network.device
        gpu = dev.type == 'cuda'
        if not gpu and not _lib.get_executor().is_mock:        # (the descriptor interpreter of the tests walks through the same bookkeeping, without streams)
            return
        if WINDOW <= 1 or (self.max_internal_size > 0 and min(first.shape[-2:]) > self.max_internal_size):
            return self.prefetch(first, affinity=affinity)
        on = (lambda st: torch.cuda.stream(st)) if gpu else (lambda st: contextlib.nullcontext())
        main = torch.cuda.current_stream(dev) if gpu else None
        # (host time: the common step finds its next frames encoded -- the first WINDOW_LEAD + 1 keys decide that, the rest of the
        # announcement is only looked at when a batch has to be formed)
        nlook = min(n, WINDOW_LEAD + 1)
        keys = [self._frame_key(images[j]) for j in range(nlook)]
        if not (nlook == WINDOW_LEAD + 1 and all(k in self._window for k in keys) and len(self._window) <= WINDOW + WINDOW_LEAD + 1):
            keys += [self._frame_key(images[j]) for j in range(nlook, min(n, WINDOW + WINDOW_LEAD + 1))]
        # encoded frames that are no longer announced (a changed schedule): dropped -- their buffers are only ever re-written by the
        # window stream itself, in its own order
        if len(keys) > nlook or len(keys) == n:                # (the whole announcement was read)
            for k in [k for k in self._window if k not in keys]:
                del self._window[k]
        ahead = 0
        while ahead < len(keys) and keys[ahead] in self._window:
            ahead += 1
        if ahead <= WINDOW_LEAD and ahead < len(keys):
            todo = []
            for j in range(ahead, min(len(keys), ahead + WINDOW)):
                if keys[j] not in self._window and keys[j] not in [keys[i] for i in todo]:
                    todo.append(j)
            # one geometry per batch (frames of one clip); a frame of another size ends the batch
            shape0 = tuple(images[todo[0]].shape[-2:])
            todo = [j for j in todo if tuple(images[j].shape[-2:]) == shape0]
            preps = [self._prepare_image(images[j]) for j in todo]          # (conversion, if any, runs on the caller's stream)
            h0, w0, H, W, pad = preps[0][1]
            geometry = (h0, w0, H, W, pad[0], pad[2])
            win = None
            if gpu:
                if self._win_stream is None:
                    self._win_stream = self._engine_stream('window', dev)
                win = self._win_stream
                win.wait_stream(main)                          # frame conversions; every read of the output set that is being recycled ...
                if self._enc_stream is not None:
                    win.wait_stream(self._enc_stream)          # ... also those of the look-ahead read-outs on the side stream
            with on(win):
                recs = self.network._encode_window([p[0] for p in preps], *geometry)
                ev = None
                if gpu:
                    ev = torch.cuda.Event()
                    ev.record(win)
            for j, (prepared, _), o in zip(todo, preps, recs):
                for t in o.values():
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(main)
                        if self._enc_stream is not None:
                            t.record_stream(self._enc_stream)
                self._window[keys[j]] = (prepared, o, ev, images[j], geometry)
        ent = self._window.get(keys[0])
        if ent is None:
            return
        # the next frame: its record becomes the pending look-ahead of `step`
        prepared, o, ev, src, geometry = ent
        ms_features, pix_feat, key, shrinkage, selection = self.network._adopt_encoded(o)
        frame_context.remember('geometry', prepared, geometry)
        self._prefetched_group = (images, keys)                 # (the announcement: _ahead_affinity looks further into it when it forms a batch)
        if affinity:
            ev = self._ahead_affinity(key, selection, ev, o, next_mem_ti=self.last_mem_ti + self.mem_every)
        self._prefetched = (keys[0], prepared, (ms_features, pix_feat, key, shrinkage, selection), ev, src, geometry)
        self._prefetched_rec = o

    def _ahead_affinity(self, key, selection, ev, o, next_mem_ti=None, first_alone=False, part=None):
        """The NEXT frame's affinity read-out on the side stream, against the bank as the caller's stream leaves it at this point
        (key / selection: that frame's, ready behind `ev`).  Returns the event `step` has to wait for instead of `ev`.
        Batched (AFF_BATCH > 1, window hints): the bank only changes on memory frames, so the read-outs of ALL announced frames up to and
        including the next memory frame `next_mem_ti` (the schedule of inference_core.py:238 if the caller brings no mask in between: a
        bank that changed after all invalidates them through its version) run as ONE pass over the bank, as far as those frames sit
        behind each other in one encoder batch.  A frame that already carries a read-out of the current bank version is not read again.
        first_alone (a memory frame, whose successor waits for this read-out right behind the insertion): the next frame is read on its
        own, the frames behind it as one batch after it -- they have a whole frame time of slack.
        part (with first_alone; clips in lock step, inference/lockstep.py: the side stream serves several banks): 'first' = only the next
        frame's own read-out, 'rest' = only the stacked pass behind it -- the driver issues every clip's 'first' before any 'rest'."""
        dev = self.network.device
        gpu = dev.type == 'cuda'
        mem = self.memory
        qo = o.get('_qo')
        done = qo.get('_readouts') if qo is not None else None
        if part != 'rest' and done and all(v[1] == mem._version for v in done.values()) and set(done) == set(mem.buckets):
            return qo.get('_readouts_ev') or ev                 # computed by an earlier batch (the consumer waits for that batch's event)
        recs, rest, rest_ev = [o], [], None
        group = []
        if AFF_BATCH > 1 and next_mem_ti is not None and self._prefetched_group is not None:
            n = min(AFF_BATCH, next_mem_ti - self.curr_ti)      # frames curr_ti + 1 .. next_mem_ti read this bank version
            images, keys = self._prefetched_group
            for j in range(min(n, len(images))):                # window entries of the announced frames, in order, as far as they are encoded
                e = self._window.get(keys[j] if j < len(keys) else self._frame_key(images[j]))
                if e is None:
                    break
                group.append(e)
        if group and group[0][1] is o:
            hwp = o['Bhi'].shape[0]

            def behind(r, last):                                # stacked operands: r's rows right behind last's
                return r['Bhi'].data_ptr() == last['Bhi'].data_ptr() + hwp * 256 and r['Blo'].data_ptr() == last['Blo'].data_ptr() + hwp * 256 \
                    and r['cq'].data_ptr() == last['cq'].data_ptr() + hwp * 4
            cand = group[1:]
            if first_alone:
                for e in cand:
                    if rest and (e[2] is not rest_ev or not behind(e[1], rest[-1])):
                        break                                   # another encoder batch: its frames are read when their turn comes
                    if not rest:
                        rest_ev = e[2]
                    rest.append(e[1])
            else:
                for e in cand:
                    if e[2] is not ev or not behind(e[1], recs[-1]):
                        break
                    recs.append(e[1])
        enc = main = None
        if gpu:
            main = torch.cuda.current_stream(dev)
            enc = self._side_stream(dev)
            enc.wait_stream(main)
            if ev is not None:
                enc.wait_event(ev)
        pool = self.network.engine().pool
        pool.offset = 1

        def record():
            if not gpu:
                return None
            e = torch.cuda.Event()
            e.record(enc)
            return e

        def operands(r):
            q = r.get('_qo')
            if q is None:
                q = r['_qo'] = dict(Bhi=r['Bhi'], Blo=r['Blo'], cq=r['cq'], h=r['h'], w=r['w'])
            return q
        held = []
        if part == 'first':
            rest = []
        try:
            with (torch.cuda.stream(enc) if gpu else contextlib.nullcontext()):
                if part == 'rest':
                    pass                                        # (the next frame's own read-out was issued by the 'first' call)
                elif len(recs) > 1:
                    qs = [operands(r) for r in recs]
                    per_frame = mem.prefetch_affinity_batch(qs, self.network, event_factory=record)
                    if per_frame:
                        ev = qs[0].get('_readouts_ev') or ev
                        held += per_frame
                else:
                    ro = mem.prefetch_affinity(key, selection, self.network)
                    if gpu:
                        ev = torch.cuda.Event()
                        ev.record(enc)
                        if ro and qo is not None and qo.get('_readouts') is ro:
                            qo['_readouts_ev'] = ev                 #  `read` orders itself behind the read-out, as it does for a batch
                    held.append(ro or {})
                if rest:
                    if gpu and rest_ev is not None:
                        enc.wait_event(rest_ev)
                    held += mem.prefetch_affinity_batch([operands(r) for r in rest], self.network, event_factory=record) or []
        finally:
            pool.offset = 0
        for pf in held:
            for v in pf.values():
                if isinstance(v[0], torch.Tensor) and v[0].is_cuda:
                    v[0].record_stream(main)
        for r in recs + rest:
            for t in r.values():
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(enc)
        return ev

    def _resize(self, x: torch.Tensor, size, *, nearest: bool = False) -> torch.Tensor:
        """F.interpolate(x[None], size, bilinear align_corners=False | nearest-exact)[0] for f32 [C,H,W] as the RESIZE kernel."""
        dev = self.network.device
        x = x.to(device=dev, dtype=F32)
        if x.stride(2) != 1:
            x = x.contiguous()
        C, H, W = x.shape
        out = torch.empty((C, int(size[0]), int(size[1])), dtype=F32, device=dev)
        ol = O.OpList()
        ol.resize(x, out, C=C, H=H, W=W, OH=out.shape[1], OW=out.shape[2], plane=x.stride(0), ldrow=x.stride(1), nearest=nearest)
        ol.finalize()
        ol.run()
        return out

    def _flip_w(self, x: torch.Tensor, out: torch.Tensor = None, alpha: float = 1.0, beta: float = 0.0) -> torch.Tensor:
        """out = alpha * torch.flip(x, dims=[-1]) + beta * out (FLIP_W kernel); x f32 with a contiguous last dimension."""
        x = x.to(device=self.network.device, dtype=F32)
        if not x.is_contiguous():
            x = x.contiguous()
        if out is None:
            out = torch.empty_like(x)
        W = x.shape[-1]
        ol = O.OpList()
        ol.flip_w(x, out, rows=x.numel() // W, W=W, alpha=alpha, beta=beta)
        ol.finalize()
        ol.run()
        return out

    def clear_memory(self):
        self._join_pending()
        self.curr_ti = -1
        self.last_mem_ti = 0
        ovf = self.memory._scratch.get('overflow')              # (carried over: reported at the next step(end=True), no sync here)
        self.memory._join_side()                                # bookkeeping launches still pending on the look-ahead stream write the old bank's counters
        self.memory = MemoryManager(cfg=self.cfg, object_manager=self.object_manager)
        if ovf is not None:
            self.memory._scratch['overflow'] = ovf
        if self._prefetched is not None:                        # a look-ahead of the old bank: order its buffers, drop it
            if self._prefetched[3] is not None:
                torch.cuda.current_stream(self._prefetched[1].device).wait_event(self._prefetched[3])
            self._prefetched = self._prefetched_rec = self._prefetched_group = None
        if self._flip is not None:
            self._flip.clear_memory()

    def clear_non_permanent_memory(self):
        self._join_pending()
        self.curr_ti = -1
        self.last_mem_ti = 0
        self.memory.clear_non_permanent_memory()
        if self._flip is not None:
            self._flip.clear_non_permanent_memory()

    def clear_sensory_memory(self):
        self._join_pending()
        self.curr_ti = -1
        self.last_mem_ti = 0
        self.memory.clear_sensory_memory()
        if self._flip is not None:
            self._flip.clear_sensory_memory()

    def update_config(self, cfg):
        self._join_pending()
        self.mem_every = cfg['mem_every']
        self.memory.update_config(cfg)
        if self._flip is not None:
            self._flip.update_config(cfg)

    # ---- memorise (inference_core.py:71-121) ------------------------------------------------------------------
    def _add_memory(self, image, pix_feat, prob, key, shrinkage, selection, *, is_deep_update=True, force_permanent=False):
        if prob.shape[1] == 0:
            log.warning('Trying to add an empty object mask to memory!')
            return
        as_permanent = 'all' if force_permanent else 'first'
        ids = self.object_manager.all_obj_ids
        self.memory.initialize_sensory_if_needed(key, ids)
        g = frame_context.recall('geometry', image)            # (h0, w0, H, W, pad_left, pad_top) of the un-padded frame
        raw = (image, g[0], g[1], g[4], g[5]) if g is not None else None
        pre = self._prefetched
        if (MEM_SPLIT and AHEAD_AFFINITY and pre is not None and self._prefetched_rec is not None and self._flip is None
                and self._lane_of_other is None):
            # The next frame is encoded already (window hints): memorise in two parts.  The mask VALUES go into the bank first; then that
            # frame's affinity read-out -- which a memory frame cannot run ahead of its own insertion -- starts on the side stream against
            # the updated bank, while this stream computes the rest of encode_mask (sensory deep update, object summaries: they touch
            # neither the bank nor the read-out).  Same launches, same inputs: bit-identical to the one-part order.
            msk_value, finish = self.network.encode_mask(
                image, pix_feat, self.memory.get_sensory(ids), prob, deep_update=is_deep_update, chunk_size=self.chunk_size,
                need_weights=self.save_aux, _raw=raw, _split=True)
            self.memory.add_memory(key, shrinkage, msk_value, None, ids, selection=selection, as_permanent=as_permanent)
            feats = pre[2]
            ev = self._ahead_affinity(feats[2], feats[4], pre[3], self._prefetched_rec, next_mem_ti=self.curr_ti + self.mem_every, first_alone=AFF_FIRST_ALONE,
                                      part='first' if (AFF_FIRST_ALONE and AFF_REST_LATER) else None)
            self._prefetched = pre[:3] + (ev,) + pre[4:]
            sensory, obj_value = finish()
            self.memory.add_object_values(obj_value, ids)
        else:
            msk_value, sensory, obj_value, _ = self.network.encode_mask(
                image, pix_feat, self.memory.get_sensory(ids), prob, deep_update=is_deep_update, chunk_size=self.chunk_size,
                need_weights=self.save_aux, _raw=raw)
            self.memory.add_memory(key, shrinkage, msk_value, obj_value, ids, selection=selection, as_permanent=as_permanent)
        self.last_mem_ti = self.curr_ti
        if is_deep_update:
            self.memory.update_sensory(sensory, ids)

    # ---- segment (inference_core.py:123-170) ---------------------------------------------------------------------
    def _segment(self, key, selection, pix_feat, ms_features, update_sensory=True) -> torch.Tensor:
        if not self.memory.engaged:
            log.warning('Trying to segment without any memory!')
            return torch.zeros((1, key.shape[-2] * 16, key.shape[-1] * 16), device=key.device, dtype=key.dtype)
        ids = self.object_manager.all_obj_ids
        memory_readout = self.memory.read(pix_feat, key, selection, self.last_mask, self.network)
        stacked = self.memory.readout_stacked(ids)
        memory_readout = stacked if stacked is not None else self.object_manager.realize_dict(memory_readout)
        sensory, _, pred_prob_with_bg = self.network.segment(ms_features, memory_readout, self.memory.get_sensory(ids),
                                                             chunk_size=self.chunk_size, update_sensory=update_sensory,
                                                             _need_logits=False, _fork=self._hinted)
        pred_prob_with_bg = pred_prob_with_bg[0]
        if update_sensory:
            self.memory.update_sensory(sensory, ids)
        return pred_prob_with_bg

    # ---- step (inference_core.py:172-328) ------------------------------------------------------------------------------
    def step(self, image: torch.Tensor, mask: Optional[torch.Tensor] = None, objects: Optional[List[int]] = None, *,
             idx_mask: bool = True, end: bool = False, delete_buffer: bool = True, force_permanent: bool = False,
             next_image: Optional[torch.Tensor] = None, next_images=None) -> torch.Tensor:
        """Same contract as the reference (inference_core.py:172-328).  ``next_image`` (optional, not in the reference): the
        frame of the following ``step``; its image encoder is started on a side stream (see ``prefetch``).  ``next_images``
        (optional): the frames of the following ``step`` calls, in order -- the encoder runs over a window of them at once
        (see ``prefetch_window``); results are bit-identical with or without either hint."""
        if objects is None and mask is not None:
            assert not idx_mask
            objects = list(range(1, mask.shape[0] + 1))

        resize_needed = False
        if self.max_internal_size > 0:
            h, w = image.shape[-2:]
            min_side = min(h, w)
            if min_side > self.max_internal_size:
                # internal-resolution path of the reference (:206-228), RESIZE kernel
                resize_needed = True
                new_h = int(h / min_side * self.max_internal_size)
                new_w = int(w / min_side * self.max_internal_size)
                image = self._resize(image, (new_h, new_w))
                if mask is not None:
                    if idx_mask:
                        mask = self._resize(mask.unsqueeze(0).float(), (new_h, new_w), nearest=True)[0].round().long()
                    else:
                        mask = self._resize(mask, (new_h, new_w))

        self.curr_ti += 1
        if self._lane_of_other is None:
            self.network.engine().pool.tick()                  # frame-slot pool: this frame's slot (plans.SlotPool)
        pre, self._prefetched, self._prefetched_rec = self._prefetched, None, None
        self._prefetched_group = None
        if self._window and not resize_needed and (pre is None or pre[0] != self._frame_key(image)):
            ent = self._window.get(self._frame_key(image))
            if ent is not None:                                # encoded ahead by the window, but not announced as the next frame
                if pre is not None and pre[3] is not None:
                    torch.cuda.current_stream(pre[1].device).wait_event(pre[3])
                pre = (self._frame_key(image), ent[0], self.network._adopt_encoded(ent[1]), ent[2], ent[3], ent[4])
        if pre is not None and not resize_needed and pre[0] == self._frame_key(image):
            # this frame's encoder already ran (or is running) on the side stream
            image = pre[1]
            if pre[3] is not None:
                if WAIT_TRACE is not None:                     # (tools/stream_waits.py: how long does the caller's stream wait for the look-ahead?)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    torch.cuda.current_stream(image.device).wait_event(pre[3])
                    b.record()
                    WAIT_TRACE.append((self.curr_ti, a, b))
                else:
                    torch.cuda.current_stream(image.device).wait_event(pre[3])
            self.image_feature_store._store[self.curr_ti] = pre[2]
            self._window.pop(pre[0], None)                     # (a frame of the look-ahead window: consumed)
            h0, w0, H, W, pl, pt = pre[5]
            frame_context.remember('geometry', image, pre[5])    # (for _add_memory / a third-party consumer of this frame)
            self.pad = pad_geometry(h0, w0, 16)[2]
        else:
            if pre is not None and pre[3] is not None:         # stale look-ahead: order the encoder plan's buffers, drop it
                torch.cuda.current_stream(pre[1].device).wait_event(pre[3])
            image, (h0, w0, H, W, self.pad) = self._prepare_image(image)
            pl, pt = self.pad[0], self.pad[2]

        self._hinted = (next_image is not None or (next_images is not None and len(next_images) > 0)) and self._flip is None
        is_mem_frame = ((self.curr_ti - self.last_mem_ti >= self.mem_every) or (mask is not None)) and (not end)
        need_segment = (mask is None) or (self.object_manager.num_obj > 0 and not self.object_manager.has_all(objects))
        update_sensory = ((self.curr_ti - self.last_mem_ti) in self.stagger_ti) and (not end)

        ms_feat, pix_feat = self.image_feature_store.get_features(self.curr_ti, image)
        key, shrinkage, selection = self.image_feature_store.get_key(self.curr_ti, image)
        self._join_pending()                                   # (the image encoder above overlapped with a deferred _add_memory)
        fl = self._flip
        if fl is not None:
            # the reference flips the PADDED frame (:231-235): flipping the raw frame swaps the left / right pads
            fl.curr_ti, fl.pad = self.curr_ti, self.pad
            image_f = self._flip_w(image)
            frame_context.remember('geometry', image_f, (h0, w0, H, W, self.pad[1], pt))
            ms_f, pix_f = fl.image_feature_store.get_features(self.curr_ti, image_f)
            key_f, shr_f, sel_f = fl.image_feature_store.get_key(self.curr_ti, image_f)
        elif (next_image is not None or (next_images is not None and len(next_images) > 0)) and not end:
            # (the next frame's read-out may run ahead only if this frame leaves the bank alone)
            aff = AHEAD_AFFINITY and not (is_mem_frame or force_permanent) and self.memory.engaged
            if next_images is not None and len(next_images) > 0:
                self.prefetch_window(next_images, affinity=aff)
            else:
                self.prefetch(next_image, affinity=aff)

        if need_segment:
            pred_prob_with_bg = self._segment(key, selection, pix_feat, ms_feat, update_sensory=update_sensory)
            if fl is not None:
                pred_f = fl._segment(key_f, sel_f, pix_f, ms_f, update_sensory=update_sensory)
                if self.memory.engaged:                        # average of the pass and the un-flipped flipped pass (:162-165)
                    pred_prob_with_bg = self._flip_w(pred_f, out=pred_prob_with_bg.contiguous(), alpha=0.5, beta=0.5)

        if mask is not None:
            k_old = self.object_manager.num_obj
            corresponding_tmp_ids, _ = self.object_manager.add_new_objects(objects)
            k_new = self.object_manager.num_obj
            if not need_segment and idx_mask and len(objects) == 0:
                if delete_buffer:
                    self.image_feature_store.delete(self.curr_ti)
                log.warning('Trying to insert an empty mask as memory!')
                return torch.zeros((1, key.shape[-2] * 16, key.shape[-1] * 16), device=key.device, dtype=key.dtype)
            pred_prob_with_bg = self._mask_to_prob(mask, objects, corresponding_tmp_ids, idx_mask, need_segment,
                                                   pred_prob_with_bg if need_segment else None, k_old, k_new,
                                                   h0, w0, H, W, pl, pt)

        self.last_mask = pred_prob_with_bg[1:].unsqueeze(0)
        if fl is not None:
            fl.last_mask = self._flip_w(self.last_mask)        # (:303-305)

        if (is_mem_frame or force_permanent) and DEFER_MEM and fl is None and next_image is None and (next_images is None or len(next_images) == 0) and not end and image.is_cuda \
                and self.last_mask.shape[1] > 0:
            main = torch.cuda.current_stream(image.device)
            side = self._side_stream(image.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._add_memory(image, pix_feat, self.last_mask, key, shrinkage, selection, force_permanent=force_permanent)
                self._pending_mem = torch.cuda.Event()
                self._pending_mem.record(side)
            self._pending_refs = (image, pix_feat, self.last_mask, key, shrinkage, selection)
            for t in self._pending_refs:                       # allocated on main, still read on the side stream
                if isinstance(t, torch.Tensor):
                    t.record_stream(side)
        elif is_mem_frame or force_permanent:
            self._add_memory(image, pix_feat, self.last_mask, key, shrinkage, selection, force_permanent=force_permanent)
            if fl is not None:
                fl._add_memory(image_f, pix_f, fl.last_mask, key_f, shr_f, sel_f, force_permanent=force_permanent)

        if delete_buffer:
            self.image_feature_store.delete(self.curr_ti)
            if fl is not None:
                fl.image_feature_store.delete(self.curr_ti)

        if end:
            self._join_pending()
            self.memory._join_side()
            self.memory.check_overflow()
        output_prob = unpad(pred_prob_with_bg, self.pad)
        if resize_needed:
            output_prob = self._resize(output_prob, (h, w))
        return output_prob

    def _mask_to_prob(self, mask, objects, tmp_ids, idx_mask, need_segment, pred, k_old, k_new, h0, w0, H, W, pl, pt):
        """inference_core.py:259-300 as two kernels (MASK_MERGE, AGG_SOFTMAX)."""
        net = self.network
        dev = net.device
        src = [-1] * k_new
        if need_segment:
            # planes of known objects keep the prediction unless the input mask provides them
            n_planes = max(k_new, pred.shape[0] - 1)
            for mask_id, tmp_id in enumerate(tmp_ids):
                # sic: float masks are indexed by tmp id in the reference (:276) -- and out of range raises there as well
                if not idx_mask and tmp_id >= mask.shape[0]:
                    raise IndexError(f'index {tmp_id} is out of bounds for dimension 0 with size {mask.shape[0]}')
                src[tmp_id - 1] = int(objects[mask_id]) if idx_mask else int(tmp_id)
            k_pred = pred.shape[0] - 1
        else:
            if idx_mask:
                n_planes = len(tmp_ids)
                src = [int(objects[mask_id]) for mask_id, _ in enumerate(tmp_ids)]
            else:
                n_planes = mask.shape[0]
                src = list(range(n_planes))
            k_pred = 0
        src = src[:n_planes] + [-1] * (n_planes - len(src))
        # the mask is padded on its own (inference_core.py:263: pad_divide_by(mask, 16)): a mask one pixel narrower than the frame
        # (examples/masks/judo/00005.png is 480 x 853 against 480 x 854 frames) lands with ITS pad offsets in the same padded size
        mh0, mw0 = int(mask.shape[-2]), int(mask.shape[-1])
        if (mh0, mw0) != (h0, w0):
            mH, mW, mpad = pad_geometry(mh0, mw0, 16)
            if (mH, mW) != (H, W):
                raise RuntimeError(f'The size of the mask ({mh0} x {mw0}, padded {mH} x {mW}) must match the size of the frame '
                                   f'({h0} x {w0}, padded {H} x {W})')
            h0, w0, pl, pt = mh0, mw0, mpad[0], mpad[2]
        if idx_mask:
            inmask = mask.to(device=dev, dtype=torch.int32).contiguous()
            nfloat = 0
        else:
            inmask = mask.to(device=dev, dtype=F32).contiguous()
            nfloat = inmask.shape[0]
        src_t = torch.tensor(src, dtype=torch.int32).to(dev)
        eng = net.engine()
        P = eng.plan(('m2p', n_planes, k_pred, h0, w0, H, W, pl, pt, not idx_mask, nfloat), plans.build_mask_to_prob,
                     n_planes, k_pred, h0, w0, H, W, pl, pt, not idx_mask, nfloat)
        prob = torch.empty((n_planes + 1, H, W), dtype=F32, device=dev)
        P.run(inmask=inmask, pred=pred.contiguous() if pred is not None else None, src=src_t, prob=prob)
        return prob

    def delete_objects(self, objects: List[int]) -> None:
        self._join_pending()
        self.object_manager.delete_objects(objects)
        self.memory.purge_except(self.object_manager.all_obj_ids)
        if self._flip is not None:
            self._flip.memory.purge_except(self.object_manager.all_obj_ids)

    def output_prob_to_mask(self, output_prob: torch.Tensor, *, dtype: torch.dtype = torch.int64) -> torch.Tensor:
        """inference_core.py:337-345: argmax over the probability planes + tmp-id -> object-id remap, as one kernel
        (PROB_TO_ID) that reads the un-padded view `step` returned in place.  ``dtype`` (not in the reference): uint8 /
        int32 / int64 output; the reference returns int64."""
        P, H, W = output_prob.shape
        dev = self.network.device
        lut = [0] * P
        for tmp_id, obj in self.object_manager.tmp_id_to_obj.items():
            if tmp_id < P:
                lut[tmp_id] = int(obj.id)
        if dtype == torch.uint8 and max(lut) > 255:
            raise ValueError('object ids above 255 need dtype=torch.int32 / int64')
        prob = output_prob
        if prob.dtype != F32 or prob.device != dev or prob.stride(2) != 1:
            prob = prob.to(device=dev, dtype=F32).contiguous()
        out = torch.empty((H, W), dtype=dtype, device=dev)
        ol = O.OpList()
        ol.prob_to_id(prob, torch.tensor(lut, dtype=torch.int32).to(dev), out, P=P, H=H, W=W, plane=prob.stride(0), ldrow=prob.stride(1))
        ol.finalize()
        ol.run()
        return out
