"""``LockstepCores`` -- C independent clips advanced in LOCK STEP through one launch plan per facade method.

No counterpart in the reference: it creates one ``InferenceCore`` per video and runs the videos one after another
(cutie/eval_vos.py:97, "one InferenceCore per video" scripting_demo.py:17).  Clips are independent and share every weight, and
within a clip the recurrence forbids batching frames -- so the rows a convolution of the per-object path sees (pixel fusion,
cutie.py:142-157; object transformer, object_transformer.py:114-177; mask decoder, big_modules.py:257-306; mask encoder,
big_modules.py:122-182) are K x HW = 4860 at 480p / 3 objects: fewer 128 x 128 tiles than the MI355X has compute units, and
every launch re-streams its weights for them.  Here the C clips of a group go through ONE plan per stage with
batch = C x K objects (clip-major):

* image encoder + key projection: the look-ahead window of ``InferenceCore`` over the frames of ALL clips as one batched plan
  (``CUTIE._encode_window``; frame-major: the C clips of a lock-step frame are neighbours in every output);
* affinity read-out: one memory bank per clip, all of them on ONE schedule -- the look-ahead read-outs of a bank version run as one
  pass over the stacked frames of all clips (``_ahead_joint`` -> ``MemoryManager.prefetch_affinity_joint``: AFF_SCORE flags&4 /
  AFF_READOUT i8 pick the bank per stacked frame), on the engine's side stream; their result is the stacked input of pixel fusion.
  (``JOINT`` off, or banks that do not line up: every clip's own look-ahead lane, ``InferenceCore._ahead_affinity``, and a gather in
  front of pixel fusion);
* pixel fusion, object transformer, decoder (+ sensory update), mask encoder + summarizer: plans built with ``clips=C``
  (model/plans.py).  What couples the objects of a clip -- the "others" mask, the foreground masks of the transformer, the soft
  aggregation + softmax, the per-clip image features -- is grouped per clip inside the launches (include/cutie_hip.h, ABI 4);
* bank insertion, consolidation, object bookkeeping: each clip's own ``MemoryManager`` / ``ObjectManager``.

Per clip every launch computes what the one-clip plan computes, with conv tiles of the same K-order class (``Plan.korder_ref``), so a
clip's probabilities are bit-identical to its own ``InferenceCore`` run (tests: test_lockstep_clips_match_sequential, CPU + GPU).

A step is batched when every clip is in the plain propagation state (memory engaged, one bucket holding all of its K objects, the
same frame / memory schedule and geometry, no mask, no ``end``); anything else -- the first frame with its masks, ``end=True`` --
runs clip by clip through the cores' own ``step``.
"""
import contextlib
import os
from typing import List, Optional, Sequence

import torch

from .. import _lib, frame_context
from ..model import plans
from ..model.cutie import group_logical
from . import inference_core as IC
from .inference_core import InferenceCore, pad_geometry, unpad

BF16, F32 = torch.bfloat16, torch.float32


class LockstepCores:
    # frames of every clip per batched encoder plan (C x WINDOW frames per plan) / encoded frames left ahead when the next batch starts
    # (defaults set per group size in __init__: clips x WINDOW ~ 12 frames per plan, the batch the conv tile table was swept for)
    WINDOW = int(os.environ.get('CUTIE_AMD_LS_WINDOW', '0'))
    WINDOW_LEAD = int(os.environ.get('CUTIE_AMD_LS_LEAD', '1'))
    # one look-ahead read-out for the banks of ALL clips (MemoryManager.prefetch_affinity_joint: the encoder window is then laid out frame-major, so
    # that frame j of every clip -- and with it the read-outs pixel fusion takes -- is one stacked tensor); off: every clip's own look-ahead lane (A/B switch)
    JOINT = os.environ.get('CUTIE_AMD_LS_JOINT', '1') not in ('', '0')

    def __init__(self, network, cfg, clips: int):
        self.network = network
        self.cfg = cfg
        self.cores = [InferenceCore(network, cfg) for _ in range(clips)]
        if self.WINDOW <= 0:
            self.WINDOW = max(1, round(12 / clips))
        for c, core in enumerate(self.cores):
            core.memory._clip_tag = c
        self._ctx = [frame_context.new_context() for _ in range(clips)]
        self._state = None              # (sens_f32 [C*K,h,w,CS], sens_bf16, objv [C*K,Q,CE+1]): the stacked per-object state of the group
        self._md_valid = False          # Engine.mask_down_bufs hold MASK_DOWN of the current last masks (written by the batched segment)
        self._qtoken = None             # content tokens of the object summaries the transformer's queries were initialised from
        self._win_stream = None
        self.batched_steps = 0          # (diagnostic: how many steps ran as one plan per stage)
        self.joint_passes = 0           # (... read-out passes over the banks of all clips, _ahead_joint)
        self.stacked_steps = 0          # (... steps whose pixel fusion took the clips' read-outs as one tensor)

    def __len__(self):
        return len(self.cores)

    # ---- state of the group --------------------------------------------------------------------------------------------------------
    def _batchable(self, images, masks, end) -> bool:
        if end or masks is not None or len(self.cores) < 2:  # (a group of one clip IS its InferenceCore)
            return False
        c0 = self.cores[0]
        if c0._flip is not None or (c0.chunk_size is not None and c0.chunk_size >= 1) or c0.save_aux:
            return False
        if plans.UNFUSED or not plans.QCHAIN or not plans.SEG_MD or not plans.STEM or IC.DEFER_MEM:     # (A/B switches of the one-clip path that the lock-step plans do not carry)
            return False
        shape = tuple(images[0].shape)
        if c0.max_internal_size > 0 and min(shape[-2:]) > c0.max_internal_size:
            return False
        K = None
        for core, img in zip(self.cores, images):
            mm = core.memory
            if tuple(img.shape) != shape or not mm.engaged or len(mm.buckets) != 1 or core._pending_mem is not None:
                return False
            b = next(iter(mm.buckets.values()))
            ids = core.object_manager.all_obj_ids
            if list(b.objects) != list(ids) or mm._ids != list(ids) or mm._objv_ids != list(ids):
                return False
            K = len(ids) if K is None else K
            if len(ids) != K or K < 1 or K + 1 > 8:
                return False
            if (core.curr_ti, core.last_mem_ti, core.mem_every) != (c0.curr_ti, c0.last_mem_ti, c0.mem_every) or core.stagger_ti != c0.stagger_ti:
                return False
            if core.last_mask is None or core.last_mask.shape[1] != K:
                return False
        return True

    def _stack_state(self, K):
        """The sensory state and the object summaries of the clips as ONE tensor each (clip-major); every clip's MemoryManager keeps
        working on its slice (views: the kernels update the state in place, memory_manager.py:360-375 / :252-271)."""
        mms = [c.memory for c in self.cores]
        st = self._state
        if st is not None and st[0].shape[0] == K * len(mms) and all(
                mm._sens_f32.data_ptr() == st[0][c * K].data_ptr() and mm._sens_bf16.data_ptr() == st[1][c * K].data_ptr()
                and mm._objv.data_ptr() == st[2][c * K].data_ptr() and mm._sens_f32.shape[0] == K for c, mm in enumerate(mms)):
            return st
        sf = torch.cat([mm._sens_f32 for mm in mms], 0).contiguous()
        sb = torch.cat([mm._sens_bf16 for mm in mms], 0).contiguous()
        ov = torch.cat([mm._objv for mm in mms], 0).contiguous()
        for c, mm in enumerate(mms):
            mm._sens_f32, mm._sens_bf16 = sf[c * K:(c + 1) * K], sb[c * K:(c + 1) * K]
            mm._objv = ov[c * K:(c + 1) * K]            # (a new content token: the queries are initialised again)
        self._state = (sf, sb, ov)
        self._qtoken = None
        return self._state

    # ---- look-ahead: the image encoder over a window of frames of all clips ------------------------------------------------------------
    def _encode_batch(self, frames: List[Sequence[torch.Tensor]], keys, *, ahead: bool):
        """frames[c][j], keys[c][j]: frame j of clip c -> one batched encoder plan over all of them, clip-major; every core's look-ahead
        window receives its clip's records (what InferenceCore.prefetch_window builds for one clip)."""
        net, cores = self.network, self.cores
        dev = net.device
        gpu = dev.type == 'cuda'
        G, n = len(cores), len(frames[0])
        preps = []
        for c, core in enumerate(cores):
            with frame_context.context(self._ctx[c]):
                preps.append([core._prepare_image(img) for img in frames[c]])
        h0, w0, H, W, pad = preps[0][0][1]
        geometry = (h0, w0, H, W, pad[0], pad[2])
        fm = self.JOINT                                  # frame-major: frame j of clip c is entry j * G + c of the batch
        flat = [preps[c][j][0] for j in range(n) for c in range(G)] if fm else [preps[c][j][0] for c in range(G) for j in range(n)]
        win = ev = None
        if gpu and ahead:
            main = torch.cuda.current_stream(dev)
            if self._win_stream is None:
                self._win_stream = cores[0]._engine_stream('window', dev)
            win = self._win_stream
            win.wait_stream(main)
            side = cores[0]._side_stream(dev)
            win.wait_stream(side)
        with (torch.cuda.stream(win) if win is not None else contextlib.nullcontext()):
            recs = net._encode_window(flat, *geometry, qrows=128 if fm else 64)       # (whole 128-row blocks per frame: the joint read-out pass)
            if win is not None:
                ev = torch.cuda.Event()
                ev.record(win)
        for c, core in enumerate(cores):
            for j in range(n):
                o = recs[j * G + c] if fm else recs[c * n + j]
                o['_wstride'] = 1 if fm else n          # frames between the clips in this batch's outputs
                if win is not None:
                    for t in o.values():
                        if isinstance(t, torch.Tensor) and t.is_cuda:
                            t.record_stream(main)
                            t.record_stream(side)
                core._window[keys[c][j]] = (preps[c][j][0], o, ev, frames[c][j], geometry)

    def _drop_lookahead(self):
        """Forget what was encoded / read ahead (a step that leaves the lock-step path, a frame that was not announced): the caller's
        stream is ordered behind the look-ahead streams first, so that their buffers may be re-used."""
        dev = self.network.device
        for core in self.cores:
            if dev.type == 'cuda':
                main = torch.cuda.current_stream(dev)
                for e in core._window.values():
                    if e[2] is not None:
                        main.wait_event(e[2])
                if core._enc_stream is not None:
                    main.wait_stream(core._enc_stream)
            core._window.clear()
            core._prefetched = core._prefetched_rec = core._prefetched_group = None

    def _prefetch(self, next_images, affinity: bool):
        """next_images[c] = the frames of clip c's following steps, in order.  Keeps WINDOW_LEAD + 1 .. WINDOW frames of every clip
        encoded ahead (one plan per batch) and, when this frame leaves the banks alone, starts the next frame's read-outs."""
        cores, net = self.cores, self.network
        n = min(len(ni) for ni in next_images)
        if n == 0:
            return
        W_, L_ = max(1, self.WINDOW), self.WINDOW_LEAD
        look = min(n, W_ + L_ + 1)
        keys = [[core._frame_key(ni[j]) for j in range(look)] for core, ni in zip(cores, next_images)]
        for core, ks in zip(cores, keys):
            for k in [k for k in core._window if k not in ks]:
                del core._window[k]
        ahead = 0
        while ahead < look and all(keys[c][ahead] in cores[c]._window for c in range(len(cores))):
            ahead += 1
        if ahead <= L_ and ahead < look:
            todo = list(range(ahead, min(look, ahead + W_)))
            self._encode_batch([[ni[j] for j in todo] for ni in next_images], [[ks[j] for j in todo] for ks in keys], ahead=True)
        for c, core in enumerate(cores):
            core._prefetched_group = (next_images[c], keys[c])          # (the announcement: _ahead_affinity forms its batches from it)
        if not affinity or self._ahead_joint(cores[0].last_mem_ti + cores[0].mem_every, first_alone=False):
            return
        for c, core in enumerate(cores):
            ent = core._window.get(keys[c][0])
            if ent is None:
                continue
            with frame_context.context(self._ctx[c]):
                _, _, key, _, selection = net._adopt_encoded(ent[1])
                core._ahead_affinity(key, selection, ent[2], ent[1], next_mem_ti=core.last_mem_ti + core.mem_every)

    def _ahead_joint(self, next_mem_ti: int, first_alone: bool) -> bool:
        """The look-ahead read-outs of all clips in one pass per bank version (what InferenceCore._ahead_affinity does for one clip, with
        MemoryManager.prefetch_affinity_joint instead of prefetch_affinity_batch): the announced frames up to and including the next
        memory frame, as far as they lie behind each other in one frame-major encoder batch; first_alone (a memory frame): the next frame
        of every clip as a pass of its own -- the next step waits for it -- and the rest of the memory cycle behind it.  False: not
        possible here (the caller lets every clip's own look-ahead lane do it)."""
        if not self.JOINT:
            return False
        cores, net = self.cores, self.network
        c0 = cores[0]
        dev = net.device
        gpu = dev.type == 'cuda'
        grp = [core._prefetched_group for core in cores]
        if IC.AFF_BATCH <= 1 or any(g is None for g in grp):
            return False
        n = min(IC.AFF_BATCH, next_mem_ti - c0.curr_ti, min(len(g[1]) for g in grp))
        frames = []
        for j in range(n):
            ent = [core._window.get(g[1][j]) for core, g in zip(cores, grp)]
            if not self._one_batch(ent) or ent[0][1]['_wstride'] != 1:
                break
            frames.append(ent)
        if not frames:
            return False
        mms = [core.memory for core in cores]
        if any(len(mm.buckets) != 1 or not mm.engaged for mm in mms):
            return False

        def operands(r):
            q = r.get('_qo')
            if q is None:
                q = r['_qo'] = dict(Bhi=r['Bhi'], Blo=r['Blo'], cq=r['cq'], h=r['h'], w=r['w'])
            return q

        def read(ent):                                          # every clip's frame carries a read-out of its bank as it is now
            for e, mm in zip(ent, mms):
                done = e[1].get('_qo', {}).get('_readouts')
                if not done or set(done) != set(mm.buckets) or any(v[1] != mm._version for v in done.values()):
                    return False
            return True
        if read(frames[0]):
            return True
        hwp = frames[0][0][1]['Bhi'].shape[0]
        G = len(cores)

        def behind(ent, last):                                  # one encoder batch, the frame's rows right behind the last frame's
            return ent[0][2] is last[0][2] and ent[0][1]['Bhi'].data_ptr() == last[0][1]['Bhi'].data_ptr() + G * hwp * 256
        runs = [[frames[0]]]
        for ent in frames[1:]:
            if len(runs) == 1 and first_alone:
                runs.append([ent])
            elif behind(ent, runs[-1][-1]):
                runs[-1].append(ent)
            else:
                break                                           # another encoder batch: read when its turn comes
        enc = main = None
        if gpu:
            main = torch.cuda.current_stream(dev)
            enc = c0._side_stream(dev)
            enc.wait_stream(main)
        pool = net.engine().pool
        pool.offset = 1

        def record():
            if not gpu:
                return None
            e = torch.cuda.Event()
            e.record(enc)
            return e
        ok = True
        try:
            with (torch.cuda.stream(enc) if gpu else contextlib.nullcontext()):
                for i, run in enumerate(runs):
                    if gpu and run[0][0][2] is not None:
                        enc.wait_event(run[0][0][2])
                    qs = [operands(e[1]) for ent in run for e in ent]
                    if not IC.MemoryManager.prefetch_affinity_joint(mms, qs, net, event_factory=record, prio=first_alone and i == 0):
                        ok = i > 0                              # (the banks do not line up: nothing was issued)
                        break
                    self.joint_passes += 1
                    if gpu:
                        for ent in run:
                            for e in ent:
                                for v in e[1]['_qo']['_readouts'].values():
                                    v[0].record_stream(main)
                                for t in e[1].values():
                                    if isinstance(t, torch.Tensor) and t.is_cuda:
                                        t.record_stream(enc)
        finally:
            pool.offset = 0
        return ok

    def _one_batch(self, ent) -> bool:
        """The C window entries are frame j of clips 0 .. C-1 of ONE batched encoder plan: clip c's outputs lie c x (frames per clip of
        the batch) frames behind clip 0's -- what the per-clip residual / skip groups of the lock-step plans address."""
        if any(e is None for e in ent):
            return False
        r0 = ent[0][1]
        ws = r0['_wstride']
        for c, e in enumerate(ent):
            r = e[1]
            if r['_wstride'] != ws or e[2] is not ent[0][2]:
                return False
            for name in ('pix_feat', 'fuse_xt', 'f8p', 'f4p'):
                t0, t = r0[name], r[name]
                if t.data_ptr() != t0.data_ptr() + c * ws * t0[0].numel() * t0.element_size():
                    return False
        return True

    # ---- the frame ----------------------------------------------------------------------------------------------------------------------
    def step(self, images: Sequence[torch.Tensor], masks: Optional[Sequence] = None, objects: Optional[Sequence] = None, *,
             end: bool = False, next_images: Optional[Sequence[Sequence[torch.Tensor]]] = None, **kw) -> List[torch.Tensor]:
        """One frame of every clip: ``images[c]`` (and, on frames that bring masks, ``masks[c]`` / ``objects[c]``) as
        ``InferenceCore.step`` takes them; ``next_images[c]`` = the frames of clip c's following steps (optional look-ahead hint, see
        ``InferenceCore.step``).  Returns the C probability tensors ``InferenceCore.step`` would return, bit for bit."""
        cores = self.cores
        assert len(images) == len(cores)
        if not self._batchable(images, masks, end):
            self._drop_lookahead()
            self._md_valid = False
            out = []
            for c, core in enumerate(cores):
                with frame_context.context(self._ctx[c]):
                    out.append(core.step(images[c], None if masks is None else masks[c], None if objects is None else objects[c], end=end, **kw))
            return out
        assert not kw, 'a lock-step frame takes images and look-ahead hints only'
        return self._step_batched(images, next_images)

    def _step_batched(self, images, next_images):
        cores, net = self.cores, self.network
        eng = net.engine()
        dev = net.device
        gpu = dev.type == 'cuda'
        G = len(cores)
        for c, core in enumerate(cores):
            core.curr_ti += 1
            core.memory._clip_tag = c                   # (clear_memory gives a core a new MemoryManager)
        eng.pool.tick()
        # this frame's encoder records: found in the look-ahead windows, or encoded now (one plan for the C frames, on this stream)
        keys = [core._frame_key(img) for core, img in zip(cores, images)]
        ent = [core._window.pop(k, None) for core, k in zip(cores, keys)]
        if not self._one_batch(ent):
            for core, k, e in zip(cores, keys, ent):
                if e is not None:
                    core._window[k] = e
            self._drop_lookahead()
            self._encode_batch([[img] for img in images], [[k] for k in keys], ahead=False)
            ent = [core._window.pop(k) for core, k in zip(cores, keys)]
        if gpu and ent[0][2] is not None:
            torch.cuda.current_stream(dev).wait_event(ent[0][2])
        recs = [e[1] for e in ent]
        prepared = [e[0] for e in ent]
        h0, w0, H, W, pl, pt = ent[0][4]
        feats = []
        for c, core in enumerate(cores):
            core.pad = pad_geometry(h0, w0, 16)[2]
            core._prefetched = core._prefetched_rec = None
            with frame_context.context(self._ctx[c]):
                feats.append(net._adopt_encoded(recs[c]))            # (ms features, pix_feat, key, shrinkage, selection) + the query operands
        c0 = cores[0]
        is_mem_frame = c0.curr_ti - c0.last_mem_ti >= c0.mem_every      # (inference_core.py:238, mask is None and not end)
        update_sensory = (c0.curr_ti - c0.last_mem_ti) in c0.stagger_ti  # (:243)
        K = cores[0].object_manager.num_obj
        sf, sb, ov = self._stack_state(K)
        if next_images is not None and len(next_images) == G:
            self._prefetch(next_images, affinity=IC.AHEAD_AFFINITY and not is_mem_frame)
        # ---- affinity read-out: per clip (its own bank), taken over from the look-ahead lane where that ran against this bank version
        h, w = recs[0]['h'], recs[0]['w']
        readouts = []
        stacked = recs[0]['_qo'].pop('_readouts_joint', None)          # (a joint look-ahead pass: the clips' read-outs are slices of this tensor)
        for c, core in enumerate(cores):
            with frame_context.context(self._ctx[c]):
                vis = core.memory.read_visual(recs[c]['_qo'], h, w, dev, net)
            readouts.append(next(iter(vis.values())))
        if stacked is not None and any(r.data_ptr() != stacked[c * K].data_ptr() or r.shape[0] != K for c, r in enumerate(readouts)):
            stacked = None                                              # (a clip read its bank again: its version had moved on)
        self.stacked_steps += stacked is not None
        # ---- pixel fusion -> object transformer -> decoder: one plan each for the C x K objects
        ws = recs[0]['_wstride']
        m = net.model_cfg
        KT = G * K
        md = self._md_valid
        P = eng.plan(('ls_fuse', G, K, h, w, md, ws, stacked is not None), plans.build_pixel_fusion, K, h, w, True, md, G, ws, stacked is not None)
        fused = eng.pool.get(('ls_fuse', G, K, h, w, eng.devstr), dict(fused=((KT, h, w, m['embed_dim']), BF16, False)), dev)['fused']
        dyn = {'pixel%d' % c: readouts[c] for c in range(G)} if stacked is None else dict(pixel=stacked)
        if not md:
            dyn.update({'last_mask%d' % c: _f32c(core.last_mask[0]) for c, core in enumerate(cores)})
        P.run(sensory_bf16=sb, fuse_xt=recs[0]['fuse_xt'], fused=fused, **dyn)
        token = tuple(core.memory._objv_token for core in cores)
        fresh = token != self._qtoken or not plans.QINIT_SKIP
        self._qtoken = token
        P = eng.plan(('ls_rq', G, K, h, w, fresh), plans.build_readout_query, K, h, w, False, fresh, G)
        out = eng.pool.get(('ls_rq', G, K, h, w, eng.devstr), dict(out=((KT, h, w, m['embed_dim']), BF16, False)), dev)['out']
        P.run(pixel=fused, obj_mem=ov, out=out)
        P = eng.plan(('ls_seg', G, K, h, w, bool(update_sensory), ws), plans.build_segment, K, h, w, bool(update_sensory), True, True, G, ws)
        prob = eng.pool.get(('ls_seg', G, K, h, w, eng.devstr), dict(prob=((G, K + 1, 16 * h, 16 * w), F32, False)), dev)['prob']
        P.run(p16=out, sensory_f32=sf, sensory_bf16=sb, prob=prob, logits_up=None, f8p=recs[0]['f8p'], f4p=recs[0]['f4p'])
        self._md_valid = True
        for c, core in enumerate(cores):
            core.last_mask = prob[c, 1:].unsqueeze(0)                   # (inference_core.py:302)
        if is_mem_frame:
            self._memorise(prepared, recs, feats, prob, (h0, w0, H, W, pl, pt), K, next_images)
        self.batched_steps += 1
        return [unpad(prob[c], core.pad) for c, core in enumerate(cores)]

    def _memorise(self, prepared, recs, feats, prob, geometry, K, next_images):
        """Memory frame (inference_core.py:71-121, :308-315) of all clips: the mask encoder once for the C x K objects, then every clip's
        bank insertion, then -- while this stream goes on with the sensory deep update and the object summaries -- the next frame's
        read-outs against the new banks on the side stream (the two-part order of InferenceCore._add_memory)."""
        cores, net = self.cores, self.network
        eng = net.engine()
        dev = net.device
        G = len(cores)
        h0, w0, H, W, pl, pt = geometry
        h, w = H // 16, W // 16
        m = net.model_cfg
        sf, sb, _ = self._state
        ws = recs[0]['_wstride']
        KT = G * K
        md = self._md_valid and plans.SUM_FUSED
        P = eng.plan(('ls_emask', G, K, h0, w0, H, W, pl, pt, md, ws), plans.build_encode_mask, K, h0, w0, H, W, pl, pt, True, md, G, ws)
        o = eng.pool.get(('ls_emask', G, K, h, w, eng.devstr),
                         dict(value=((KT, h, w, m['value_dim']), BF16, False),
                              summ=((KT, m['object_summarizer']['num_summaries'], m['embed_dim'] + 1), F32, False)), dev)
        value, summ = o['value'], o['summ']
        dyn = dict(masks=prob, pix_feat=recs[0]['pix_feat'], sensory_f32=sf, sensory_bf16=sb, value=value, summ=summ,
                   **{'image%d' % c: prepared[c] for c in range(G)})
        cut = P.meta['value_done']
        P.run_part(0, cut, **dyn)
        for c, core in enumerate(cores):
            ids = core.object_manager.all_obj_ids
            _, _, key, shrinkage, selection = feats[c]
            with frame_context.context(self._ctx[c]):
                core.memory.add_memory(key, shrinkage, group_logical(value[c * K:(c + 1) * K]), None, ids, selection=selection, as_permanent='first')
        if next_images is not None and IC.AHEAD_AFFINITY and IC.MEM_SPLIT:
            # every clip's next frame first (the next step waits for them), the stacked passes for the rest of the memory cycle behind them
            nxt = []
            for c, core in enumerate(cores):
                grp = core._prefetched_group
                ent = core._window.get(grp[1][0]) if grp is not None and len(grp[1]) else None
                if ent is not None:
                    with frame_context.context(self._ctx[c]):
                        f = net._adopt_encoded(ent[1])
                    nxt.append((c, core, ent, f))
            joint = len(nxt) == G and self._ahead_joint(cores[0].curr_ti + cores[0].mem_every, first_alone=IC.AFF_FIRST_ALONE)
            for part in (() if joint else ('first', 'rest') if IC.AFF_FIRST_ALONE else (None,)):
                for c, core, ent, f in nxt:
                    with frame_context.context(self._ctx[c]):
                        core._ahead_affinity(f[2], f[4], ent[2], ent[1], next_mem_ti=core.curr_ti + core.mem_every,
                                             first_alone=IC.AFF_FIRST_ALONE, part=part)
        P.run_part(cut, None, first=False, **dyn)
        for c, core in enumerate(cores):
            ids = core.object_manager.all_obj_ids
            with frame_context.context(self._ctx[c]):
                core.memory.add_object_values(summ[c * K:(c + 1) * K].unsqueeze(0), ids)
            core.last_mem_ti = core.curr_ti

    def output_prob_to_mask(self, probs: Sequence[torch.Tensor], **kw) -> List[torch.Tensor]:
        return [core.output_prob_to_mask(p, **kw) for core, p in zip(self.cores, probs)]

    # ---- the rest of InferenceCore's surface, clip by clip (inference_core.py:52-69, :330-335) ------------------------------------------
    def __getitem__(self, c: int) -> InferenceCore:
        return self.cores[c]

    def delete_objects(self, objects: Sequence[Sequence[int]]) -> None:
        """objects[c]: the ids to delete from clip c (the group leaves the batched path for a step: the last masks still carry their planes)."""
        for c, (core, ids) in enumerate(zip(self.cores, objects)):
            if ids:
                with frame_context.context(self._ctx[c]):
                    core.delete_objects(list(ids))

    def _each(self, method, *a, **k):
        self._drop_lookahead()
        self._md_valid = False
        for c, core in enumerate(self.cores):
            with frame_context.context(self._ctx[c]):
                getattr(core, method)(*a, **k)

    def clear_memory(self):
        self._each('clear_memory')

    def clear_non_permanent_memory(self):
        self._each('clear_non_permanent_memory')

    def clear_sensory_memory(self):
        self._each('clear_sensory_memory')

    def update_config(self, cfg):
        self._each('update_config', cfg)


def _f32c(t):
    if t.dtype != F32:
        t = t.to(F32)
    return t if t.is_contiguous() else t.contiguous()
