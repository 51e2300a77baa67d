"""``MemoryManager`` -- sensory / object / working / long-term memory on MI355X; mirrors the surface of
cutie/inference/memory_manager.py:14-383 on top of the contiguous bank of kv_memory_store.py.

Reference behaviour kept (with file:line):
  * buckets = objects first seen in the same frame (kv_memory_store.py:96-117); first insertion of a bucket is
    permanent ('first'), ``force_permanent`` makes every insertion permanent ('all')   (:119-130)
  * FIFO working memory of ``max_mem_frames-1`` frames (memory_manager.py:38,296)
  * long-term memory: usage counting (:151-162), consolidation of the oldest frames into ``num_prototypes``
    prototypes by top-usage + dense-softmax potentiation (:309-358), pruning of obsolete prototypes (:287-291)
  * object memory = streaming sum of the summaries (:252-271); sensory memory per object (:360-375)
Per-object state lives in stacked tensors in tmp-id order (objects of one bucket are always contiguous there).
"""
import itertools
import logging
import os
from typing import Dict, List

import torch

from .. import frame_context, ops as O
from .kv_memory_store import Bucket, StoreView, LIFE_EPS_BITS
from .object_manager import ObjectManager

log = logging.getLogger()
BF16, F32 = torch.bfloat16, torch.float32
CAND_CAP = 1024          # candidate slots per query column (typical fill ~35; see csrc/affinity.hip)
_UNFUSED = os.environ.get('CUTIE_AMD_UNFUSED', '0') not in ('', '0')      # diagnostic A/B switch, see model/plans.py
_VALIDATE = os.environ.get('CUTIE_AMD_VALIDATE', '0') not in ('', '0')
BANK_WRITE = os.environ.get('CUTIE_AMD_BANK_WRITE', '1') not in ('', '0')      # the copies / fills of an insertion in one launch (A/B switch)
BATCH_FORMS = os.environ.get('CUTIE_AMD_AFF_BATCH_FORMS', '1') not in ('', '0')   # stacked read-outs pick their score kernels by frame count (A/B switch)
JOINT_DMA = int(os.environ.get('CUTIE_AMD_JOINT_DMA', '0'))      # clips in lock step: the joint read-out stages its memory tiles by LDS-DMA in the score pass (1) / the candidate pass (2) (A/B switch)
PATCH_PLANS = os.environ.get('CUTIE_AMD_PATCH_PLANS', '1') not in ('', '0')    # sizes that a memory frame changes are patched into the cached affinity / commit plans (off: rebuilt; A/B + test switch)
COMMIT_ON_SIDE = os.environ.get('CUTIE_AMD_COMMIT_SIDE', '1') not in ('', '0')  # bookkeeping of a consumed look-ahead read-out on the look-ahead stream (A/B switch)
# bank versions are drawn from one process-wide counter: a look-ahead read-out tagged with the version of one manager can never pass
# the check of another (InferenceCore.clear_memory replaces the manager; per-manager counters would restart at 0 and collide)
_VERSIONS = itertools.count(1)


class MemoryManager:
    def __init__(self, cfg, object_manager: ObjectManager):
        self.object_manager = object_manager
        self.sensory_dim = cfg.model.sensory_dim
        self.top_k = cfg.top_k
        self.chunk_size = cfg.chunk_size
        self.save_aux = cfg.save_aux
        self.use_long_term = cfg.use_long_term
        self.count_long_term_usage = cfg.long_term.count_usage
        self._read_cfg(cfg)
        self.CK = self.CV = None
        self.H = self.W = None
        self.buckets: Dict[int, Bucket] = {}
        self._next_bucket = 0
        self.work_mem = StoreView(self, False)
        self.long_mem = StoreView(self, True) if self.use_long_term else None
        # stacked per-object state, rows in the order of self._ids
        self._ids: List[int] = []
        self._sens_f32 = None        # [K,h,w,CS]
        self._sens_bf16 = None
        self._objv = None            # [K,Q,C+1] f32 (a property: every assignment takes a new content token, see _objv_token)
        self._objv_ids: List[int] = []
        self._orphan_objv = {}       # summaries of deleted objects (the reference never purges obj_v, :298-307)
        self._scratch = {}
        self._commit_plans = {}      # bucket id -> (signature, one-launch OpList) of _commit_ahead
        self.config_stale = True
        self.engaged = False
        self.aux = None
        self._version = next(_VERSIONS)   # replaced whenever the bank changes (invalidates look-ahead read-outs)
        self._ahead_parity = 0
        self._batch_parity = 0       # (the stacked read-outs alternate between two sets of usage side buffers of their own)
        self._clip_tag = 0           # which clip of a lock-step group this bank belongs to (inference/lockstep.py): its own read-out slots in the engine's pool

    # The object summaries change only when a frame is memorised (or objects are purged); the transformer's query initialisation
    # depends on nothing else, so CUTIE.readout_query skips it while the token it is handed stays the same (tokens are process-wide
    # unique: an engine may serve several managers one after the other).
    _TOKENS = itertools.count(1)

    @property
    def _objv(self):
        return self.__dict__.get('_objv_t')

    @_objv.setter
    def _objv(self, v):
        self.__dict__['_objv_t'] = v
        self._objv_token = next(MemoryManager._TOKENS)

    def _read_cfg(self, cfg):
        if self.use_long_term:
            self.max_mem_frames = cfg.long_term.max_mem_frames - 1
            self.min_mem_frames = cfg.long_term.min_mem_frames - 1
            self.num_prototypes = cfg.long_term.num_prototypes
            self.max_long_tokens = cfg.long_term.max_num_tokens
            self.buffer_tokens = cfg.long_term.buffer_tokens
        else:
            self.max_mem_frames = cfg.max_mem_frames - 1

    def update_config(self, cfg) -> None:
        self._join_side()
        self.config_stale = True
        self._version = next(_VERSIONS)
        self.top_k = cfg['top_k']
        assert self.use_long_term == cfg.use_long_term, 'cannot update this'
        assert self.count_long_term_usage == cfg.long_term.count_usage, 'cannot update this'
        self._read_cfg(cfg)

    # ---- sensory memory (memory_manager.py:360-375) --------------------------------------------------
    def initialize_sensory_if_needed(self, sample_key: torch.Tensor, ids: List[int]):
        new = [o for o in ids if o not in self._ids]
        if not new:
            return
        _, _, h, w = sample_key.shape
        dev = sample_key.device
        zf = torch.zeros((len(new), h, w, self.sensory_dim), dtype=F32, device=dev)
        zb = torch.zeros((len(new), h, w, self.sensory_dim), dtype=BF16, device=dev)
        if self._sens_f32 is None or not self._ids:
            self._sens_f32, self._sens_bf16 = zf, zb
        else:
            self._sens_f32 = torch.cat([self._sens_f32, zf], 0)
            self._sens_bf16 = torch.cat([self._sens_bf16, zb], 0)
        self._ids = self._ids + new

    def _rows(self, ids: List[int], order: List[int]):
        """(start, stop) of `ids` inside the stacked state ordered by `order`; ids must be a contiguous run."""
        pos = [order.index(o) for o in ids]
        assert pos == list(range(pos[0], pos[0] + len(pos))), 'objects of a bucket must be contiguous in tmp-id order'
        return pos[0], pos[0] + len(pos)

    def get_sensory(self, ids: List[int]):
        """logical [1,K,CS,h,w] fp32 view of the stacked state (+ its bf16 shadow as attribute)."""
        a, b = self._rows(ids, self._ids)
        frame_context.remember('sensory_bf16', self._sens_f32[a:b], self._sens_bf16[a:b])     # found again by CUTIE.segment / encode_mask
        return self._sens_f32[a:b].permute(0, 3, 1, 2).unsqueeze(0)

    def update_sensory(self, sensory: torch.Tensor, ids: List[int]):
        a, b = self._rows(ids, self._ids)
        phys = sensory[0].permute(0, 2, 3, 1)
        if phys.data_ptr() != self._sens_f32[a:b].data_ptr():          # foreign tensor: copy in
            self._sens_f32[a:b].copy_(phys)
            self._sens_bf16[a:b].copy_(phys)
        # else: the kernels already updated the state in place

    def clear_sensory_memory(self):
        self._ids, self._sens_f32, self._sens_bf16 = [], None, None

    # ---- helpers ------------------------------------------------------------------------------------------
    def _buf(self, name, shape, dtype, dev):
        t = self._scratch.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != dev:
            t = torch.zeros(shape, dtype=dtype, device=dev)
            self._scratch[name] = t
        return t

    def _get_mask_by_ids(self, mask: torch.Tensor, obj_ids: List[int]) -> torch.Tensor:
        tmp = [self.object_manager.find_tmp_by_id(o) - 1 for o in obj_ids]
        if tmp == list(range(tmp[0], tmp[0] + len(tmp))):
            return mask[:, tmp[0]:tmp[0] + len(tmp)]
        return mask[:, tmp]

    # ---- read (memory_manager.py:112-208) ---------------------------------------------------------------------

    def _buf_rows(self, name, rows, row_shape, dtype, dev):
        """Scratch of at least `rows` rows (grow-only: the frame count of a batched read-out varies between 1 and mem_every)."""
        t = self._scratch.get(name)
        if t is None or t.shape[0] < rows or tuple(t.shape[1:]) != tuple(row_shape) or t.device != dev:
            t = torch.zeros((rows,) + tuple(row_shape), dtype=dtype, device=dev)
            self._scratch[name] = t
        return t[:rows]

    def _affinity(self, bucket: Bucket, q, h: int, w: int, dev, ahead: bool = False, frames: int = 1):
        """Affinity read-out of one bucket for the query operands q: similarity -> exact top-k -> softmax -> sparse value gather
        (+ usage bookkeeping), 4 launches on the current stream.  Returns readout bf16 [K, h, w, CV].
        ahead (the look-ahead lane of `prefetch_affinity`, which runs on another stream, possibly for a frame that is never read): a
        second set of scratch buffers, and NO bookkeeping on the bank -- the life counters are left alone and the usage of this read-out
        is accumulated into a side buffer (cleared by the selection launch); `_commit_ahead` applies both on the caller's stream when,
        and only when, the read-out is consumed.
        frames > 1 (ahead only; `prefetch_affinity_batch`): q are the operands of the FIRST of `frames` consecutive frames of one
        encoder batch (their Bhi / Blo / cq rows follow each other in memory, HWp rows per frame): one pass over the bank for all of
        them.  Returns [(readout, usage side buffer)] per frame instead of one read-out."""
        if frames > 1:
            return self._affinity_batch(bucket, q, h, w, dev, frames)
        tag = '#ahead' if ahead else ''
        HW = h * w
        HWp = q['Bhi'].shape[0]
        K = len(bucket.objects)
        ranges = [r for r in bucket.ranges() if r[1] > 0]
        G = sum(-(-n // 16) for _, n in ranges)
        # pass-0 tile maxima [HWp, Gld] with the per-query thresholds right behind them (pass 1 reads both: it skips the tiles
        # that cannot hold a candidate)
        Gld = -(-max(G, 1) // 64) * 64
        # (grow-only: the tile count changes with every memorised frame, and an exact-size buffer was re-allocated -- and zero-filled, 5-10 MB
        # -- each time its padded width crossed a multiple of 64 tiles; pass 0 writes every entry that the selection and pass 1 read)
        gbuf = self._buf_rows('gmax_tau' + tag, HWp * Gld + HWp, (), F32, dev)
        gmax, tau = gbuf[:HWp * Gld], gbuf[HWp * Gld:]
        cval = self._buf('cand_val' + tag, (HW, CAND_CAP), F32, dev)
        cidx = self._buf('cand_idx' + tag, (HW, CAND_CAP), torch.int32, dev)
        count = self._buf('count' + tag, (HW * O.OpList.AFF_CSTRIDE,), torch.int32, dev)
        ovf = self._buf('overflow', (1,), torch.int32, dev)
        pool = getattr(self, '_pool', None)                      # (set by read / prefetch_affinity: the engine's frame-slot pool)
        readout = (pool.get(('readout', self._clip_tag, bucket.id, K, h, w, str(dev)), dict(r=((K, h, w, self.CV), BF16, False)), dev)['r'] if pool is not None
                   else torch.empty((K, h, w, self.CV), dtype=BF16, device=dev))
        # The affinity plan of a bucket only changes when its token ranges do (memory frames, consolidation, purge) or a
        # setting is updated: the descriptors are built once per such state with named pointer slots and re-bound per frame
        # (host time matters once several clips share one interpreter).
        tick_work = self.use_long_term and bucket.n_work > 0
        tick_long = self.use_long_term and bucket.n_long > 0 and self.count_long_term_usage
        nslots = int(bucket.use.shape[0]) if self.use_long_term else 0
        clear_long = self.use_long_term and bucket.n_long > 0 and not self.count_long_term_usage
        # `key`: what decides WHICH launches the plan holds; `vals`: the sizes inside them, which change with every memorised frame --
        # those are patched into the descriptors (host time: a rebuild is ~0.1 ms of Python on every memory frame)
        flat = [v for r in ranges for v in r] + [0] * (6 - 2 * len(ranges))
        key = (len(ranges), K, HW, HWp, self.top_k, self.use_long_term, tick_work, tick_long, clear_long, bucket.work_start, nslots)
        vals = (tuple(flat), G, bucket.n_work, bucket.n_long)
        plans_ = bucket.__dict__.setdefault('_aff_plans', {})
        cached = plans_.get(ahead)
        if cached is not None and cached[0] == key and (cached[1] == vals or (PATCH_PLANS and not _UNFUSED)):
            if cached[1] != vals:
                ol, ops_ = cached[2], cached[3]
                for op in (ops_['score0'], ops_['score1']):
                    ol.patch_ints(op, 3, flat + [G])               # AFF_SCORE i3..8 = the token ranges, i9 = their 16-token tiles
                ol.patch_ints(ops_['select'], 2, [G])
                if not ahead:
                    ticks_n = ([bucket.n_work] if tick_work else []) + ([bucket.n_long] if tick_long else [])
                    if ticks_n:
                        ol.patch_ints(ops_['select'], 4, ticks_n)  # AFF_SELECT i4, i5 = the counted ranges (in the order they were given)
                if ops_['clear'] is not None:
                    ol.patch_ints(ops_['clear'], 0, [2 * bucket.n_long])
                plans_[ahead] = cached = (key, vals, ol, ops_)
        else:
            D = O.Dyn
            ol = O.OpList()
            ops_ = dict(clear=None)
            # (prio: a one-frame read-out is waited for by the caller's stream -- its waves take issue priority over the window encoder
            # and the stacked read-outs that share the compute units)
            common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=CAND_CAP, prio=True)
            ops_['score0'] = ol.aff_score(D('Ahi'), D('Alo'), D('scale'), D('Bhi'), D('Blo'), D('cq'), D('gmax'), None, None, None, mode=0, **common)
            # usage bookkeeping (kv_memory_store.py:151-162): life += 1 for every counted token -- rides on the selection launch,
            # as does the clearing of pass 1's candidate counters
            ticks = []
            if tick_work and not ahead:
                ticks.append((D('life', 4 * bucket.work_start), bucket.n_work))
            if tick_long and not ahead:
                ticks.append((D('life'), bucket.n_long))
            if ahead and self.use_long_term:
                ops_['select'] = ol.aff_select(D('gmax'), D('tau'), HW=HW, HWp=HWp, G=G, top_k=self.top_k, clear_count=D('count'), zero=(D('usage'), 2 * nslots), prio=True)
            elif _UNFUSED:
                ol.memset32(D('count'), HW * O.OpList.AFF_CSTRIDE, 0)
                ops_['select'] = ol.aff_select(D('gmax'), D('tau'), HW=HW, HWp=HWp, G=G, top_k=self.top_k)
                for life, n in ticks:
                    ol.usage_tick(life, n)
            else:
                ops_['select'] = ol.aff_select(D('gmax'), D('tau'), HW=HW, HWp=HWp, G=G, top_k=self.top_k, clear_count=D('count'), ticks=ticks, prio=True)
            ops_['score1'] = ol.aff_score(D('Ahi'), D('Alo'), D('scale'), D('Bhi'), D('Blo'), D('cq'), D('tau'), D('cval'), D('cidx'), D('count'),
                                          mode=1, gmax_precedes_tau=True, **common)
            # usage (kv_memory_store.py:151-162, the column sums of the affinity): accumulated in unsigned 64-bit FIXED POINT (2^-40) side counters
            # -- integer atomics commute, a float sum's last bits follow the order in which the read-out's blocks arrive, and a near-tie of
            # the consolidation's usage ranking follows those bits a few hundred frames later (tools/lockstep_soak.py) -- and added to the
            # bank's fp32 counters with ONE rounding per read-out: by the last launch of this plan, or when a look-ahead read-out is consumed
            ol.aff_readout(D('cval'), D('cidx'), D('count'), D('vptrs'), D('usage') if self.use_long_term else None, D('readout'),
                           D('ovf'), HW=HW, cap=CAND_CAP, top_k=self.top_k, K=K, CV=self.CV, prio=True, usage_fx=True)
            if clear_long:
                # long_term.count_usage=False: the reference keeps no usage for long-term tokens (memory_manager.py:145-147);
                # the read-out kernel accumulates usage for every slot, so the long-term part is cleared again
                ops_['clear'] = ol.memset32(D('usage'), 2 * bucket.n_long, 0)
            if self.use_long_term and not ahead:
                ol.usage_tick(None, 0, None, 0, use=D('use'), delta=D('usage'), n_use=nslots, delta_fx=True, clear_delta=True)
            plans_[ahead] = cached = (key, vals, ol, ops_)
        dyn = dict(count=count, Ahi=bucket.Ahi, Alo=bucket.Alo, scale=bucket.scale, Bhi=q['Bhi'], Blo=q['Blo'], cq=q['cq'],
                   gmax=gmax, tau=tau, cval=cval, cidx=cidx, vptrs=bucket.vptrs(), readout=readout, ovf=ovf)
        if self.use_long_term:
            if ahead:
                # two side buffers per bucket, alternating PER FRAME (prefetch_affinity flips the parity once per call, not once per
                # bucket: with an even number of buckets a per-bucket flip handed every bucket the same buffer on every frame): the next
                # look-ahead (side stream) may start before the caller's stream has applied this one
                udelta = self._buf(f'udelta{self._ahead_parity}#{bucket.id}', (nslots,), torch.int64, dev)
                dyn.update(life=bucket.life, usage=udelta)
                self._last_udelta = udelta
            else:                                               # (zero between two read-outs: the plan's last launch clears what it has added)
                dyn.update(life=bucket.life, usage=self._buf(f'udelta_direct#{bucket.id}', (nslots,), torch.int64, dev), use=bucket.use)
        cached[2].run(**dyn)
        return readout

    def _affinity_batch(self, bucket: Bucket, q, h: int, w: int, dev, frames: int):
        """One read-out per BANK VERSION instead of one per frame (no counterpart in the reference, which reads once per frame,
        memory_manager.py:112-208): the bank changes on memory frames only (inference_core.py:238), and what frame t + j reads depends on
        nothing but (bank, key_{t+j}, selection_{t+j}) (memory_utils.py:7-77).  The stacked query operands of `frames` consecutive frames
        go through the same four launches; per query the arithmetic is that of the one-frame plan (same bits).  Every frame gets its own
        read-out tensor and its own usage side buffer: `read` commits a frame's usage when, and only when, it consumes that frame."""
        F = frames
        HW = h * w
        HWp = q['Bhi'].shape[0]
        K = len(bucket.objects)
        ranges = [r for r in bucket.ranges() if r[1] > 0]
        G = sum(-(-n // 16) for _, n in ranges)
        Gld = -(-max(G, 1) // 64) * 64
        rows = F * HWp
        gbuf = self._buf_rows('gmax_tau#batch', rows * Gld + rows, (), F32, dev)     # (pass 0 writes every entry pass 1 / the selection read)
        gmax, tau = gbuf[:rows * Gld], gbuf[rows * Gld:]
        cval = self._buf_rows('cand_val#batch', rows, (CAND_CAP,), F32, dev)
        cidx = self._buf_rows('cand_idx#batch', rows, (CAND_CAP,), torch.int32, dev)
        count = self._buf_rows('count#batch', rows, (O.OpList.AFF_CSTRIDE,), torch.int32, dev)
        ovf = self._buf('overflow', (1,), torch.int32, dev)
        pool = getattr(self, '_pool', None)
        spec = dict(r=((F, K, h, w, self.CV), BF16, False))
        readout = (pool.get_ring(('readout#batch', self._clip_tag, bucket.id, F, K, h, w, str(dev)), spec, dev, ring=3)['r'] if pool is not None
                   else torch.empty(spec['r'][0], dtype=BF16, device=dev))
        nslots = int(bucket.use.shape[0]) if self.use_long_term else 0
        clear_long = self.use_long_term and bucket.n_long > 0 and not self.count_long_term_usage
        flat = [v for r in ranges for v in r] + [0] * (6 - 2 * len(ranges))
        key = (len(ranges), K, HW, HWp, self.top_k, self.use_long_term, clear_long, nslots, F, BATCH_FORMS)     # (see _affinity: launches / sizes)
        vals = (tuple(flat), G, bucket.n_long)
        plans_ = bucket.__dict__.setdefault('_aff_plans', {})
        cached = plans_.get(('batch', F))
        if cached is not None and cached[0] == key and (cached[1] == vals or PATCH_PLANS):
            if cached[1] != vals:
                ol, ops_ = cached[2], cached[3]
                for op in (ops_['score0'], ops_['score1']):
                    ol.patch_ints(op, 3, flat + [G])
                ol.patch_ints(ops_['select'], 2, [G])
                for op in ops_['clear']:
                    ol.patch_ints(op, 0, [2 * bucket.n_long])
                plans_[('batch', F)] = cached = (key, vals, ol, ops_)
        else:
            D = O.Dyn
            ol = O.OpList()
            ops_ = dict(clear=[])
            common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=CAND_CAP, frames=F)
            # kernel forms by the number of stacked frames (same bits whatever the form, tests/test_gpu_kernels.py; isolated stage
            # times at 12.2 k tokens, tools/aff_batch_ab.py, profiles/r05_affinity.md): from three frames on the score pass runs 64 queries
            # per wave on the LDS-DMA kernel (its longer prologue is amortised: 70.6 against 80.0 us at F = 5, MFMA utilisation 0.44
            # against 0.39) and the candidate pass stages its memory tiles by LDS-DMA (94 against 103 us)
            big = F >= 3 and BATCH_FORMS
            nq0, dma1 = (4, True) if big else (None, None)
            ops_['score0'] = ol.aff_score(D('Ahi'), D('Alo'), D('scale'), D('Bhi'), D('Blo'), D('cq'), D('gmax'), None, None, None, mode=0, nq=nq0, **common)
            ops_['select'] = ol.aff_select(D('gmax'), D('tau'), HW=HW, HWp=HWp, G=G, top_k=self.top_k, clear_count=D('count'), frames=F,
                                           zero=(D('usage'), 2 * F * nslots) if self.use_long_term else None)
            ops_['score1'] = ol.aff_score(D('Ahi'), D('Alo'), D('scale'), D('Bhi'), D('Blo'), D('cq'), D('tau'), D('cval'), D('cidx'), D('count'),
                                          mode=1, gmax_precedes_tau=True, nq=2 if big else None, dma=dma1, **common)
            ol.aff_readout(D('cval'), D('cidx'), D('count'), D('vptrs'), D('usage') if self.use_long_term else None, D('readout'),
                           D('ovf'), HW=HW, cap=CAND_CAP, top_k=self.top_k, K=K, CV=self.CV, frames=F, HWp=HWp, usage_stride=nslots, usage_fx=True)
            if clear_long:
                for f in range(F):                                      # (memory_manager.py:145-147, as in the one-frame plan)
                    ops_['clear'].append(ol.memset32(D('usage', 8 * f * nslots), 2 * bucket.n_long, 0))
            plans_[('batch', F)] = cached = (key, vals, ol, ops_)
        dyn = dict(count=count, Ahi=bucket.Ahi, Alo=bucket.Alo, scale=bucket.scale, Bhi=q['Bhi'], Blo=q['Blo'], cq=q['cq'],
                   gmax=gmax, tau=tau, cval=cval, cidx=cidx, vptrs=bucket.vptrs(), readout=readout, ovf=ovf)
        udelta = None
        if self.use_long_term:
            # two sets of side buffers per bucket, alternating per BATCH (a counter of the stacked read-outs alone -- ADVICE r05: on the
            # shared counter of the one-frame look-ahead, which flips once more per memory cycle, every batch landed on the same set):
            # the next batch (side stream) may start before the caller's stream has applied the last frame of this one
            udelta = self._buf_rows(f'udelta#batch{self._batch_parity}#{bucket.id}', F, (nslots,), torch.int64, dev)
            dyn.update(usage=udelta)
        cached[2].run(**dyn)
        return [(readout[f], udelta[f] if udelta is not None else None) for f in range(F)]

    def _commit_ahead(self, bucket: Bucket, udelta: torch.Tensor, network=None) -> None:
        """Bookkeeping of a consumed look-ahead read-out, one launch: the life counters of the counted token ranges advance by one and
        the usage the read-out parked in its side buffer is added to the bank's (kv_memory_store.py:151-162) -- exactly what `_affinity`
        does itself when it runs inside `read`.  Issued when, and only when, `read` consumes the read-out (a dropped look-ahead has
        counted nothing) -- but on the look-ahead's OWN stream: nothing on the caller's stream needs the counters before the next
        memory frame (`_join_side`), the read-outs that follow on that stream see them in order, and the frame's critical path is one
        launch shorter."""
        if not self.use_long_term:
            return
        tick_work = bucket.n_work > 0
        tick_long = bucket.n_long > 0 and self.count_long_term_usage
        # (host time: the launch stays the same while the bank's arrays do; the two counted lengths are patched into its descriptor)
        sig = (bucket.life.data_ptr(), bucket.use.data_ptr(), bucket.work_start, tick_work, tick_long, int(udelta.shape[0]))
        counts = (bucket.n_work if tick_work else 0, bucket.n_long if tick_long else 0)
        cached = self._commit_plans.get(bucket.id)
        if cached is None or cached[0] != sig:
            ol = O.OpList()
            ol.usage_tick(bucket.life[bucket.work_start:] if tick_work else None, counts[0], bucket.life if tick_long else None, counts[1],
                          use=bucket.use, delta=O.Dyn('delta'), n_use=sig[5], delta_fx=True)
            cached = self._commit_plans[bucket.id] = (sig, ol, counts)
        elif cached[2] != counts:
            if PATCH_PLANS:
                cached[1].patch_ints(0, 0, counts)               # USAGE_TICK i0, i1
                cached = self._commit_plans[bucket.id] = (sig, cached[1], counts)
            else:
                self._commit_plans.pop(bucket.id)
                return self._commit_ahead(bucket, udelta, network)
        ol = cached[1]
        side = None
        if COMMIT_ON_SIDE and network is not None and udelta.is_cuda:
            eng = network.engine()
            side = None if eng.one_lane else eng.__dict__.get('_streams', {}).get('side')
        if side is None:
            ol.run(delta=udelta)
            return
        ol.run(_stream=side, delta=udelta)                       # (the read-out that filled udelta ran on this stream: in order)
        self._side_pending = side

    def _join_side(self) -> None:
        """The caller's stream waits for bookkeeping launches that `_commit_ahead` left on the look-ahead stream: before anything on it
        reads or rewrites the usage / life counters (memorising with its long-term maintenance, a read-out that counts on the bank
        directly, purges)."""
        side = self.__dict__.pop('_side_pending', None)
        if side is not None:
            torch.cuda.current_stream(side.device).wait_stream(side)

    def prefetch_affinity(self, query_key: torch.Tensor, selection: torch.Tensor, network) -> None:
        """Look-ahead lane (no counterpart in the reference): the affinity read-out of the NEXT frame depends only on that frame's
        key and on the bank, so when the current frame does not write the bank (not a memory frame) it can run ahead on the encoder's
        side stream.  The result rides on the query operands and is used by `read` if the bank is still the one it was read from
        (`_version`).  Usage counters are updated here, exactly once (a later invalidation -- objects deleted between two steps --
        would count the frame's read twice; the reference's GUI path never deletes mid-propagation)."""
        if not self.engaged or self.CV is None:
            return None
        q = network.query_operands(query_key, selection)
        h, w = q['h'], q['w']
        dev = q['Bhi'].device
        self._pool = network.engine().pool
        self._ahead_parity ^= 1
        out = {}
        for bid, b in self.buckets.items():
            r = self._affinity(b, q, h, w, dev, ahead=True)
            out[bid] = (r, self._version, self._last_udelta if self.use_long_term else None)
        q['_readouts'] = out
        return out

    def prefetch_affinity_batch(self, qs: List[dict], network, event_factory=None):
        """Look-ahead read-out of SEVERAL frames against one bank version (`_affinity_batch`).  qs: the query-operand dicts (the `_qo` of
        the encoder records) of consecutive frames of one encoder batch, next frame first.  Each dict receives its `_readouts` exactly
        as `prefetch_affinity` attaches them, plus `_readouts_ev` -- the event `read` waits for before it touches them."""
        if not self.engaged or self.CV is None or not qs:
            return None
        q0 = qs[0]
        h, w = q0['h'], q0['w']
        dev = q0['Bhi'].device
        HWp = q0['Bhi'].shape[0]
        for a, b in zip(qs, qs[1:]):                                    # stacked operands: frame f + 1 right behind frame f
            assert b['Bhi'].data_ptr() == a['Bhi'].data_ptr() + HWp * 256 and b['Blo'].data_ptr() == a['Blo'].data_ptr() + HWp * 256 \
                and b['cq'].data_ptr() == a['cq'].data_ptr() + HWp * 4, 'frames of a batched read-out must come from one encoder batch'
        self._pool = network.engine().pool
        self._batch_parity ^= 1
        per_frame = [dict() for _ in qs]
        for bid, b in self.buckets.items():
            for f, (r, ud) in enumerate(self._affinity_batch(b, q0, h, w, dev, len(qs))):
                per_frame[f][bid] = (r, self._version, ud)
        ev = event_factory() if event_factory is not None else None
        for q, out in zip(qs, per_frame):
            q['_readouts'] = out
            q['_readouts_ev'] = ev
        return per_frame

    @staticmethod
    def prefetch_affinity_joint(mms: List['MemoryManager'], qs: List[dict], network, event_factory=None, prio: bool = False) -> bool:
        """Clips in lock step (inference/lockstep.py; no counterpart in the reference): the look-ahead read-outs of F frames of C clips --
        C banks with ONE schedule, hence one set of token ranges -- as one score / select / score / read-out sequence.  qs: the F x C
        query-operand dicts FRAME-MAJOR (entry e = frame * C + clip reads bank e % C), rows stacked as `prefetch_affinity_batch` wants
        them.  Per entry the arithmetic is that of the one-frame plan on the entry's own bank (tests/test_gpu_kernels.py:
        test_affinity_frames_of_several_banks_in_one_pass).  Every dict receives `_readouts` / `_readouts_ev` as from
        `prefetch_affinity_batch`; the first dict of every frame also `_readouts_joint` = the [C * K, h, w, CV] read-outs of that frame
        (the clips' slices of one tensor: pixel fusion takes them without a gather).  False (nothing done): the banks do not line up."""
        C, E = len(mms), len(qs)
        m0 = mms[0]
        if C < 2 or E == 0 or E % C or any(not mm.engaged or mm.CV is None or len(mm.buckets) != 1 for mm in mms):
            return False
        bs = [next(iter(mm.buckets.values())) for mm in mms]
        b0 = bs[0]
        cfgs = lambda mm: (mm.top_k, mm.use_long_term, mm.count_long_term_usage, mm.CV)
        sig = lambda b: (len(b.objects), tuple(b.ranges()), int(b.use.shape[0]) if m0.use_long_term else 0)
        q0 = qs[0]
        h, w, dev, HWp = q0['h'], q0['w'], q0['Bhi'].device, q0['Bhi'].shape[0]
        if any(cfgs(mm) != cfgs(m0) for mm in mms) or any(sig(b) != sig(b0) for b in bs) or HWp % 128:
            return False
        for a, b in zip(qs, qs[1:]):
            if not (b['Bhi'].data_ptr() == a['Bhi'].data_ptr() + HWp * 256 and b['Blo'].data_ptr() == a['Blo'].data_ptr() + HWp * 256
                    and b['cq'].data_ptr() == a['cq'].data_ptr() + HWp * 4):
                return False
        HW, K = h * w, len(b0.objects)
        ranges = [r for r in b0.ranges() if r[1] > 0]
        G = sum(-(-n // 16) for _, n in ranges)
        Gld = -(-max(G, 1) // 64) * 64
        rows = E * HWp
        pool = network.engine().pool
        gbuf = m0._buf_rows('gmax_tau#joint', rows * Gld + rows, (), F32, dev)
        gmax, tau = gbuf[:rows * Gld], gbuf[rows * Gld:]
        cval = m0._buf_rows('cand_val#joint', rows, (CAND_CAP,), F32, dev)
        cidx = m0._buf_rows('cand_idx#joint', rows, (CAND_CAP,), torch.int32, dev)
        count = m0._buf_rows('count#joint', rows, (O.OpList.AFF_CSTRIDE,), torch.int32, dev)
        ovf = m0._buf('overflow', (1,), torch.int32, dev)
        spec = dict(r=((E, K, h, w, m0.CV), BF16, False))
        readout = pool.get_ring(('readout#joint', C, E, K, h, w, str(dev)), spec, dev, ring=3)['r']
        nslots = sig(b0)[2]
        clear_long = m0.use_long_term and b0.n_long > 0 and not m0.count_long_term_usage
        flat = [v for r in ranges for v in r] + [0] * (6 - 2 * len(ranges))
        key = (len(ranges), K, HW, HWp, m0.top_k, m0.use_long_term, clear_long, nslots, E, C)
        vals = (tuple(flat), G, b0.n_long)
        plans_ = m0.__dict__.setdefault('_joint_plans', {})
        cached = plans_.get((E, prio))
        if cached is not None and cached[0] == key:
            if cached[1] != vals:
                ol, ops_ = cached[2], cached[3]
                for op in (ops_['score0'], ops_['score1']):
                    ol.patch_ints(op, 3, flat + [G])
                ol.patch_ints(ops_['select'], 2, [G])
                for op in ops_['clear']:
                    ol.patch_ints(op, 0, [2 * b0.n_long])
                plans_[(E, prio)] = cached = (key, vals, ol, ops_)
        else:
            D = O.Dyn
            ol = O.OpList()
            ops_ = dict(clear=[])
            common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=CAND_CAP, frames=E, nq=2, banks=(D('table'), C), prio=prio)
            ops_['score0'] = ol.aff_score(None, None, None, D('Bhi'), D('Blo'), D('cq'), D('gmax'), None, None, None, mode=0, dma=bool(JOINT_DMA & 1), **common)
            ops_['select'] = ol.aff_select(D('gmax'), D('tau'), HW=HW, HWp=HWp, G=G, top_k=m0.top_k, clear_count=D('count'), frames=E,
                                           zero=(D('usage'), 2 * E * nslots) if m0.use_long_term else None, prio=prio)
            ops_['score1'] = ol.aff_score(None, None, None, D('Bhi'), D('Blo'), D('cq'), D('tau'), D('cval'), D('cidx'), D('count'),
                                          mode=1, gmax_precedes_tau=True, dma=bool(JOINT_DMA & 2), **common)
            ol.aff_readout(D('cval'), D('cidx'), D('count'), D('vptrs'), D('usage') if m0.use_long_term else None, D('readout'),
                           D('ovf'), HW=HW, cap=CAND_CAP, top_k=m0.top_k, K=K, CV=m0.CV, frames=E, HWp=HWp, usage_stride=nslots, banks=C, prio=prio, usage_fx=True)
            if clear_long:
                for e in range(E):                                      # (memory_manager.py:145-147, as in the one-frame plan)
                    ops_['clear'].append(ol.memset32(D('usage', 8 * e * nslots), 2 * b0.n_long, 0))
            plans_[(E, prio)] = cached = (key, vals, ol, ops_)
        # the banks' operand bases and value pointers, bank-major (rebuilt when an array of a bank moved: it grew)
        ptrs = tuple((b.Ahi.data_ptr(), b.Alo.data_ptr(), b.scale.data_ptr(), b.vptrs().data_ptr()) for b in bs)
        tbl = m0.__dict__.get('_joint_table')
        if tbl is None or tbl[0] != ptrs:
            table = torch.tensor([list(p[:3]) for p in ptrs], dtype=torch.int64).to(dev)
            tbl = m0._joint_table = (ptrs, table, torch.cat([b.vptrs() for b in bs]), [b.vptrs() for b in bs])
        dyn = dict(count=count, table=tbl[1], Bhi=q0['Bhi'], Blo=q0['Blo'], cq=q0['cq'], gmax=gmax, tau=tau, cval=cval, cidx=cidx,
                   vptrs=tbl[2], readout=readout, ovf=ovf)
        udelta = None
        if m0.use_long_term:
            m0._joint_parity = m0.__dict__.get('_joint_parity', 0) ^ 1      # (two sets of usage side buffers, alternating per pass: see _affinity_batch)
            udelta = m0._buf_rows(f'udelta#joint{m0._joint_parity}', E, (nslots,), torch.int64, dev)
            dyn.update(usage=udelta)
        cached[2].run(**dyn)
        ev = event_factory() if event_factory is not None else None
        for e, q in enumerate(qs):
            mm, b = mms[e % C], bs[e % C]
            q['_readouts'] = {b.id: (readout[e], mm._version, udelta[e] if udelta is not None else None)}
            q['_readouts_ev'] = ev
            if e % C == 0:
                q['_readouts_joint'] = readout[e:e + C].view(C * K, h, w, m0.CV)
        return True

    def read_visual(self, q: dict, h: int, w: int, dev, network) -> Dict[int, torch.Tensor]:
        """The affinity half of `read` (memory_manager.py:126-167: similarity, top-k softmax, value read-out): bucket id -> read-out
        bf16 [K, h, w, CV] for the query operands q.  A read-out that the look-ahead lane computed against this very bank version is taken
        over -- and only now counted (`_commit_ahead`); otherwise it is computed here, on the caller's stream."""
        self._pool = network.engine().pool
        ahead = q.pop('_readouts', None) or {}
        ahead_ev = q.pop('_readouts_ev', None)
        if ahead_ev is not None and ahead:                              # a look-ahead read-out: ordered behind its own event
            torch.cuda.current_stream(dev).wait_event(ahead_ev)
        out = {}
        for bucket in self.buckets.values():
            K = len(bucket.objects)
            pre = ahead.get(bucket.id)
            if pre is not None and pre[1] == self._version and pre[0].shape[0] == K:
                out[bucket.id] = pre[0]                                 # computed ahead on the side stream (the caller has waited for it)
                if pre[2] is not None:
                    self._commit_ahead(bucket, pre[2], network)         # its bookkeeping, now that it is used
            else:
                self._join_side()
                out[bucket.id] = self._affinity(bucket, q, h, w, dev)
        return out

    def read(self, pix_feat: torch.Tensor, query_key: torch.Tensor, selection: torch.Tensor, last_mask: torch.Tensor,
             network) -> Dict[int, torch.Tensor]:
        q = network.query_operands(query_key, selection)
        h, w = pix_feat.shape[-2:]
        dev = pix_feat.device
        visual = self.read_visual(q, h, w, dev, network)
        all_readout = {}
        for bucket in self.buckets.values():
            K = len(bucket.objects)
            readout = visual[bucket.id]
            # chunk_size > 0 (memory_manager.py:169-186): pixel fusion and the object transformer run per group of chunk_size
            # objects -- this is NOT only a memory knob: the "others" mask of the fusion and the foreground / background
            # attention masks of the transformer are computed inside a group.  (encode_mask / segment chunks are equivalent
            # to the batched form and stay batched.)
            cs = self.chunk_size if (self.chunk_size is not None and self.chunk_size >= 1) else K
            chunks = []
            for c0 in range(0, K, cs):
                objects = bucket.objects[c0:c0 + cs]
                this_sensory = self.get_sensory(objects)
                this_last_mask = self._get_mask_by_ids(last_mask, objects)
                visual_readout = readout[c0:c0 + cs].permute(0, 3, 1, 2).unsqueeze(0)
                pixel_readout = network.pixel_fusion(pix_feat, visual_readout, this_sensory, this_last_mask)
                a, b = self._rows(objects, self._objv_ids)
                this_obj_mem = self._objv[a:b].unsqueeze(0).unsqueeze(2)                # [1,K,1,Q,C+1]
                readout_memory, aux_features = network.readout_query(pixel_readout, this_obj_mem, _last_aux=self.save_aux or _UNFUSED,
                                                                     _summary_token=(self._objv_token, a, b))
                for i, obj in enumerate(objects):
                    all_readout[obj] = readout_memory[:, i]
                chunks.append((objects, readout_memory))
                if self.save_aux:
                    self.aux = self._aux_output(this_sensory, pixel_readout, aux_features, Q=int(this_obj_mem.shape[3]))
            self._last_readout = chunks[0] if len(chunks) == 1 else (None, None)
        if _VALIDATE:
            self.check_overflow()
        return all_readout

    @staticmethod
    def _aux_output(sensory, pixel_readout, aux_features, Q: int = 16) -> dict:
        """cfg.save_aux (memory_manager.py:197-206): the intermediate tensors of the last object chunk, under the reference's keys.
        The reference itself cannot produce this dict in eval mode (it indexes aux_features['attn_mask'], which QueryTransformer
        only fills while training: KeyError on the first read -- tests/golden/edge_cases.json, save_aux_on_read), so this is the
        intended content rather than an observed one: logits are copies (the plan's buffers are rewritten by the next frame),
        attn_mask is object_transformer.py:179-205 applied to the last block's logits, head 0, as float [1,K,Q,h,w] (1 = blocked);
        the attention weights are not kept by the fused attention kernels (None, as with need_weights=False)."""
        logits = [t.float().clone() for t in aux_features['logits']] if aux_features else None
        attn_mask = None
        if logits:
            lg = logits[-1]                                               # [1,K,h,w]
            prob = torch.sigmoid(lg)
            bg = torch.prod(1 - prob, dim=1, keepdim=True).clamp(1e-7, 1 - 1e-7)
            pr = prob.clamp(1e-7, 1 - 1e-7)
            agg = torch.cat([torch.log(bg / (1 - bg)), torch.log(pr / (1 - pr))], 1)
            fg = agg[:, 1:] >= agg.max(dim=1, keepdim=True)[0]            # [1,K,h,w]
            m = torch.cat([(~fg).unsqueeze(2).expand(-1, -1, Q // 2, -1, -1), fg.unsqueeze(2).expand(-1, -1, Q // 2, -1, -1)], 2).clone()
            full = m.flatten(3).all(-1)                                   # a fully blocked row is un-blocked (:203)
            m[full] = False
            attn_mask = m.float()
        return {'sensory': sensory, 'pixel_readout': pixel_readout, 'q_logits': logits, 'q_weights': None, 'p_weights': None,
                'attn_mask': attn_mask}

    def check_overflow(self) -> int:
        """Number of queries whose candidate list overflowed CAND_CAP since the last check (their top-k was taken from a
        truncated list, e.g. > 1024 exactly tied scores from duplicated permanent frames); warns and resets the counter.  Reads
        one int from the device (a synchronisation): called at the end of a clip (``step(end=True)``), by ``clear_memory`` and
        on every read when $CUTIE_AMD_VALIDATE is set."""
        ovf = self._scratch.get('overflow')
        if ovf is None:
            return 0
        n = int(ovf.item())
        if n:
            log.warning(f'affinity read-out: {n} queries had more than {CAND_CAP} candidates above the top-k threshold; '
                        f'their read-out used a truncated candidate list')
            ovf.zero_()
        return n

    def readout_stacked(self, all_obj_ids):
        """The per-object dict of read() re-stacked in tmp-id order (ObjectManager.realize_dict) without a copy when
        a single bucket holds every object."""
        objs, t = getattr(self, '_last_readout', (None, None))
        if objs is not None and list(objs) == list(all_obj_ids):
            return t
        return None

    # ---- write (memory_manager.py:210-296) ------------------------------------------------------------------------
    def add_memory(self, key, shrinkage, msk_value, obj_value, objects: List[int], selection=None, *, as_permanent=False) -> None:
        # the default is the reference's (memory_manager.py:218) and just as unusable: its store asserts the same
        # (kv_memory_store.py:79); InferenceCore always passes 'no' / 'first' / 'all'
        assert as_permanent in ['no', 'first', 'all']
        self._join_side()
        self._version = next(_VERSIONS)
        bs = key.shape[0]
        assert bs == 1 and shrinkage.shape[0] == 1 and msk_value.shape[0] == 1
        self.engaged = True
        if self.H is None or self.config_stale:
            self.config_stale = False
            self.H, self.W = msk_value.shape[-2:]
            self.HW = self.H * self.W
            self.max_work_tokens = self.max_mem_frames * self.HW
            if self.use_long_term:
                self.min_work_tokens = self.min_mem_frames * self.HW
            for b in self.buckets.values():                              # update_config changed the limits: resize the slabs
                self._fit_bucket(b)
        HW = self.HW
        dev = key.device
        self.CK = key.shape[1]
        self.CV = msk_value.shape[2]
        # physical views: key [HW,CK] f32, shrinkage [HW] f32, selection [HW,CK] f32, values [K,HW,CV] bf16
        kphys = key[0].permute(1, 2, 0).reshape(HW, self.CK)
        sphys = shrinkage.reshape(HW)
        ephys = selection[0].permute(1, 2, 0).reshape(HW, self.CK) if selection is not None else None
        kphys, sphys = kphys.to(F32).contiguous(), sphys.to(F32).contiguous()
        if ephys is not None:
            ephys = ephys.to(F32).contiguous()
        vphys = msk_value[0].permute(0, 2, 3, 1)
        if vphys.dtype != BF16 or not vphys.is_contiguous():
            vphys = vphys.to(BF16).contiguous()
        vphys = vphys.reshape(len(objects), HW, self.CV)

        ol = O.OpList()
        # ---- object memory: streaming sum (:252-271)
        if obj_value is not None:
            self.add_object_values(obj_value, objects, ol)

        # ---- bucket assignment (kv_memory_store.py:96-117)
        enabled = []
        new_objs = [o for o in objects if not any(o in b.objects for b in self.buckets.values())]
        if new_objs:
            work_cap = self.max_work_tokens
            b = Bucket(self._next_bucket, new_objs, HW, self.CK, self.CV, dev, use_long_term=self.use_long_term,
                       work_cap=work_cap, long_cap=self.max_long_tokens if self.use_long_term else 0)
            self.buckets[b.id] = b
            self._next_bucket += 1
        for b in self.buckets.values():
            if any(o in objects for o in b.objects):
                enabled.append(b)

        for b in enabled:
            to_perm = (as_permanent == 'all') or (as_permanent == 'first' and b.n_perm == 0)
            if to_perm:
                if b.n_perm + HW > b.P:
                    b.grow_perm(b.n_perm + HW)
                slot = b.perm_start + b.n_perm
                b.n_perm += HW
            else:
                if b.Wc < HW:
                    continue                                            # max_mem_frames == 1: no working memory at all
                if self.use_long_term:
                    if b.n_work + HW > b.Wc:
                        self._fit_bucket(b, extra_work=HW)
                    slot = b.work_start + b.n_work                      # linear; consolidation compacts
                    b.n_work += HW
                else:
                    frames = b.Wc // HW                                 # FIFO ring (:296, kv_memory_store.py:206-207)
                    slot = b.work_start + (b.ring % frames) * HW
                    b.ring += 1
                    b.n_work = min(b.n_work + HW, frames * HW)
            region_end = (b.perm_start + b.P) if to_perm else (b.work_start + b.Wc)
            assert slot + HW <= region_end, ('memory bank overrun', slot, HW, region_end)
            ol.key_prep(kphys, sphys, b.Ahi[slot:], b.Alo[slot:], b.scale[slot:], n=HW, query=False)
            copies, fills = [], []                                      # every tensor of the insertion in one launch (BANK_WRITE)
            if self.use_long_term:
                copies.append((kphys, b.rawkey[slot:], 4 * self.CK * HW))
                copies.append((sphys, b.rawshr[slot:], 4 * HW))
                if ephys is not None:
                    copies.append((ephys, b.rawsel[slot:], 4 * self.CK * HW))
                fills.append((b.use[slot:], HW, 0))
                fills.append((b.life[slot:], HW, LIFE_EPS_BITS))
            for o in b.objects:
                if o in objects:
                    copies.append((vphys[objects.index(o)], b.values[o][slot:], 2 * HW * self.CV))
            if BANK_WRITE:
                ol.bank_write(copies, fills)
            else:                                                       # (A/B switch: one launch per tensor, as before)
                for src, dst, nbytes in copies:
                    ol.copy2d(src, dst, rows=1, rowbytes=nbytes, src_stride=nbytes, dst_stride=nbytes)
                for dst, words, pattern in fills:
                    ol.memset32(dst, words, pattern)
        if len(ol):
            ol.run()

        # ---- long-term maintenance (:281-293)
        if self.use_long_term:
            for b in list(self.buckets.values()):
                if b.n_work >= self.max_work_tokens:
                    if b.n_long >= self.max_long_tokens - self.num_prototypes:
                        self._remove_obsolete(b, self.max_long_tokens - self.num_prototypes - self.buffer_tokens)
                    self._compress(b)

    def add_object_values(self, obj_value, objects: List[int], ol=None) -> None:
        """The object-memory half of add_memory (memory_manager.py:252-271: the summaries of a memorised frame are added to the
        running sums).  On its own when InferenceCore memorises in two parts (add_memory(obj_value=None) first)."""
        own = ol is None
        if own:
            ol = O.OpList()
        ov = obj_value[0].to(F32).contiguous()                         # [K,Q,C+1]
        known = [o for o in objects if o in self._objv_ids]
        new = [o for o in objects if o not in self._objv_ids]
        if known:
            ka, kb = self._rows(known, self._objv_ids)
            ia = objects.index(known[0])
            ol.axpy(ov[ia:ia + len(known)], self._objv[ka:kb], n=len(known) * ov.shape[1] * ov.shape[2], a=1.0)
            self._objv_token = next(MemoryManager._TOKENS)               # (updated in place)
        if new:
            ia = objects.index(new[0])
            rows = ov[ia:ia + len(new)].clone()
            for j, o in enumerate(new):
                if o in self._orphan_objv:                            # object re-added after deletion
                    rows[j] += self._orphan_objv.pop(o)
            self._objv = rows if (self._objv is None or not self._objv_ids) else torch.cat([self._objv, rows], 0)
            self._objv_ids = self._objv_ids + new
        if own and len(ol):
            ol.run()

    def _fit_bucket(self, b: Bucket, extra_work: int = 0):
        """Make the slab capacities of a bucket match the current memory settings (they are sized when the bucket is created;
        ``update_config`` -- the GUI's memory sliders -- may change max_mem_frames / max_num_tokens afterwards).  FIFO mode: the
        working region is exactly the ring of max_mem_frames-1 frames (a smaller setting drops the oldest frames, like the
        reference's trim at the next insertion).  Long-term mode: room for the tokens present plus one more frame, and for one
        more batch of prototypes behind the long-term tokens."""
        if self.use_long_term:
            Wc = max(self.max_work_tokens, b.n_work + extra_work)
            L = max(self.max_long_tokens, b.n_long + self.num_prototypes)
            if Wc != b.Wc or L > b.L:
                b.reserve(L=max(L, b.L), Wc=Wc)
        elif b.Wc != self.max_work_tokens:
            b.reserve(Wc=self.max_work_tokens)

    # ---- long-term consolidation (memory_manager.py:309-358) ----------------------------------------------------------
    def _compress(self, b: Bucket):
        HW, dev = self.HW, b.device
        P = self.num_prototypes
        n = b.n_work - self.min_work_tokens            # candidates: the oldest working tokens
        if n <= 0:
            return
        if b.n_long + P > b.L:
            self._fit_bucket(b)
        assert b.n_long + P <= b.L, ('long-term region overrun', b.n_long, P, b.L)
        ws = b.work_start
        ol = O.OpList()
        K = len(b.objects)
        order = self._buf('proto_order', (P,), torch.int32, dev)
        psel = self._buf('proto_sel', (P, b.CK), F32, dev)
        colmax = self._buf('consol_colmax', (P,), torch.int32, dev)
        ldS = O.OpList.consol_lds(n)
        S = self._buf_rows('consol_S', P * ldS, (), F32, dev)
        dst = b.n_long                                                                    # prototypes appended to the LT region
        # topk(usage, P) (:339); the launch that scatters the order also gathers the prototypes' keys (into the LT region) and
        # selections through it (:341-345) and clears the column maxima of the similarity pass
        ol.rank_select(b.use[ws:], b.life[ws:], order, n=n, k=P, scratch=self._buf_rows('rank_scratch', O.OpList.RS_SPLIT * n, (), torch.int32, dev),
                       gathers=[(b.rawkey[ws:], b.rawkey[dst:], 4 * b.CK), (b.rawsel[ws:], psel, 4 * b.CK)], zero=(colmax, P))
        # potentiation (:347-356): similarities candidates x prototypes, softmax over the candidates, applied to the values of every
        # object and to the shrinkage -- three launches (csrc/bank.hip)
        ol.consol_aff(b.rawkey[ws:], b.rawshr[ws:], b.rawkey[dst:], psel, S, colmax, n=n, P=P)
        ol.consol_read(S, colmax, b.vptrs(), b.rawshr[ws:], self._buf_rows('consol_part', O.OpList.consol_scratch_floats(n, P, b.CV, K), (), F32, dev),
                       b.rawshr[dst:], n=n, P=P, C=b.CV, K=K, src=ws, dst=dst)
        ol.key_prep(b.rawkey[dst:], b.rawshr[dst:], b.Ahi[dst:], b.Alo[dst:], b.scale[dst:], n=P, query=False)
        fills = [(b.use[dst:], P, 0), (b.life[dst:], P, LIFE_EPS_BITS)]
        # drop the consolidated tokens: keep the newest min_work_tokens, moved to the region start.  The copies are parallel: source and
        # destination must not overlap.  Fewer tokens kept than dropped (the default: 4 of 9 frames stay): every array in one go, six
        # arrays per launch, the counters of the new prototypes set by the first of them.  Otherwise -- min_mem_frames > max_mem_frames / 2
        # -- the move is issued as stream-ordered chunks of at most n rows, ascending, so that every chunk's destination ends where its
        # source begins.
        keep = self.min_work_tokens
        if keep <= n and BANK_WRITE:
            ol.bank_write([(t[ws + n:], t[ws:], rowbytes * keep) for t, rowbytes in b.arrays()] if keep > 0 else [], fills)
        else:
            ol.bank_write([], fills)
            for t, rowbytes in b.arrays():
                for c0 in range(0, keep, n):
                    nb = rowbytes * min(n, keep - c0)
                    ol.copy2d(t[ws + n + c0:], t[ws + c0:], rows=1, rowbytes=nb, src_stride=nb, dst_stride=nb)
        ol.run()
        b.n_long += P
        b.n_work = keep

    def _remove_obsolete(self, b: Bucket, max_size: int):
        """kv_memory_store.py:209-242: keep the `max_size` most-used long-term tokens (in top-k order)."""
        dev = b.device
        n = b.n_long
        order = self._buf('lt_order', (max_size,), torch.int32, dev)
        ol = O.OpList()
        ol.rank_select(b.use, b.life, order, n=n, k=max_size)
        tmps = []
        for t, rowbytes in b.arrays():
            tmp = torch.empty((max_size,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            ol.gather_rows(t, order, tmp, k=max_size, rowbytes=rowbytes, src_stride=rowbytes, dst_stride=rowbytes)
            tmps.append((t, tmp, rowbytes))
        for t, tmp, rowbytes in tmps:
            nb = rowbytes * max_size
            ol.copy2d(tmp, t, rows=1, rowbytes=nb, src_stride=nb, dst_stride=nb)
        ol.run()
        b.n_long = max_size

    # ---- object deletion / clearing ----------------------------------------------------------------------------------------
    def purge_except(self, obj_keep_idx: List[int]) -> None:
        self._join_side()
        self._version = next(_VERSIONS)
        keep = set(obj_keep_idx)
        for bid in list(self.buckets.keys()):
            b = self.buckets[bid]
            b.remove_objects(keep)
            if not b.objects:
                del self.buckets[bid]
        if self._ids:
            rows = [i for i, o in enumerate(self._ids) if o in keep]
            if len(rows) != len(self._ids):
                idx = torch.tensor(rows, dtype=torch.long, device=self._sens_f32.device)
                self._sens_f32 = self._sens_f32.index_select(0, idx)
                self._sens_bf16 = self._sens_bf16.index_select(0, idx)
                self._ids = [self._ids[i] for i in rows]
        if self._objv_ids:
            rows = [i for i, o in enumerate(self._objv_ids) if o in keep]
            if len(rows) != len(self._objv_ids):
                for i, o in enumerate(self._objv_ids):
                    if o not in keep:
                        self._orphan_objv[o] = self._objv[i].clone()
                idx = torch.tensor(rows, dtype=torch.long, device=self._objv.device)
                self._objv = self._objv.index_select(0, idx)
                self._objv_ids = [self._objv_ids[i] for i in rows]
        self._last_readout = (None, None)
        if not self.buckets:
            self.engaged = False

    def clear_non_permanent_memory(self):
        self._join_side()
        self._version = next(_VERSIONS)
        for b in self.buckets.values():
            b.n_work = 0
            b.ring = 0
            b.n_long = 0

    # reference attribute names used by callers / tests
    @property
    def sensory(self):
        return {o: self._sens_f32[i].permute(2, 0, 1).unsqueeze(0) for i, o in enumerate(self._ids)}

    @property
    def obj_v(self):
        return {o: self._objv[i].unsqueeze(0) for i, o in enumerate(self._objv_ids)}
