"""Launch-plan builders for the CUTIE stages (one plan = one ``cutie_exec`` call).

Each builder lays out the kernels of one facade method of the reference ``CUTIE`` (cutie/model/cutie.py)
over static scratch buffers; per-call inputs/outputs are named dynamic pointer slots (``ops.Dyn``).
Activations are NHWC bf16 ([B,H,W,C], B = objects), fp32 where the reference forces fp32
(logits, GRU state, summaries, keys).
"""
import math
import os

import numpy as np
import torch

from .. import _lib
from .. import ops as O
from ..ops import Dyn
from .param_spec import resnet_layers

BF16, F32 = torch.bfloat16, torch.float32


class Act:
    """NHWC activation handle: tensor (or Dyn slot), batch, height, width, channels, channel stride."""
    __slots__ = ('t', 'B', 'H', 'W', 'C', 'ld')

    def __init__(self, t, B, H, W, C, ld=None):
        self.t, self.B, self.H, self.W, self.C, self.ld = t, B, H, W, C, (C if ld is None else ld)


def positional_encoding(h, w, dim_total, scale, temperature):
    """[h,w,dim_total] sinusoidal PE (x half, then y half): positional_encoding.py:20-97 of the reference
    (normalize=True, eps=1e-6).  Constant per resolution -> computed once on the host at plan build."""
    dim = int(math.ceil(dim_total / 4) * 2)
    inv_freq = 1.0 / (temperature ** (torch.arange(0, dim, 2).float() / dim))
    py = torch.arange(h, dtype=torch.float32)
    px = torch.arange(w, dtype=torch.float32)
    py = py / (py[-1] + 1e-6) * scale
    px = px / (px[-1] + 1e-6) * scale
    sy, sx = py[:, None] * inv_freq[None], px[:, None] * inv_freq[None]
    ey = torch.stack((sy.sin(), sy.cos()), -1).flatten(-2)
    ex = torch.stack((sx.sin(), sx.cos()), -1).flatten(-2)
    emb = torch.zeros(h, w, dim * 2)
    emb[:, :, :dim] = ex[None]
    emb[:, :, dim:] = ey[:, None]
    return emb


TILE_CACHE_ENV = 'CUTIE_AMD_TILE_CACHE'
# timing effort of the conv autotuner: min over TUNE_REPS runs of TUNE_ITERS launches ($CUTIE_AMD_TUNE="reps x iters")
TUNE_REPS, TUNE_ITERS = (int(v) for v in os.environ.get('CUTIE_AMD_TUNE', '3x8').lower().split('x'))


# Timing-based tile selection for conv geometries that are NOT in the tuned table is opt-in ($CUTIE_AMD_AUTOTUNE=1, or
# Engine.autotune = True): timing is noisy, and a different (tile, split-K) changes the fp32 summation order, so by default such
# geometries take the deterministic static choice of ops.pick_tile and results are bit-reproducible across processes.
# diagnostic: $CUTIE_AMD_UNFUSED=1 restores the unfused launch sequences of round 1 for in-box A/B timing (tools/r2_call*.sh)
UNFUSED = os.environ.get('CUTIE_AMD_UNFUSED', '0') not in ('', '0')
QCHAIN = os.environ.get('CUTIE_AMD_QCHAIN', '1') not in ('', '0')     # query side of a transformer block in 4 launches (0: the round-2 seven)
# Plans replayed as HIP graphs once their pointer signature repeats (_lib.HipExecutor.run_cached): opt-in.  Measured on the MI355X
# (profiles/r03_host.md): 94 % of the plan runs replay, the frame rate does not move (995.8 against 1003.3 fps) -- the look-ahead frame
# is bound by the two streams sharing the compute units, not by the host's ~650 us of launches.
GRAPHS = os.environ.get('CUTIE_AMD_GRAPHS', '0') not in ('', '0')
GRAPH_MIN_OPS = 8
QFFN_SLICE = int(os.environ.get('CUTIE_AMD_QFFN_SLICE', '64'))       # hidden columns per QFFN block (64 | 128)
# the decoder's last launch (up-sampling + softmax of the logits) on an auxiliary stream next to the sensory update (4 launches that
# depend on the logits as well, not on each other's branch): A/B switch
# positional terms of the transformer's pixel projections composed INTO each block's projection conv ([pixel_b | x] -> k | v | q2) instead
# of one conv x -> [pixel | R_0 | R_1 | R_2] in front of the blocks (A/B switch)
PROJ_X = os.environ.get('CUTIE_AMD_PROJ_X', '1') not in ('', '0')
SEG_FORK = os.environ.get('CUTIE_AMD_SEG_FORK', '1') not in ('', '0')
# the object summarizer's five per-pixel linears as two composed launches (A/B switch; Engine '.in_fw0' / '.fw2')
SUM_FUSED = os.environ.get('CUTIE_AMD_SUM_FUSED', '1') not in ('', '0')
# MASK_DOWN of the next frame's pixel fusion inside the up-sampling launch of the current one (A/B switch)
SEG_MD = os.environ.get('CUTIE_AMD_SEG_MD', '1') not in ('', '0')
# read_from_query's output projection + residual inside the ATTN_P2Q launch (csrc/qchain.hip: p2q_out_kernel).  Bit-identical and OFF: the
# fused launch needs every head of a pixel in one workgroup, i.e. all of Wkv + Wo (384 KB) through ONE compute unit -- 21.6 us against
# 7.0 + 4.8 us + a boundary for the two launches (frame: -1.8 %; profiles/r04_frame_chain.md section 6)
P2Q_OUT = os.environ.get('CUTIE_AMD_P2Q_OUT', '0') not in ('', '0')       # (honoured only by the diagnostic library, see build_readout_query)
QINIT_SKIP = os.environ.get('CUTIE_AMD_QINIT_SKIP', '1') not in ('', '0')   # query initialisation only when the object summaries changed (A/B switch)
ONE_LANE = os.environ.get('CUTIE_AMD_ONE_LANE', '0') not in ('', '0')   # every look-ahead lane of a clip on the caller's own stream (several clips in flight: see cutie_amd/parallel.py)
ECA_HEAD = os.environ.get('CUTIE_AMD_ECA_HEAD', '1') not in ('', '0')   # mask_pred of a transformer block inside the ECA launch (A/B switch)
QNEXT = os.environ.get('CUTIE_AMD_QNEXT', '1') not in ('', '0')       # ATTN_P2Q also projects the next block's ATTN_Q2P queries
AUTOTUNE = os.environ.get('CUTIE_AMD_AUTOTUNE', '0') not in ('', '0')
TOUCH_REWIRE = os.environ.get('CUTIE_AMD_WPF_REWIRE', '1') not in ('', '0')       # A/B switch: next-weights ranges recomputed after the tile table
PACKAGED_TILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tiles_gfx950.json')


_p2q_warned = []


def _p2q_out_available():
    ex = _lib.get_executor()
    ok = ex.is_mock or _lib.has_diag_kernels()
    if not ok and not _p2q_warned:
        _p2q_warned.append(1)
        import warnings
        warnings.warn('CUTIE_AMD_P2Q_OUT=1 needs the diagnostic kernel library (make -C cutie_amd/csrc DIAG=1); ignored')
    return ok


def load_tile_cache():
    """Autotuned conv tiles persisted across processes.  Seeds, in this order: the packaged table of thoroughly timed
    choices for the common 480p shapes (cutie_amd/tiles_gfx950.json, written by tools/tune_tiles.py: removes the run-to-run
    variance of quick tuning and the tuning time at start-up), then $CUTIE_AMD_TILE_CACHE if it names a JSON file (that file
    is rewritten whenever a plan is tuned).  Keys are conv geometries, so one table serves every resolution / object count;
    geometries not in the table are tuned at their first use.  CUTIE_AMD_TILE_CACHE=none disables both."""
    import json
    path = os.environ.get(TILE_CACHE_ENV)
    if path == 'none':
        return {}
    cache = {}
    for f in (PACKAGED_TILES, path):
        if f and os.path.exists(f):
            try:
                cache.update({tuple(k): tuple(v) for k, v in json.load(open(f))['tiles']})
            except Exception:
                pass
    return cache


def save_tile_cache(cache):
    import json
    path = os.environ.get(TILE_CACHE_ENV)
    if not path or path == 'none':
        return
    tmp = path + '.tmp%d' % os.getpid()
    with open(tmp, 'w') as f:
        json.dump({'tiles': [[list(k), list(v)] for k, v in cache.items()]}, f)
    os.replace(tmp, path)


# Activation arenas.  A plan's scratch buffers used to be one allocation each, alive forever: ~450 MB of distinct addresses per 480p /
# 3-object frame on top of 70 MB of weights -- twice the 256 MB Infinity Cache, so every layer found its operands (and the previous
# layer's output) in HBM.  After a plan is built, its buffers are packed into an arena by live range (first / last launch that touches
# them, read off the descriptors), and the plans that run one after the other on a stream share ONE arena: the frame then cycles through
# ~150 MB of addresses and the memory-side cache keeps them (tools/cold_probe.py: a conv whose operands come from HBM costs 2-7 us
# more than the same launch served from the Infinity Cache).  $CUTIE_AMD_ARENA=0 restores one allocation per buffer.
ARENA = os.environ.get('CUTIE_AMD_ARENA', '1') not in ('', '0')
ARENA_POISON = os.environ.get('CUTIE_AMD_ARENA_POISON', '0') not in ('', '0')      # tests: garbage in the arena before every run
ARENA_ALIGN = 256


class Arena:
    """One growable device allocation shared by the plans of a stream class; plans re-bind their pointers when it moves."""

    def __init__(self, device):
        self.device = device
        self.tensor = None
        self.size = 0
        self.plans = []

    def require(self, nbytes):
        if nbytes <= self.size:
            return
        self.size = -(-int(nbytes * 1.1) // (1 << 20)) * (1 << 20)
        if self.tensor is not None:
            # launches in flight may still use the arena that is being outgrown: it is kept until the NEXT growth (by then every
            # launch that was issued against it has long been followed by others in stream order), not forever
            self._old = [self.tensor]
        self.tensor = torch.zeros(self.size, dtype=torch.uint8, device=self.device)
        for p in self.plans:
            p.rebind_arena()

    def base(self):
        return self.tensor.data_ptr()


_USE_COUNT = getattr(torch._C, '_storage_Use_Count', None)     # private torch API: without it the pool hands out fresh tensors only


def _storage_users(t):
    """How many OTHER tensors / views share t's storage (0: nobody else holds it)."""
    return _USE_COUNT(t.untyped_storage()._cdata) - 2


class SlotPool:
    """Per-frame outputs of the plans at STABLE addresses, so that a plan's pointer signature repeats and it can be replayed as a HIP
    graph (_lib.HipExecutor.run_cached).  Every group serves frame f from its slot f % SLOTS (InferenceCore.step ticks the frame
    number; the look-ahead lane asks for the NEXT frame's slot through `offset`), so the signatures of plans that consume each other's
    outputs repeat with period SLOTS.  A slot whose tensors are still referenced elsewhere (a caller that keeps results, the feature
    store with delete_buffer=False, a deferred memorising that still reads them, the same frame encoded twice) is NOT recycled: that
    call is served from fresh torch allocations, and the plan simply runs launch by launch.
    Cross-stream order: a slot is reused SLOTS frames later; the look-ahead encoder starts behind stream.wait_stream(main), i.e.
    behind every read of the frame that used the slot before."""
    SLOTS = int(os.environ.get('CUTIE_AMD_POOL_SLOTS', '6'))

    IDLE_FRAMES = int(os.environ.get('CUTIE_AMD_POOL_IDLE', '512'))      # a group nobody asked for in this many frames is dropped

    def __init__(self):
        self.groups = {}                  # key -> {slot index: (names, tensors, storages, storage handles)}
        self.last_used = {}               # key -> frame number of the last request
        self.frame = 0
        self.offset = 0

    def tick(self):
        self.frame += 1
        if self.frame % 64 == 0 and self.IDLE_FRAMES > 0:
            # groups are keyed by (stage, objects, resolution, ...): an interactive session that adds / removes objects, or a multi-scale
            # evaluation, would otherwise pin SLOTS copies of every shape it has ever seen.  Tensors still referenced
            # elsewhere simply stay alive with their holders.
            for k in [k for k, f in self.last_used.items() if self.frame - f > self.IDLE_FRAMES]:
                self.groups.pop(k, None)
                del self.last_used[k]

    @staticmethod
    def _make_slot(specs, dev):
        """One set of tensors + what `_free` needs: the storages are kept referenced here (that is the 2 of `_free`: this wrapper and
        the tensor) so that asking for their use count does not build a Python storage object per tensor and request."""
        names = tuple(specs)
        tensors = [torch.zeros(sh, dtype=dt, device=dev) for (sh, dt, z) in specs.values()]
        stores = [t.untyped_storage() for t in tensors]
        return names, tensors, stores, [st._cdata for st in stores]

    @staticmethod
    def _free(slot):
        """Nobody else holds a tensor of the slot or a view of one (aliases are what `_hand_out` returns)."""
        for c in slot[3]:
            if _USE_COUNT(c) != 2:
                return False
        return True

    @staticmethod
    def _hand_out(slot):
        # aliases (detach: same storage, shape and strides; a third of the cost of view(shape)): whoever keeps one -- or a view of
        # it -- keeps the slot busy
        return {nm: t.detach() for nm, t in zip(slot[0], slot[1])}

    def get(self, key, specs, dev):
        """specs: {name: (shape, dtype, zero)} -> {name: tensor}"""
        if self.SLOTS <= 0 or _USE_COUNT is None:
            return {n: (torch.zeros if z else torch.empty)(sh, dtype=dt, device=dev) for n, (sh, dt, z) in specs.items()}
        slots = self.groups.get(key)
        if slots is None:
            slots = self.groups[key] = {}
        self.last_used[key] = self.frame
        i = (self.frame + self.offset) % self.SLOTS
        slot = slots.get(i)
        if slot is None:
            slot = slots[i] = self._make_slot(specs, dev)
        if self._free(slot):
            return self._hand_out(slot)
        return {n: (torch.zeros if z else torch.empty)(sh, dtype=dt, device=dev) for n, (sh, dt, z) in specs.items()}

    def get_ring(self, key, specs, dev, ring=3):
        """Like `get`, for outputs that are not produced once per frame (the look-ahead window of the image encoder: one set per
        batch of frames): the group cycles through `ring` sets of its own; a set somebody still holds is not recycled."""
        if self.SLOTS <= 0 or _USE_COUNT is None:
            return {n: (torch.zeros if z else torch.empty)(sh, dtype=dt, device=dev) for n, (sh, dt, z) in specs.items()}
        slots = self.groups.setdefault(key, {'next': 0})
        self.last_used[key] = self.frame
        i = slots['next'] % ring
        slots['next'] = i + 1
        slot = slots.get(i)
        if slot is None:
            slot = slots[i] = self._make_slot(specs, dev)
        if self._free(slot):
            return self._hand_out(slot)
        return {n: (torch.zeros if z else torch.empty)(sh, dtype=dt, device=dev) for n, (sh, dt, z) in specs.items()}


def weights_go_cold(px16, K=None):
    """Does it pay that a conv touches the weights of the next one on its way out (ops.OpList.finalize; conv_pc.hip)?  A frame of px16
    stride-16 pixels and K objects moves ~0.18 MB x px16 x (1 + K) through HBM (1.2 GB at 480p / 3 objects against 256 MB of Infinity
    Cache).  Small frames: the weights stay warm between two uses and the touch only lengthens every launch.  Large frames: they are
    cold, but a launch runs for 50-300 us and its first K steps are noise against the tail the touch adds.  In between the launches are
    short AND cold.  Measured in one box each (tools/r3_call44.sh, 47, 48, 49; frames/s without the look-ahead hint, touch on / off):
    240p x 1 object 1270 / 1291, 240p x 3 1185 / 1197, 360p x 2 1025 / 1034, 480p x 1 945 / 943, 480p x 3 771 / 754,
    1080p x 5 191.5 / 193.1.  Plans that do not know K (the image encoder) decide on the pixels alone."""
    return 1400 <= px16 < 4500 if K is None else 4800 <= px16 * (1 + K) < 18000


class Plan:
    def __init__(self, eng, touch=True, prio=True):
        self.eng = eng
        self.prio = prio                 # its convs belong to the frame's critical path (ops.F_PRIO); the look-ahead encoder plans: False
        self.dev = eng.device
        self.ol = O.OpList(scratch_owner=eng, touch_next_weights=touch, prio=prio)
        self.bufs = {}
        self.meta = {}
        self.tuned = False
        self.persistent = set()          # buffers whose contents must survive between runs (zero padding written once, outputs read by the caller)
        self.arena = None
        self._arena_slots = []           # (op index, pointer slot, byte offset inside the arena)
        self._arena_views = {}           # name -> (offset, shape, dtype)
        self._graphs = {}                # pointer signature -> HIP graph of this plan (_lib.HipExecutor.run_cached)
        self.korder_ref = 0              # B > 1: a batched twin of a B = 1 plan -- tiles from the K-order class the B = 1 plan uses (autotune_convs)

    def buf(self, name, shape, dtype=BF16, persistent=False):
        assert name not in self.bufs, name
        t = torch.zeros(tuple(int(s) for s in shape), dtype=dtype, device=self.dev)
        self.bufs[name] = t
        if persistent:
            self.persistent.add(name)
        return t

    def pack_into(self, arena):
        """Pack the non-persistent buffers into `arena` by live range and re-point the descriptors (called once, when the plan is built)."""
        arr = self.ol.finalize()
        ptrs = arr['p'].astype(np.int64)
        items = []
        for name, t in self.bufs.items():
            if name in self.persistent or t.numel() == 0:
                continue
            base, nbytes = t.data_ptr(), t.numel() * t.element_size()
            hit = (ptrs >= base) & (ptrs < base + nbytes)
            if not hit.any():
                continue
            rows = np.nonzero(hit.any(axis=1))[0]
            items.append(dict(name=name, base=base, nbytes=nbytes, first=int(rows[0]), last=int(rows[-1]), hit=hit, t=t))
        # first-fit by start of life; a region is reusable by buffers whose first launch comes AFTER the last launch of its previous owner
        items.sort(key=lambda b: (b['first'], -b['nbytes']))
        placed = []                                       # (offset, end, last)
        total = 0
        for b in items:
            size = -(-b['nbytes'] // ARENA_ALIGN) * ARENA_ALIGN
            live = sorted((o, e) for (o, e, last) in placed if last >= b['first'])
            off = 0
            for o, e in live:
                if off + size <= o:
                    break
                off = max(off, e)
            b['off'] = off
            placed.append((off, off + size, b['last']))
            total = max(total, off + size)
        self.arena = arena
        self._arena_slots = []
        for b in items:
            idx, slot = np.nonzero(b['hit'])
            for i, sl in zip(idx, slot):
                self._arena_slots.append((int(i), int(sl), b['off'] + int(ptrs[i, sl] - b['base'])))
            self._arena_views[b['name']] = (b['off'], tuple(b['t'].shape), b['t'].dtype)
        self.ol.keep = [k for k in self.ol.keep if not any(k is b['t'] for b in items)]
        arena.plans.append(self)
        arena.require(total)
        self.rebind_arena()
        self.meta['arena_bytes'] = total
        self.meta['unpacked_bytes'] = sum(b['nbytes'] for b in items)

    def rebind_arena(self):
        if self.arena is None or self.arena.tensor is None:
            return
        base = self.arena.base()
        p = self.ol.arr['p']
        for (i, sl, off) in self._arena_slots:
            p[i, sl] = base + off
            self.ol.recs[i][4][sl] = base + off          # (a later finalize() rebuilds the array from the records)
        for name, (off, shape, dtype) in self._arena_views.items():
            n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            self.bufs[name] = self.arena.tensor[off:off + n].view(dtype).view(shape)

    graph_head = 0                       # leading launches that read caller-owned (volatile) pointers: kept out of the graph

    def run(self, **dyn):
        if not self.tuned:
            self.tuned = True
            self.autotune_convs(**dyn)
        if ARENA_POISON and self.arena is not None and self.arena.tensor is not None:
            self.arena.tensor.fill_(0xFF)               # (tests) nothing may rely on what an earlier run or another plan left in the arena
        ol = self.ol
        ex = _lib.get_executor()
        if GRAPHS and len(ol.recs) >= GRAPH_MIN_OPS and hasattr(ex, 'run_cached'):
            if ol.arr is None:
                ol.finalize()
            if dyn:
                ol.bind(**dyn)
            ex.run_cached(ol.arr, self._graphs, self.graph_head)   # replayed as one HIP graph once its pointer signature repeats
        else:
            ol.run(**dyn)

    def run_part(self, lo, hi, first=True, **dyn):
        """Launches [lo, hi) of the plan (None = to the end): a caller that has something to do between two parts of a stage
        (InferenceCore._add_memory: the bank insertion between the value encoder and the summarizer).  No graph replay."""
        if not self.tuned:
            self.tuned = True
            self.autotune_convs(**dyn)
        if first and ARENA_POISON and self.arena is not None and self.arena.tensor is not None:
            self.arena.tensor.fill_(0xFF)
        ol = self.ol
        if ol.arr is None:
            ol.finalize()
        if dyn:
            ol.bind(**dyn)
        _lib.get_executor().run(ol.arr[lo:hi])

    def autotune_convs(self, **dyn):
        """Pick the fastest tile shape for every conv of this plan by timing the candidates on the device
        (hipEvents on the launch stream, cutie_time_ops).  Called once per plan, at its first run, with the real
        dynamic buffers bound; results are cached per conv geometry on the engine.  No-op under the test interpreter."""
        from .. import _lib
        ex = _lib.get_executor()
        ex = getattr(ex, 'ex', ex)                       # bench.py's recording shim wraps the real executor
        if ex.is_mock or not hasattr(ex, 'time_ops'):
            return
        ol = self.ol
        if ol.arr is None:
            ol.finalize()
        ol.bind(**dyn)
        arr = ol.arr
        cache = self.eng.tile_cache
        tuned_any = False
        for n in range(len(arr)):
            if arr['kind'][n] != O.CONV:
                continue
            i = arr['i'][n]
            M, cout, cin = int(i[0]) * int(i[7]) * int(i[8]), int(i[9]), int(i[3]) + int(i[4])
            key = (M, cout, cin, int(i[11]), int(i[13]), int(arr['flags'][n]) & 3, int(i[1]), int(i[2]))
            best = cache.get(key)
            twin = self.korder_ref and int(i[0]) % self.korder_ref == 0     # (a lock-step plan also holds launches of ONE clip, B = 1: those are the one-clip plan's own geometry)
            if twin:
                # batched twin of a B = 1 plan (the look-ahead window of the image encoder): whatever the table says for the batched
                # geometry, the tile must sum over K in the order of the tile the B = 1 plan runs for this layer -- that keeps frame b
                # of the batch bit-identical to the same frame through the B = 1 plan
                Bk = self.korder_ref
                key1 = (M // Bk,) + key[1:]
                ref = cache.get(key1) or (O.COUT1_TILE if (int(i[17]) == O.COUT1_TILE) else
                                          O.pick_tile(M // Bk, cout, cin, dict(kh=int(i[11]), c2=int(i[4]))), 1)
                want = O.korder_class(*ref)
                if isinstance(want, str) and cache.get(key + (want,)) is not None:
                    best = cache[key + (want,)]              # a table entry FOR this class (two layers of one batched geometry may need different classes)
                if best is None or O.korder_class(*best) != want:
                    best = (int(i[17]), 1) if O.korder_class(int(i[17]), int(i[19])) == want else tuple(ref)
                assert O.korder_class(*best) == want, (key, best, ref)
            if best is None and not getattr(self.eng, 'autotune', AUTOTUNE):
                continue                                     # keep the deterministic static choice made when the plan was built
            if best is None:
                one = arr[n:n + 1].copy()
                best, best_t = (int(i[17]), 1), None
                cands = O.tile_candidates(M, cout, cin, int(i[16]),
                                          geom=dict(kh=int(i[11]), stride=int(i[13]), pad=int(i[14]), W=int(i[2]), c2=int(i[4])))
                if int(i[4]) or arr['p'][n, 4]:              # 2-source or residual conv: no cout1 kernel
                    cands = [t for t in cands if t != O.COUT1_TILE]
                for t in cands:
                    for sk in O.splitk_candidates(M, cout, int(i[16]), t):
                        one['i'][0, 17], one['i'][0, 19] = t, sk
                        ms = min(ex.time_ops(one, TUNE_ITERS) for _ in range(TUNE_REPS))
                        if best_t is None or ms < best_t:
                            best, best_t = (t, sk), ms
                cache[key] = best
                tuned_any = True
            apply = True
            if (best[0] in O.DMA_TILES or best[0] in O.PC_TILES) and not O.dma_tiles_enabled():
                apply = False
            if (arr['p'][n, 7] or arr['p'][n, 8]) and best[0] not in O.DMA_TILES and best[0] not in O.PC_TILES:
                apply = False                                # GAP accumulation / zero job exist in conv_dma_kernel only
            if apply:
                arr['i'][n, 17], arr['i'][n, 19] = best
            if twin:
                #  whatever path was taken above -- table entry, static choice kept, a skipped assignment --, the tile this
                # conv RUNS must sum over K like the one-frame plan's: that is what keeps a frame of the batch bit-identical to it
                assert O.korder_class(int(arr['i'][n, 17]), int(arr['i'][n, 19])) == want, (key, int(arr['i'][n, 17]), int(arr['i'][n, 19]), want)
        if TOUCH_REWIRE:
            ol.wire_next_weights()                           # the touch ranges follow the tiles the table has just put in place
        self.tuned = True
        if tuned_any:
            save_tile_cache(cache)

    # ---- conv helper ------------------------------------------------------------------
    def conv(self, wname, x, *, name=None, out=None, stride=1, pad=None, x2=None, res=None, res_bcast=False,
             relu_in=False, act=O.ACT_NONE, out_f32=False, ldy=None, gap_acc=None, zero=None, persistent=False, res_group=None):
        w = self.eng.w[wname]
        if pad is None:
            pad = (w.kh - 1) // 2
        OH = (x.H + 2 * pad - w.kh) // stride + 1
        OW = (x.W + 2 * pad - w.kw) // stride + 1
        if out is None:
            t = self.buf(name or wname, (x.B, OH, OW, w.cout), F32 if out_f32 else BF16, persistent=persistent)
            out = Act(t, x.B, OH, OW, w.cout, w.cout if ldy is None else ldy)
        else:
            assert out.B == x.B and out.H == OH and out.W == OW and out.C == w.cout, (wname, out.B, out.H, out.W, out.C, OH, OW, w.cout)
        if res is not None:
            assert res.C == w.cout and res.H == OH and res.W == OW
        self.ol.conv(x.t, w, out.t, B=x.B, H=x.H, W=x.W, C1=x.C, ldx1=x.ld, OH=OH, OW=OW, ldy=out.ld, stride=stride,
                     pad=pad, x2=None if x2 is None else x2.t, C2=0 if x2 is None else x2.C,
                     ldx2=0 if x2 is None else x2.ld, res=None if res is None else res.t,
                     ldr=0 if res is None else res.ld, res_bcast=res_bcast, relu_in=relu_in, act=act, out_f32=out_f32,
                     gap_acc=gap_acc, zero=zero, prio=self.prio, res_group=res_group)
        return out

    # ---- shared blocks ------------------------------------------------------------------
    def ca_block(self, prefix, x, name, out=None, head=None):
        """CAResBlock (channel_attn.py:7-39): conv3x3(relu) x2, ECA channel attention, residual.
        head = (weight name of a Cout = 1, 1x1 conv with fused input ReLU, its f32 output tensor): computed by the ECA launch."""
        if head is not None:
            head = (self.eng.w[head[0]], head[1])
        w = self.eng.w[prefix + '.conv2']
        gap = self.buf(name + '.gap', (x.B, x.C), F32)
        HW = x.H * x.W
        if not UNFUSED and O.conv_side_jobs_ok(cin=x.C, cout=w.cout, kh=w.kh):
            # ECA's global average pool rides on the convs: conv1 clears the accumulator, conv2 adds the per-object channel sums of
            # what it stores (fixed point, integer atomics: order-independent), ECA_APPLY turns them into means -- no GAP launch
            sums = self.buf(name + '.gapsum', (x.B, x.C), torch.int64)
            t1 = self.conv(prefix + '.conv1', x, name=name + '.t1', relu_in=True, act=O.ACT_RELU, zero=sums)
            t2 = self.conv(prefix + '.conv2', t1, name=name + '.t2', gap_acc=sums)
            if out is None:
                out = Act(self.buf(name + '.out', (x.B, x.H, x.W, x.C)), x.B, x.H, x.W, x.C)
            self.ol.eca_apply(t2.t, gap, self.eng.w[prefix + '.conv.weight'], x.t, out.t, B=x.B, HW=HW, C=x.C, fixed_sums=sums, head=head)
            return out
        t1 = self.conv(prefix + '.conv1', x, name=name + '.t1', relu_in=True, act=O.ACT_RELU)
        t2 = self.conv(prefix + '.conv2', t1, name=name + '.t2')
        self.ol.gap(t2.t, gap, B=x.B, HW=HW, C=x.C, partial_only=True)
        if out is None:
            out = Act(self.buf(name + '.out', (x.B, x.H, x.W, x.C)), x.B, x.H, x.W, x.C)
        self.ol.eca_apply(t2.t, gap, self.eng.w[prefix + '.conv.weight'], x.t, out.t, B=x.B, HW=HW, C=x.C, head=head)
        return out

    def fusion_block(self, prefix, x, g, name, out=None, xt=None, xt_group=None):
        """GroupFeatureFusionBlock (group_modules.py:102-127).  xt: x_transform(x) computed elsewhere (the encoder plan).
        xt_group = (objects per clip, rows between the clips' xt maps): clips in lock step -- every clip's objects add their clip's map."""
        if xt is None:
            xt = self.conv(prefix + '.distributor.x_transform', x, name=name + '.xt')
        g0 = self.conv(prefix + '.distributor.g_transform', g, name=name + '.g0', res=xt, res_bcast=True, res_group=xt_group)
        g1 = self.ca_block(prefix + '.block1', g0, name + '.b1')
        return self.ca_block(prefix + '.block2', g1, name + '.b2', out=out)

    def resnet(self, prefix, x, taps=None):
        """ResNet trunk after the stem/pool (resnet.py:51-124); taps: layer attr -> output Act override."""
        bottleneck, layers = resnet_layers(self.eng.m, prefix)
        feats = {}
        for (lname, planes, nb, lstride) in layers:
            for bi in range(nb):
                p = f'{prefix}.{lname}.{bi}'
                s = lstride if bi == 0 else 1
                last = bi == nb - 1
                out = taps.get(lname) if (taps and last) else None
                if bottleneck:
                    t1 = self.conv(p + '.conv1', x, act=O.ACT_RELU)
                    t2 = self.conv(p + '.conv2', t1, stride=s, act=O.ACT_RELU)
                    r = self.conv(p + '.downsample.0', x, stride=s) if (p + '.downsample.0') in self.eng.w else x
                    x = self.conv(p + '.conv3', t2, res=r, act=O.ACT_RELU, out=out)
                else:
                    t1 = self.conv(p + '.conv1', x, stride=s, act=O.ACT_RELU)
                    r = self.conv(p + '.downsample.0', x, stride=s) if (p + '.downsample.0') in self.eng.w else x
                    x = self.conv(p + '.conv2', t1, res=r, act=O.ACT_RELU, out=out)
            feats[lname] = x
        return x, feats


# ---------------------------------------------------------------------------------------------------
STEM = os.environ.get('CUTIE_AMD_STEM', '1') not in ('', '0')
STEM_BATCH = os.environ.get('CUTIE_AMD_STEM_BATCH', '1') not in ('', '0')     # the frames of an encoder window / the clips of a lock-step group in one STEM launch (A/B switch)


def stem_ok(eng, name):
    w = eng.w.get(name)
    return STEM and not UNFUSED and w is not None and w.cout == 64 and w.kh == 7 and w.kw == 7 and w.cin_padded == 8 and w.bias is not None


def build_encode(eng, h0, w0, H, W, pad_left, pad_top, B=1, qrows=64):
    """CUTIE.encode_image + transform_key (cutie.py:61-64,92-98; big_modules.py:45-54,81-87) + query-side
    similarity operands.  dyn in: image f32 [3,h0,w0].  dyn out: f16,f8,f4,pix_feat (bf16 NHWC), key,shr,sel
    (f32 [hw,*]), Bhi,Blo (bf16 [HWp,128]), cq (f32 [HWp]).
    B > 1 (the look-ahead WINDOW of InferenceCore, no counterpart in the reference): B frames `image0` .. `image<B-1>` through ONE
    plan -- every conv sees B x the rows (M = B * OH * OW; at 480p the stride-16 layers of one frame have 1620 rows, a third of a
    round of workgroups), the outputs are [B, ...] with frame b's slice laid out exactly like the B = 1 output.  Rows of different
    frames never meet (batch = outermost dimension of NHWC) and the tiles are taken from the same K-order class as the B = 1 plan's
    (Plan.korder_ref), so frame b's results are bit-identical to the B = 1 plan's.
    qrows: every frame's query operands (Bhi, Blo, cq) are padded to a multiple of this many rows (zero rows; 64 = a wave's queries; clips in lock
    step ask for 128 -- one bank per 128-row block of a joint read-out pass, AFF_SCORE flags&4)."""
    P = Plan(eng, touch=weights_go_cold(B * (H // 16) * (W // 16)), prio=False)      # (mostly on a look-ahead stream: yields to the frame's own launches)
    m = eng.m
    img = (lambda b: Dyn('image')) if B == 1 else (lambda b: Dyn('image%d' % b))
    if stem_ok(eng, 'pixel_encoder.conv1'):                 # IMG_PREP + 7x7 conv + max pool in one launch (csrc/stem.hip)
        pool = P.buf('pool', (B, H // 4, W // 4, 64))
        SM = O.OpList.STEM_MAX_IMAGES if STEM_BATCH else 1
        for b0 in range(0, B, SM):                          # the frames of a window in launches of up to 12 (one frame is a partial round of blocks)
            P.ol.stem(img(b0), None, eng.w['pixel_encoder.conv1'], pool[b0:], h0=h0, w0=w0, H=H, W=W, pad_left=pad_left, pad_top=pad_top, K=1,
                      mean=m['pixel_mean'], std=m['pixel_std'], relu=True, more_images=[img(b) for b in range(b0 + 1, min(B, b0 + SM))])
        x = Act(pool, B, H // 4, W // 4, 64)
    else:
        img8 = P.buf('img8', (B, H, W, 8))
        for b in range(B):
            P.ol.img_prep(img(b), None, img8[b], h0=h0, w0=w0, H=H, W=W, pad_left=pad_left, pad_top=pad_top, K=1,
                          mean=m['pixel_mean'], std=m['pixel_std'])
        x = P.conv('pixel_encoder.conv1', Act(img8, B, H, W, 8), stride=2, pad=3, act=O.ACT_RELU)
        pool = P.buf('pool', (B, x.H // 2, x.W // 2, 64))
        P.ol.maxpool(x.t, pool, B=B, H=x.H, W=x.W, C=64)
        x = Act(pool, B, x.H // 2, x.W // 2, 64)
    h4, w4, h8, w8, h, w = H // 4, W // 4, H // 8, W // 8, H // 16, W // 16
    ms = m['pixel_encoder']['ms_dims']
    taps = {'res2': Act(Dyn('f4'), B, h4, w4, ms[2]), 'layer2': Act(Dyn('f8'), B, h8, w8, ms[1]),
            'layer3': Act(Dyn('f16'), B, h, w, ms[0])}
    f16, _ = P.resnet('pixel_encoder', x, taps)
    pix = P.conv('pix_feat_proj', f16, out=Act(Dyn('pix_feat'), B, h, w, m['pixel_dim']))
    build_key_ops(P, f16, h, w, qrows)
    # Image-only convolutions of the mask decoder and the pixel fuser (DecoderFeatureProcessor, big_modules.py:244-255; the
    # x_transform of MainToGroupDistributor, group_modules.py:112-115): they do not depend on the memory, so they run here -- with the
    # look-ahead encoder on the side stream, off the frame's critical path -- and reach their consumers through frame_context.
    W_ = eng.w
    for wname, src, dname in (('mask_decoder.decoder_feat_proc.transforms.0', taps['layer2'], 'f8p'),
                              ('mask_decoder.decoder_feat_proc.transforms.1', taps['res2'], 'f4p'),
                              ('pixel_fuser.fuser.distributor.x_transform', pix, 'fuse_xt')):
        P.conv(wname, src, out=Act(Dyn(dname), B, src.H, src.W, W_[wname].cout))
    P.meta.update(h=h, w=w, B=B)
    if B > 1:
        P.korder_ref = B
    return P


def build_key_ops(P, f16, h, w, qrows=64):
    m = P.eng.m
    CK = m['key_dim']
    B = f16.B
    kx = P.conv('key_proj.pix_feat_proj', f16, name='keyx')
    P.conv('key_proj.key_proj', kx, out=Act(Dyn('key'), B, h, w, CK), out_f32=True)
    P.conv('key_proj.d_proj', kx, out=Act(Dyn('shr'), B, h, w, 1), out_f32=True, act=O.ACT_SQ1)
    P.conv('key_proj.e_proj', kx, out=Act(Dyn('sel'), B, h, w, CK), out_f32=True, act=O.ACT_SIGMOID)
    hw, HWp = h * w, -(-h * w // qrows) * qrows
    for b in range(B):                                       # (the padding rows [hw, HWp) of every frame's operands stay zero)
        P.ol.key_prep(Dyn('key', b * hw * CK * 4), Dyn('sel', b * hw * CK * 4), Dyn('Bhi', b * HWp * 128 * 2), Dyn('Blo', b * HWp * 128 * 2),
                      Dyn('cq', b * HWp * 4), n=hw, query=True)


def build_transform_key(eng, h, w):
    """Stand-alone CUTIE.transform_key for callers that bring their own f16 (dyn in: f16)."""
    P = Plan(eng)
    build_key_ops(P, Act(Dyn('f16'), 1, h, w, eng.m['pixel_encoder']['ms_dims'][0]), h, w)
    return P


def build_pixel_fusion(eng, K, h, w, pre=False, pre_md=False, clips=1, wstride=1, stacked=False):
    """CUTIE.pixel_fusion (cutie.py:142-157; big_modules.py:207-235).
    dyn in: pix_feat, pixel (readout) bf16 [K,h,w,CV], sensory_bf16 [K,h,w,CS], last_mask f32 [K,16h,16w];
    pre: fuse_xt bf16 [1,h,w,CE] = x_transform(pix_feat) from the encoder plan instead of pix_feat.
    dyn out: fused bf16 [K,h,w,CE].
    clips > 1 (clips in lock step, cutie_amd/inference/lockstep.py; no counterpart in the reference, which runs one InferenceCore per
    video): `clips` independent clips of K objects each through ONE plan -- every per-object tensor has clips * K rows (clip-major), the
    per-clip tensors (fuse_xt: needs pre) are the slices of one frame in the encoder window's output, `wstride` frames apart; the read-outs
    of the clips (dyn in pixel0 .. pixel<clips-1>, one tensor per memory bank) are gathered by the first launch -- or, stacked, come as
    one tensor `pixel` [clips * K, h, w, CV] (a read-out pass over all banks, MemoryManager.prefetch_affinity_joint); dyn in last_mask0.. when
    the masks were not down-sampled by the segment in front (pre_md).  Per clip the launches compute what the one-clip plan computes."""
    G, Kc = clips, K
    K = G * Kc
    assert G == 1 or pre, 'clips in lock step take x_transform(pix_feat) from the encoder plan'
    P = Plan(eng, touch=weights_go_cold(h * w, K))
    m = eng.m
    if pre_md:                                           # written by the previous frame's up-sampling launch (build_segment(md=True))
        pair = eng.mask_down_bufs(K, h, w, G)[0]
        P.ol.keep.append(pair)
    else:
        pair = P.buf('pair', (K, h, w, 64), persistent=True)       # (mask, others) in channels 0, 1 of a ZEROED 64-channel tensor: a whole K tile
        m16 = P.buf('m16', (K, h, w), F32)
        for c in range(G):                               # ("others" = the other objects of the SAME clip: one launch per clip)
            P.ol.mask_down(Dyn('last_mask' if G == 1 else 'last_mask%d' % c), pair[c * Kc:], m16[c * Kc:], K=Kc, H=16 * h, W=16 * w, pair_channels=64)
    if G > 1 and not stacked:
        px = P.buf('pixel_all', (K, h, w, m['value_dim']))
        nb = Kc * h * w * m['value_dim'] * 2
        P.ol.bank_write([(Dyn('pixel%d' % c), px[c * Kc:], nb) for c in range(G)])
        pixel = Act(px, K, h, w, m['value_dim'])
    else:
        pixel = Act(Dyn('pixel'), K, h, w, m['value_dim'])
    p16 = P.conv('pixel_fuser.sensory_compress', Act(Dyn('sensory_bf16'), K, h, w, m['sensory_dim']),
                 x2=Act(pair, K, h, w, 64), res=pixel, name='p16')
    xt = Act(Dyn('fuse_xt'), G, h, w, eng.w['pixel_fuser.fuser.distributor.x_transform'].cout) if pre else None
    P.fusion_block('pixel_fuser.fuser', Act(Dyn('pix_feat'), 1, h, w, m['pixel_dim']), p16, 'fuse',
                   out=Act(Dyn('fused'), K, h, w, m['embed_dim']), xt=xt, xt_group=(Kc, wstride * h * w) if G > 1 else None)
    if G > 1:
        P.korder_ref = G                                 # tiles from the K-order class of the one-clip plan's: same bits per clip
    return P


def build_readout_query(eng, K, h, w, last_aux=True, fresh=True, clips=1):
    """CUTIE.readout_query -> QueryTransformer.forward (object_transformer.py:114-177).
    dyn in: pixel bf16 [K,h,w,C], obj_mem f32 [K,Q,C+1].  dyn out: out bf16 [K,h,w,C].
    Aux logits of every block are kept in bufs['aux_logits'] (f32 [blocks+1,K,hw]) for tests.  last_aux=False skips the mask_pred
    head after the LAST block: its logits mask no attention any more (the reference computes them anyway, object_transformer.py:164,
    and only save_aux / training read them).
    clips > 1: that many clips of K objects in lock step (see build_pixel_fusion) -- every launch of the transformer is per object; the
    foreground masks are decided among the objects of one clip (ATTN_Q2P i9)."""
    G, Kc = clips, K
    K = G * Kc
    P = Plan(eng, touch=weights_go_cold(h * w, K))
    if G > 1:
        P.korder_ref = G
    m, ol, W = eng.m, P.ol, eng.w
    ot = m['object_transformer']
    C, Q, heads, nb = m['embed_dim'], ot['num_queries'], ot['num_heads'], ot['num_blocks']
    HW, M = h * w, K * Q
    t = 'object_transformer'
    f = lambda name, shape: P.buf(name, shape, F32)
    query, query_emb = eng.query_bufs(K)                  # (engine-level: the variant with fresh=False reads what the other one wrote)
    use_chain = QCHAIN and not UNFUSED and C == 256 and Q == 16 and heads == 8 and HW <= 24576 and ot['ff_dim'] % QFFN_SLICE == 0
    assert G == 1 or use_chain, 'clips in lock step need the chain form of the query side'
    # fixed-point accumulators of the query chain (three per block), cleared by the first launch of the plan
    qacc = P.buf('qacc', (3 * nb, M, C), torch.int64) if use_chain else None
    zero_on_conv = None
    if C == 256 and Q == 16 and not UNFUSED:
        if fresh or not use_chain:
            ol.query_init2(Dyn('obj_mem'), query, query_emb, rows=M, w_init=W[t + '.summary_to_query_init'], res_init=eng.rep_embedding('query_init', K),
                           w_emb=W[t + '.summary_to_query_emb'], res_emb=eng.rep_embedding('query_emb', K), zero=qacc)
        elif O.conv_side_jobs_ok(cin=C, cout=C, kh=1):
            zero_on_conv = qacc        # fresh=False: the summaries have not changed since the last run -- the queries in query_bufs are
                                       # still theirs; the accumulators are cleared by the first conv of the plan instead
        else:
            # (with $CUTIE_AMD_DMA_TILES=0 that conv runs on a register-staged tile, which has no side jobs -- the clearing
            # gets a small launch of its own, like ca_block does for its GAP accumulators)
            ol.memset32(qacc, 2 * qacc.numel(), 0)
    else:
        vals = f('vals', (M, C))
        ol.query_init(Dyn('obj_mem'), vals, rows=M, C=C)
        ol.linear(vals, W[t + '.summary_to_query_init'], query, M=M, res=eng.rep_embedding('query_init', K))
        ol.linear(vals, W[t + '.summary_to_query_emb'], query_emb, M=M, res=eng.rep_embedding('query_emb', K))
    pix_in = Act(Dyn('pixel'), K, h, w, C)
    if UNFUSED:
        # pixel = pixel_init_proj(x), pixel_pe = pixel_emb_proj(x) + PE: one conv, two channel slices of its output
        both = P.conv(t + '.pixel_init_emb', pix_in, name='pixel_init_emb', res=Act(eng.pe0(h, w), 1, h, w, 2 * C), res_bcast=True)
        pixel = Act(both.t, K, h, w, C, 2 * C)
        pixel_pe = Act(both.t.view(-1)[C:], K, h, w, C, 2 * C)
        R_all = P.conv(t + '.pe_proj_all', pixel_pe, name='R_all')               # [Wk.pe | 0 | Wq2.pe] of every block
        R_of = lambda b: Act(R_all.t.view(-1)[b * 3 * C:], K, h, w, 3 * C, nb * 3 * C)
    elif PROJ_X:
        # the positional terms live inside each block's projection (Engine: '.pixel_proj_x'): only pixel = pixel_init_proj(x) here
        pixel = P.conv(t + '.pixel_init_proj', pix_in, name='pixel_init', zero=zero_on_conv)
        R_of = None
    else:
        # pixel_pe feeds nothing but the positional terms R_b, so its projection is composed with theirs at load time
        # (Engine: '.pixel_init_R'): ONE conv x -> [pixel | R_0 | R_1 | ...], the PE part of R as a per-pixel broadcast residual
        CR = C + nb * 3 * C
        both = P.conv(t + '.pixel_init_R', pix_in, name='pixel_init_R', res=Act(eng.pe_r(h, w), 1, h, w, CR), res_bcast=True, zero=zero_on_conv)
        pixel = Act(both.t, K, h, w, C, CR)
        R_of = lambda b: Act(both.t.view(-1)[C + b * 3 * C:], K, h, w, 3 * C, CR)
    aux = P.buf('aux_logits', (nb + 1, K, HW), F32, persistent=True)      # read by the caller after the run
    fused_mask = HW <= 24576 and not UNFUSED                      # ATTN_Q2P derives the foreground mask from the logits itself (flags in LDS)
    fg = None if fused_mask else P.buf('fg', (K, HW), torch.uint8)
    nfg = None if fused_mask else P.buf('nfg', (K,), torch.int32)
    P.conv(t + '.mask_pred.0.1', pixel, relu_in=True, out_f32=True, out=Act(aux[0], K, h, w, 1))
    if not fused_mask:
        ol.aux_mask(aux[0], fg, nfg, K=K, HW=HW)
    x = query
    prev_acc = q_pre = xn_next = None
    for b in range(nb):
        q = f'{t}.blocks.{b}'
        n = f'b{b}.'
        if R_of is None:                                         # k | v | q2 of the pixels, positional terms included: [pixel_b | x] -> 3C
            kvq = P.conv(q + '.pixel_proj_x', pixel, x2=pix_in, name=n + 'kvq', res=Act(eng.pe_rb(h, w, b), 1, h, w, 3 * C), res_bcast=True)
        else:
            kvq = P.conv(q + '.pixel_proj', pixel, name=n + 'kvq', res=R_of(b))    # k | v | q2 of the pixels
        # read_from_pixel (CrossAttention, transformer_layers.py:75-98): residual is the normed x.  Every LayerNorm of the
        # block is fused into the linear that consumes it (the normalised rows are kept where the reference reuses them).
        ln = lambda name: (W[q + name + '.weight'], W[q + name + '.bias'])
        xn = xn_next if xn_next is not None else f(n + 'xn', (M, C))
        xn_next = None
        fuse_proj = fused_mask and C == 256 and not UNFUSED      # the small projections run inside the attention launches
        chain = fuse_proj and use_chain
        if chain:
            # The query side of a block in FOUR launches (was seven): each attention launch also applies its output projection, per
            # head, summed over the heads into a fixed-point accumulator; the next launch adds it (+ bias + residual) while it stages
            # its rows.  The FFN is one launch over slices of its hidden layer, summed the same way (csrc/qchain.hip).
            Wo1, Wo2 = W[q + '.read_from_pixel.out'], W[q + '.self_attn.out']
            W1, W2 = W[q + '.ffn.linear1'], W[q + '.ffn.linear2']
            a1, a2, a3 = qacc[3 * b], qacc[3 * b + 1], qacc[3 * b + 2]
            if q_pre is not None:                                  # projected by the previous block's ATTN_P2Q launch (xn too)
                ol.attn_q2p(None, kvq.t, None, None, None, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=aux[b],
                            q_pre=q_pre, out_proj=(Wo1, a1), clip_objects=Kc)
            else:
                ol.attn_q2p(None, kvq.t, None, None, None, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=aux[b],
                            proj=dict(x=x, W=W[q + '.read_from_pixel.q'], emb=query_emb, ln=ln('.read_from_pixel.norm'), ln_out=xn),
                            acc_in=prev_acc, out_proj=(Wo1, a1), clip_objects=Kc)
            y = f(n + 'y', (M, C))
            ol.attn_self(None, None, None, K=K, Q=Q, C=C, heads=heads,
                         proj=dict(x=xn, W=W[q + '.self_attn.qkv'], emb=query_emb, ln=ln('.self_attn.norm'), ln_out=y),
                         acc_in=(a1, Wo1.bias), out_proj=(Wo2, a2))
            x2 = f(n + 'x2', (M, C))
            ol.qffn(y, x2, a3, rows=M, ln=ln('.ffn.norm'), W1=W1, W2=W2, acc_in=(a2, Wo2.bias), hid_slice=QFFN_SLICE)
            x = x2                                                 # + b2 + a3 / 2^32: added by whoever reads x
            prev_acc = (a3, W2.bias)
            next_q = None
            if b + 1 < nb and QNEXT:
                qn = f'{t}.blocks.{b + 1}'
                q_pre, xn_next = f(f'b{b + 1}.q', (M, C)), f(f'b{b + 1}.xn', (M, C))
                next_q = dict(ln=(W[qn + '.read_from_pixel.norm.weight'], W[qn + '.read_from_pixel.norm.bias']), W=W[qn + '.read_from_pixel.q'],
                              q_out=q_pre, xn_out=xn_next)
            else:
                q_pre = None
            # pixel + out_proj(attention) by the same launch: no 1x1 conv behind it.  The kernel exists in the diagnostic library only (make DIAG=1):
            # with the product library the switch is ignored, with a warning (it used to fail at launch time in the middle of a frame)
            fuse_out = P2Q_OUT and pixel.ld == C and pixel.B == K and _p2q_out_available()
            pf = Act(P.buf(n + 'pf', (K, h, w, C)), K, h, w, C) if fuse_out else None
            pa = None if fuse_out else P.buf(n + 'pa', (K, h, w, C))
            ol.attn_p2q(kvq.t.view(-1)[2 * C:], None, None, pf.t if fuse_out else pa, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C,
                        proj=dict(x=x, W=W[q + '.read_from_query.kv'], emb=query_emb), acc_in=prev_acc, next_q=next_q,
                        out=dict(Wo=W[q + '.read_from_query.out_blob'], res=pixel.t) if fuse_out else None)
        else:
            pf = None
            att = f(n + 'att', (M, C))
            if fuse_proj:
                ol.attn_q2p(None, kvq.t, None, None, att, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=aux[b],
                            proj=dict(x=x, W=W[q + '.read_from_pixel.q'], emb=query_emb, ln=ln('.read_from_pixel.norm'), ln_out=xn))
            else:
                qp = f(n + 'qp', (M, C))
                ol.linear(x, W[q + '.read_from_pixel.q'], qp, M=M, x_add=query_emb, add_rows=M, ln=ln('.read_from_pixel.norm'), ln_out=xn)
                ol.attn_q2p(qp, kvq.t, fg, nfg, att, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=aux[b] if fused_mask else None)
            x1 = f(n + 'x1', (M, C))
            ol.linear(att, W[q + '.read_from_pixel.out'], x1, M=M, res=xn)
            # self attention (transformer_layers.py:28-41): q | k | v in one launch, the query PE feeds q and k only
            y = f(n + 'y', (M, C))
            sa = f(n + 'sa', (M, C))
            if fuse_proj:
                ol.attn_self(None, None, sa, K=K, Q=Q, C=C, heads=heads,
                             proj=dict(x=x1, W=W[q + '.self_attn.qkv'], emb=query_emb, ln=ln('.self_attn.norm'), ln_out=y))
            else:
                qkv = f(n + 'qkv', (M, 3 * C))
                ol.linear(x1, W[q + '.self_attn.qkv'], qkv, M=M, x_add=query_emb, add_rows=M, add_cols=2 * C, ln=ln('.self_attn.norm'), ln_out=y)
                ol.attn_self(qkv, qkv.view(-1)[2 * C:], sa, K=K, Q=Q, C=C, heads=heads, ldqk=3 * C, ldv=3 * C)
            x2 = f(n + 'x2', (M, C))
            ol.linear(sa, W[q + '.self_attn.out'], x2, M=M, res=y)
            # FFN (transformer_layers.py:113-118)
            hid = f(n + 'hid', (M, ot['ff_dim']))
            ol.linear(x2, W[q + '.ffn.linear1'], hid, M=M, relu=True, ln=ln('.ffn.norm'))
            x3 = f(n + 'x3', (M, C))
            ol.linear(hid, W[q + '.ffn.linear2'], x3, M=M, res=x2)
            x = x3
            # read_from_query (no norm, residual on the pixels): k | v of the queries in one launch
            pa = P.buf(n + 'pa', (K, h, w, C))
            if fuse_proj:
                ol.attn_p2q(kvq.t.view(-1)[2 * C:], None, None, pa, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C,
                            proj=dict(x=x, W=W[q + '.read_from_query.kv'], emb=query_emb))
            else:
                kv2 = f(n + 'kv2', (M, 2 * C))
                ol.linear(x, W[q + '.read_from_query.kv'], kv2, M=M, x_add=query_emb, add_rows=M, add_cols=C)
                ol.attn_p2q(kvq.t.view(-1)[2 * C:], kv2, kv2.view(-1)[C:], pa, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C, ldkv=2 * C)
        if pf is None:
            pf = P.conv(q + '.read_from_query.out', Act(pa, K, h, w, C), name=n + 'pf', res=pixel)
        # PixelFFN (transformer_layers.py:121-136)
        last = b == nb - 1
        want_aux = not last or last_aux
        fuse_head = want_aux and ECA_HEAD and C == 256 and not UNFUSED        # the mask_pred head rides on the block's ECA launch
        pixel = P.ca_block(q + '.pixel_ffn.conv', pf, n + 'ffn', out=Act(Dyn('out'), K, h, w, C) if last else None,
                           head=(f'{t}.mask_pred.{b + 1}.1', aux[b + 1]) if fuse_head else None)
        if want_aux and not fuse_head:
            P.conv(f'{t}.mask_pred.{b + 1}.1', pixel, relu_in=True, out_f32=True, out=Act(aux[b + 1], K, h, w, 1))
        if not last and not fused_mask:
            ol.aux_mask(aux[b + 1], fg, nfg, K=K, HW=HW)
    return P


def build_segment(eng, K, h, w, update_sensory, pre=False, md=False, clips=1, wstride=1):
    """CUTIE.segment -> MaskDecoder.forward + sigmoid/aggregate/x4/softmax (cutie.py:172-203;
    big_modules.py:257-306; modules.py:8-68).
    dyn in: f8, f4 (bf16), p16 (memory readout) bf16 [K,h,w,C], sensory_f32 / sensory_bf16 [K,h,w,CS] (in-place).
    pre: f8p, f4p (decoder_feat_proc of f8 / f4, from the encoder plan) instead of f8, f4.
    dyn out: prob f32 [K+1,16h,16w] (+ logits_up if bound).
    clips > 1: that many clips of K objects in lock step (see build_pixel_fusion; needs pre): f8p / f4p are the slices of one frame in the
    encoder window's output, `wstride` frames apart; dyn out prob f32 [clips, K+1, 16h, 16w]."""
    G, Kc = clips, K
    K = G * Kc
    assert G == 1 or (pre and Kc + 1 <= 8 and not UNFUSED), 'clips in lock step: image-only decoder convs from the encoder plan, <= 7 objects per clip'
    P = Plan(eng, touch=weights_go_cold(h * w, K))
    if G > 1:
        P.korder_ref = G
    m, ol = eng.m, P.ol
    up = m['mask_decoder']['up_dims']
    ms = m['pixel_encoder']['ms_dims']
    CS = m['sensory_dim']
    h8, w8, h4, w4 = 2 * h, 2 * w, 4 * h, 4 * w
    if pre:
        f8p, f4p = Act(Dyn('f8p'), G, h8, w8, up[0]), Act(Dyn('f4p'), G, h4, w4, up[1])
    else:
        f8p = P.conv('mask_decoder.decoder_feat_proc.transforms.0', Act(Dyn('f8'), 1, h8, w8, ms[1]), name='f8p')
        f4p = P.conv('mask_decoder.decoder_feat_proc.transforms.1', Act(Dyn('f4'), 1, h4, w4, ms[2]), name='f4p')
    p16 = Act(Dyn('p16'), K, h, w, up[0])
    u8 = P.buf('u8', (K, h8, w8, up[0]))
    ol.upsample2x_add(p16.t, f8p.t, u8, B=K, h=h, w=w, C=up[0], skip_group=(Kc, wstride * h8 * w8) if G > 1 else None)
    u8 = Act(u8, K, h8, w8, up[0])
    t1 = P.conv('mask_decoder.up_16_8.out_conv.conv1', u8, relu_in=True, act=O.ACT_RELU)
    ds = P.conv('mask_decoder.up_16_8.out_conv.downsample', u8)
    p8 = P.conv('mask_decoder.up_16_8.out_conv.conv2', t1, res=ds, name='p8')
    u4 = P.buf('u4', (K, h4, w4, up[1]))
    ol.upsample2x_add(p8.t, f4p.t, u4, B=K, h=h8, w=w8, C=up[1], skip_group=(Kc, wstride * h4 * w4) if G > 1 else None)
    u4 = Act(u4, K, h4, w4, up[1])
    t2 = P.conv('mask_decoder.up_8_4.out_conv.conv1', u4, relu_in=True, act=O.ACT_RELU)
    p4 = P.conv('mask_decoder.up_8_4.out_conv.conv2', t2, res=u4, name='p4')
    logits = P.buf('logits', (K, h4, w4), F32)
    P.conv('mask_decoder.pred', p4, relu_in=True, out_f32=True, out=Act(logits, K, h4, w4, 1))
    P.meta['logits_done'] = len(ol.recs)              # two independent branches follow: the sensory update, and the up-sampling + softmax (LAST launch)
    if update_sensory:
        # area-pooled g8 / g4 / logits written side by side: the second source of the ONE conv that replaces g16_conv + g8_conv +
        # g4_conv (Engine: '.g_all'); the logits take a whole 64-channel K tile (channel 0 written, the rest stays zero)
        CT = up[1] + up[2] + 64
        gcat = P.buf('gcat', (K, h, w, CT), persistent=True)          # (the padding channels of the logits tile are zero from the allocation)
        ol.area_down3([dict(x=p8.t, y=gcat, B=K, H=h8, W=w8, C=up[1], ldx=up[1], ldy=CT, r=2),
                       dict(x=p4.t, y=gcat.view(-1)[up[1]:], B=K, H=h4, W=w4, C=up[2], ldx=up[2], ldy=CT, r=4),
                       dict(x=logits, y=gcat.view(-1)[up[1] + up[2]:], B=K, H=h4, W=w4, C=1, ldx=1, ldy=CT, r=4, f32_in=True, Cz=8)])
        g3 = P.conv('mask_decoder.sensory_update.g_all', p16, x2=Act(gcat, K, h, w, CT), name='g3')
        vals = P.conv('mask_decoder.sensory_update.transform', g3, x2=Act(Dyn('sensory_bf16'), K, h, w, CS),
                      out_f32=True, name='gru_vals')
        ol.gru(vals.t, Dyn('sensory_f32'), Dyn('sensory_bf16'), n=K * h * w, C=CS)
    if Kc + 1 <= 16 and not UNFUSED:              # aggregation recomputed per bilinear tap inside the up-sampling launch
        # md: the launch also derives what the NEXT frame's pixel fusion needs from these probabilities (MASK_DOWN: the stride-16 means
        # of the object planes and the (mask, others) pairs), into buffers both plans know (Engine.mask_down_bufs)
        ol.up4_softmax(logits, Dyn('prob'), Dyn('logits_up'), P=Kc + 1, h=h4, w=w4, from_logits=True, clips=G,
                       mask_down=(eng.mask_down_bufs(K, h, w, G)[1], eng.mask_down_bufs(K, h, w, G)[0], 64) if md else None)
    else:
        agg = P.buf('agg', (K + 1, h4, w4), F32)
        ol.seg_agg(logits, agg, K=K, hw=h4 * w4)
        ol.up4_softmax(agg, Dyn('prob'), Dyn('logits_up'), P=K + 1, h=h4, w=w4)
    return P


def build_encode_mask(eng, K, h0, w0, H, W, pad_left, pad_top, deep_update=True, pre_md=False, clips=1, wstride=1):
    """CUTIE.encode_mask -> MaskEncoder.forward + ObjectSummarizer.forward (cutie.py:66-90;
    big_modules.py:122-182; object_summarizer.py:55-89).
    dyn in: image f32 [3,h0,w0], masks f32 [K,H,W], pix_feat, sensory_f32/sensory_bf16 (in-place deep update).
    dyn out: value bf16 [K,h,w,CV], summ f32 [K,Q,C+1].
    clips > 1: that many clips of K objects in lock step (see build_pixel_fusion): dyn in image0 .., masks = the probabilities
    f32 [clips, K+1, H, W] of the segment in front (clip c's object planes at [c, 1:]), pix_feat = clip 0's slice of the encoder window's
    output (`wstride` frames between the clips); value / summ / the sensory state hold clips * K objects."""
    G, Kc = clips, K
    K = G * Kc
    P = Plan(eng, touch=weights_go_cold((H // 16) * (W // 16), K))
    if G > 1:
        P.korder_ref = G
    m, ol = eng.m, P.ol
    h, w = H // 16, W // 16
    CV, CS, CE, Q = m['value_dim'], m['sensory_dim'], m['embed_dim'], m['object_summarizer']['num_summaries']
    masks_of = (lambda c: Dyn('masks')) if G == 1 else (lambda c: Dyn('masks', 4 * (c * (Kc + 1) + 1) * H * W))
    assert G == 1 or stem_ok(eng, 'mask_encoder.conv1')
    if stem_ok(eng, 'mask_encoder.conv1'):
        pool = P.buf('pool', (K, H // 4, W // 4, 64))
        # ("others" = the other objects of the same clip, one frame per clip: the clips are the FRAMES of one launch -- or a launch per clip)
        iname = lambda c: Dyn('image' if G == 1 else 'image%d' % c)
        SM = O.OpList.STEM_MAX_IMAGES if STEM_BATCH else 1
        for c0 in range(0, G, SM):
            ol.stem(iname(c0), masks_of(c0), eng.w['mask_encoder.conv1'], pool[c0 * Kc:], h0=h0, w0=w0, H=H, W=W, pad_left=pad_left, pad_top=pad_top,
                    K=Kc, mean=m['pixel_mean'], std=m['pixel_std'], relu=True, more_images=[iname(c) for c in range(c0 + 1, min(G, c0 + SM))],
                    mask_stride=(Kc + 1) * H * W)
    else:
        x8 = P.buf('x8', (K, H, W, 8))
        ol.img_prep(Dyn('image'), Dyn('masks'), x8, h0=h0, w0=w0, H=H, W=W, pad_left=pad_left, pad_top=pad_top, K=K,
                    mean=m['pixel_mean'], std=m['pixel_std'])
        x = P.conv('mask_encoder.conv1', Act(x8, K, H, W, 8), stride=2, pad=3)          # conv+bn, then maxpool, then relu
        pool = P.buf('pool', (K, x.H // 2, x.W // 2, 64))
        ol.maxpool(x.t, pool, B=K, H=x.H, W=x.W, C=64, relu=True)
    g16, _ = P.resnet('mask_encoder', Act(pool, K, H // 4, W // 4, 64))
    xt = xt_group = None
    if G > 1:                                            # x_transform of every clip's pix_feat (rows of one clip only: M = hw each)
        wx = eng.w['mask_encoder.fuser.distributor.x_transform']
        xt_t = P.buf('fuse.xt', (G, h, w, wx.cout))
        for c in range(G):
            P.conv('mask_encoder.fuser.distributor.x_transform', Act(Dyn('pix_feat', 2 * c * wstride * h * w * m['pixel_dim']), 1, h, w, m['pixel_dim']),
                   out=Act(xt_t[c:c + 1], 1, h, w, wx.cout))
        xt, xt_group = Act(xt_t, G, h, w, wx.cout), (Kc, h * w)
    value = P.fusion_block('mask_encoder.fuser', Act(Dyn('pix_feat'), 1, h, w, m['pixel_dim']), g16, 'fuse',
                           out=Act(Dyn('value'), K, h, w, CV), xt=xt, xt_group=xt_group)
    P.meta['value_done'] = len(ol.recs)               # what follows reads named tensors only (value, masks, sensory): see CUTIE._encode_mask_split
    if deep_update:
        vals = P.conv('mask_encoder.sensory_update.transform', value, x2=Act(Dyn('sensory_bf16'), K, h, w, CS),
                      out_f32=True, name='gru_vals')
        ol.gru(vals.t, Dyn('sensory_f32'), Dyn('sensory_bf16'), n=K * h * w, C=CS)
    # object summarizer
    if pre_md:                                           # written by the last segment's up-sampling launch (build_segment(md=True)): these very masks
        m16 = eng.mask_down_bufs(K, h, w, G)[1]
    else:
        pair = P.buf('pair', (K, h, w, 8))
        m16 = P.buf('m16', (K, h, w), F32)
        for c in range(G):
            ol.mask_down(masks_of(c), pair[c * Kc:], m16[c * Kc:], K=Kc, H=H, W=W)
    if SUM_FUSED and not UNFUSED:
        # two launches instead of five (Engine: '.in_fw0', '.fw2'): value -> [f1 | w1] with the composed weights, then the block-diagonal
        # second layers -> fp32 [feature | weight logits]
        fw1 = P.conv('object_summarizer.in_fw0', value, res=Act(eng.pe_sum(h, w), 1, h, w, 2 * CE), res_bcast=True, act=O.ACT_RELU, name='sum.fw1')
        fw2 = P.conv('object_summarizer.fw2', fw1, out_f32=True, name='sum.fw2')
        ol.summarize(fw2.t, fw2.t.view(-1)[CE:], m16, Dyn('summ'), K=K, HW=h * w, C=CE, Q=Q, feat_f32=True, ldf=CE + Q, ldw=CE + Q)
        return P
    pe = Act(eng.pe(h, w), 1, h, w, CE)
    v1 = P.conv('object_summarizer.input_proj', value, res=pe, res_bcast=True, name='sum.v1')
    f1 = P.conv('object_summarizer.feature_pred.0', v1, act=O.ACT_RELU, name='sum.f1')
    feat = P.conv('object_summarizer.feature_pred.2', f1, name='sum.feat')
    w1 = P.conv('object_summarizer.weights_pred.0', v1, act=O.ACT_RELU, name='sum.w1')
    wl = P.conv('object_summarizer.weights_pred.2', w1, out_f32=True, name='sum.wl')
    ol.summarize(feat.t, wl.t, m16, Dyn('summ'), K=K, HW=h * w, C=CE, Q=Q)
    return P


def build_mask_to_prob(eng, Knew, Kold, h0, w0, H, W, pad_left, pad_top, float_mode, nfloat):
    """Input-mask handling of InferenceCore.step (inference_core.py:259-300): merge -> aggregate -> softmax.
    dyn in: inmask (i32 [h0,w0] | f32 [n,h0,w0]), pred f32 [Kold+1,H,W] (or 0), src i32 [Knew].  dyn out: prob."""
    P = Plan(eng)
    planes = P.buf('planes', (Knew, H, W), F32)
    P.ol.mask_merge(Dyn('inmask'), Dyn('pred'), Dyn('src'), planes, h0=h0, w0=w0, H=H, W=W, pad_left=pad_left,
                    pad_top=pad_top, Knew=Knew, Kold=Kold, nfloat=nfloat, float_mode=float_mode)
    P.ol.agg_softmax(planes, Dyn('prob'), K=Knew, HW=H * W)
    return P
