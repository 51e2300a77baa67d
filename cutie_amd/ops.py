"""Launch-plan builder: Python mirror of the ``cutie_op`` descriptor of include/cutie_hip.h.

An ``OpList`` records kernel descriptors over torch-allocated device buffers (torch = memory + stream
plumbing only); ``run()`` hands the whole array to ``cutie_exec`` in ONE C call.  Pointer slots can be
declared *dynamic* (named) and patched per call, so a cached plan is reused across frames.
"""
import os

import numpy as np
import torch

from . import _lib

NI, NF, NP = 24, 6, 16
OP_DTYPE = np.dtype([('kind', '<i4'), ('flags', '<i4'), ('i', '<i4', (NI,)), ('f', '<f4', (NF,)), ('p', '<u8', (NP,))])
assert OP_DTYPE.itemsize == _lib.OP_STRUCT_SIZE

# op kinds (include/cutie_hip.h)
(CONV, MAXPOOL, IMG_PREP, UPSAMPLE2X_ADD, AREA_DOWN, MASK_DOWN, GAP, ECA_APPLY, GRU, SEG_AGG, UP4_SOFTMAX,
 MASK_MERGE, AGG_SOFTMAX, LINEAR, LAYERNORM, QUERY_INIT, AUX_MASK, ATTN_Q2P, ATTN_SELF, ATTN_P2Q, SUMMARIZE,
 ADD_PE, KEY_PREP, AFF_SCORE, AFF_SELECT, AFF_READOUT, MEMSET32, COPY2D, AXPY, USAGE_TICK, RANK_SELECT,
 GATHER_ROWS, CONSOL_AFF, CONSOL_READ, CAST, PROB_TO_ID, RESIZE, FLIP_W, AREA_DOWN3, QFFN, STEM, BANK_WRITE) = range(1, 43)

KIND_NAMES = {}
for _n in ('CONV MAXPOOL IMG_PREP UPSAMPLE2X_ADD AREA_DOWN MASK_DOWN GAP ECA_APPLY GRU SEG_AGG UP4_SOFTMAX MASK_MERGE '
           'AGG_SOFTMAX LINEAR LAYERNORM QUERY_INIT AUX_MASK ATTN_Q2P ATTN_SELF ATTN_P2Q SUMMARIZE ADD_PE KEY_PREP '
           'AFF_SCORE AFF_SELECT AFF_READOUT MEMSET32 COPY2D AXPY USAGE_TICK RANK_SELECT GATHER_ROWS CONSOL_AFF '
           'CONSOL_READ CAST PROB_TO_ID RESIZE FLIP_W AREA_DOWN3 QFFN STEM').split():
    KIND_NAMES[globals()[_n]] = _n

F_RELU_IN, F_OUT_F32, F_RES_BCAST = 1, 2, 4
# launches of the frame's critical path raise their waves' issue priority (include/cutie_hip.h CUTIE_F_PRIO; A/B switch)
PRIO = os.environ.get('CUTIE_AMD_PRIO', '1') not in ('', '0')
F_PRIO, F_AFF_PRIO = 256, 64
F_PLAIN = 8 if os.environ.get('CUTIE_AMD_COUT1_ROWS', '1') in ('', '0') else 0      # A/B switch of conv_cout1_rows_kernel
F_TILE_OFF = 128 if os.environ.get('CUTIE_AMD_COUT1_TILE', '1') in ('', '0') else 0  # A/B switch of conv_cout1_tile_kernel
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SQ1 = 0, 1, 2, 3
ACT_SHIFT = 4
UP4_SCALAR = 2 if os.environ.get('CUTIE_AMD_UP4_VEC', '1') in ('', '0') else 0       # UP4_SOFTMAX flags&2: one pixel per thread (A/B switch)
AREA_RT = 8 if os.environ.get('CUTIE_AMD_AREA_R', '1') in ('', '0') else 0          # AREA_DOWN3 flags&8: bodies with a run-time pooling ratio (A/B switch)
GRU_SCALAR = 1 if os.environ.get('CUTIE_AMD_GRU4', '1') in ('', '0') else 0            # GRU flags&1: one channel per thread (A/B switch)
KEYPREP_LOOP = 2 if os.environ.get('CUTIE_AMD_KEYPREP_LOOP', '0') not in ('', '0') else 0   # KEY_PREP flags&2: c_j by one lane per row (A/B switch)
UP4_LANES = 16 if os.environ.get('CUTIE_AMD_UP4_SHARED', '1') in ('', '0') else 0     # UP4_SOFTMAX flags&16 (mask-down form): every lane aggregates its own six source pixels (A/B switch)
SELECT_COARSE = 2 if os.environ.get('CUTIE_AMD_SELECT_FINE', '1') in ('', '0') else 0    # AFF_SELECT flags&2: 16 | 32 | 64 values per lane only (A/B switch)
UP4_RTK = 8 if os.environ.get('CUTIE_AMD_UP4_KC', '1') in ('', '0') else 0           # UP4_SOFTMAX flags&8: kernels with a run-time object count (A/B switch)

# conv tile table (mirrors the switch in csrc/conv_igemm.hip): id -> (BM, BN, BK)
TILES = {0: (128, 128, 32), 1: (128, 64, 32), 2: (64, 64, 32), 3: (256, 16, 32), 4: (64, 128, 32),
         5: (64, 64, 64), 6: (64, 128, 64), 7: (128, 128, 64), 8: (64, 64, 128), 9: (32, 64, 64),
         10: (128, 64, 64), 11: (32, 64, 128), 12: (32, 128, 64),
         13: (64, 64, 64), 14: (128, 64, 64), 15: (64, 128, 64), 16: (128, 128, 64), 17: (64, 64, 128), 18: (128, 128, 32),   # 13+: 8 waves
         # 20+: WK K groups per block (intra-block split-K: WK copies of the 4-wave pipeline on one output tile)
         20: (64, 64, 64), 21: (64, 64, 128), 22: (32, 64, 128), 23: (128, 64, 64), 24: (64, 128, 64), 25: (128, 128, 64),
         26: (32, 64, 64), 27: (64, 64, 64), 28: (32, 64, 64), 29: (64, 128, 32), 30: (64, 64, 32)}
TILE_WK = {20: 2, 21: 2, 22: 2, 23: 2, 24: 2, 25: 2, 26: 2, 27: 4, 28: 4, 29: 2, 30: 2}
# (Round 3 removed three kernel families that no geometry of the cold-cache sweeps selects any more -- profiles/r03_conv_sweep_cold.md:
# 40..44 patch-resident 3x3 in the register-staged kernel, 50..56 buffer-load kernel, 90..95 strip-resident 3x3.  The halo tiles of
# conv_pc.hip, 120.., are what became of the resident-input idea.)

# 60+: LDS-DMA kernel (conv_dma.hip): operands global -> LDS by buffer_load ... lds, rolled K loop with ~2-4 non-MFMA
# instructions per MFMA.  id -> (BM, BN, BK); needs Cin % 64 == 0 (both sources of a virtual concat), no split-K.
DMA_TILES = {60: (128, 128, 64), 61: (128, 128, 64), 62: (128, 128, 64), 63: (128, 64, 64), 64: (64, 128, 64), 65: (64, 64, 64),
             66: (64, 64, 64), 67: (32, 64, 64), 68: (256, 128, 64), 69: (32, 128, 64), 70: (128, 128, 64), 71: (128, 64, 64),
             72: (64, 64, 64), 73: (64, 64, 64), 74: (64, 64, 64), 75: (32, 64, 64), 76: (128, 64, 64), 77: (64, 128, 64), 78: (32, 128, 64),
             80: (64, 64, 128), 81: (64, 64, 128), 82: (32, 64, 128), 83: (32, 64, 128), 84: (64, 128, 128), 85: (128, 64, 128),
             86: (96, 64, 64), 87: (96, 64, 64), 88: (64, 64, 64), 89: (64, 64, 64)}


# 100+: conv_pc.hip -- producer / consumer split (NPW producer waves issue every LDS-DMA piece, WM x WN consumer waves only read
# fragments and issue MFMAs), epilogue straight from the accumulators.  100..: stream mode (any kernel size / stride);
# 120..: halo mode, 3x3 / stride 1 / pad 1 with the (TH+2) x (TW+2) input patch of a 64-channel slice resident in LDS.
# id -> (BM, BN, BK); PC_HALO: id -> (TH, TW)
PC_TILES = {100: (64, 64, 64), 101: (64, 64, 64), 102: (32, 64, 64), 103: (128, 64, 64), 104: (64, 128, 64), 105: (128, 128, 64),
            106: (128, 128, 64), 107: (64, 64, 64), 108: (128, 128, 64), 109: (32, 64, 64), 110: (96, 64, 64), 111: (96, 128, 64),
            120: (64, 64, 64), 121: (64, 64, 64), 122: (128, 64, 64), 123: (128, 128, 64), 124: (64, 128, 64), 125: (32, 64, 64),
            126: (128, 128, 64), 127: (64, 128, 64), 129: (96, 64, 64), 130: (32, 64, 64), 131: (64, 128, 64),
            132: (64, 64, 64), 133: (96, 128, 64), 134: (320, 64, 64),
            # pair steps (two K tiles per barrier): UNMEASURED, branch next/pc-pairstep
            140: (64, 64, 64), 141: (96, 64, 64), 142: (96, 64, 64), 143: (32, 64, 64), 144: (128, 64, 64),
            145: (96, 64, 64), 146: (128, 64, 64), 147: (32, 64, 64)}
PC_HALO = {120: (8, 8), 121: (4, 16), 122: (8, 16), 123: (8, 16), 124: (8, 8), 125: (4, 8), 126: (8, 16), 127: (4, 16),
           129: (10, 9), 130: (5, 6), 131: (6, 9), 132: (6, 9), 133: (10, 9), 134: (20, 16),
           145: (10, 9), 146: (8, 16), 147: (5, 6)}

ALL_TILES = {**TILES, **DMA_TILES, **PC_TILES}


def korder_class(tile, splitk=1):
    """Which fp32 summation order over K a conv tile implements -- two tiles of one class give bit-identical outputs on the same
    operands (tests/test_gpu_kernels.py::test_tiles_of_a_korder_class_agree_bitwise).  'stream': K tiles in the packed order
    k = (kh * KW + kw) * Cin + c, two v_mfma_f32_16x16x32_bf16 per 64 channels in ascending k (conv_dma.hip, the stream tiles of
    conv_pc.hip); 'igemm': the register-staged tiles without split-K (same order, not asserted against 'stream'); 'halo': 64-channel slice outermost, its nine taps inside
    (conv_pc.hip, tiles 120..); anything that splits K (WK tiles, split-K) is a class of its own."""
    if tile in PC_HALO:
        return 'halo'
    if tile in TILE_WK or splitk != 1:
        return ('split', tile, splitk)
    if tile in (3, COUT1_TILE):
        return ('small', tile)
    return 'stream' if (tile in DMA_TILES or tile in PC_TILES) else 'igemm'


def pc_tile_ok(tile, *, cin, kh, stride=1, pad=None, c2=0):
    """conv_pc_kernel eligibility (mirrors launch_pc in conv_pc.hip)."""
    if cin % 64 or (c2 and (c2 % 64 or (cin - c2) % 64)) or kh > 5:
        return False
    if tile in PC_HALO:
        return kh == 3 and stride == 1 and (pad is None or pad == 1)
    return True


def pc_blocks(tile, B, OH, OW, cout):
    """Workgroups of a conv on a conv_pc tile."""
    bm, bn, _ = PC_TILES[tile]
    if tile in PC_HALO:
        th, tw = PC_HALO[tile]
        return B * -(-OH // th) * -(-OW // tw) * -(-cout // bn)
    return -(-(B * OH * OW) // bm) * -(-cout // bn)


def dma_tiles_enabled():
    return os.environ.get('CUTIE_AMD_DMA_TILES', '1') not in ('', '0')


def dma_tile_ok(tile, *, cin, kh, c2=0):
    """conv_dma_kernel eligibility (mirrors launch_dma in conv_dma.hip): a K tile (64 or 128 channels) never straddles a tap or a
    source."""
    bk = DMA_TILES[tile][2]
    return cin % bk == 0 and (c2 == 0 or (c2 % bk == 0 and (cin - c2) % bk == 0)) and kh * kh <= 32


def conv_side_jobs_ok(*, cin, cout, kh, c2=0):
    """Can a conv of this geometry carry the GAP accumulation / zero job (conv_dma_kernel only)?"""
    return dma_tiles_enabled() and dma_tile_ok(60, cin=cin, kh=kh, c2=c2) and cout % 8 == 0 and cout > 16


NUM_CU = 256
# bytes of the next conv's weights a conv_pc launch touches on its way out (0: off); CUTIE_AMD_WPF overrides
WEIGHT_PREFETCH = int(os.environ.get('CUTIE_AMD_WPF', str(8 << 20)))
WEIGHT_PREFETCH_2 = int(os.environ.get('CUTIE_AMD_WPF2', '1'))
WEIGHT_PREFETCH_BLOCK = int(os.environ.get('CUTIE_AMD_WPF_BLOCK', str(48 << 10)))      # bytes per block (see OpList.finalize)


COUT1_TILE = 19          # dedicated per-pixel dot-product kernel (conv_cout1_kernel)


def cout1_ok(cout, cin, c2=0, res=False):
    lp = cin // 8
    return cout == 1 and c2 == 0 and not res and cin % 8 == 0 and 1 <= lp <= 64 and (lp & (lp - 1)) == 0


def tile_candidates(M, cout, cin, kpad=None, geom=None):
    """Tile ids that are legal for a conv (BK > 32 needs Cin >= 32; tiny Cout uses the 256x16 tile; the K groups of a
    WK tile must divide the K tiles; geom = dict(kh, stride, pad, W, c2) enables the LDS-DMA families)."""
    if cout <= 16:
        return [3] + ([COUT1_TILE] if cout1_ok(cout, cin) else [])
    out = []
    for t, (bm, bn, bk) in TILES.items():
        if t == 3 or (bk > 32 and cin < 32):
            continue
        wk = TILE_WK.get(t, 1)
        if wk > 1 and (kpad is None or (kpad // bk) % wk or kpad // bk < 2 * wk):
            continue
        if bn == 128 and cout <= 64:
            continue
        if bm * 2 > max(M, 64) * 2 and bm > 64:          # do not pad a tiny M to a huge tile
            continue
        out.append(t)
    if dma_tiles_enabled() and geom is not None and dma_tile_ok(60, cin=cin, kh=geom['kh'], c2=geom.get('c2', 0)):
        for t, (bm, bn, bk) in DMA_TILES.items():
            if not (bn == 128 and cout <= 64) and not (bm > 64 and bm > max(M, 64)) and not (bm == 256 and M < 16384) \
                    and dma_tile_ok(t, cin=cin, kh=geom['kh'], c2=geom.get('c2', 0)):
                out.append(t)
    if dma_tiles_enabled() and geom is not None:
        for t, (bm, bn, bk) in PC_TILES.items():
            if pc_tile_ok(t, cin=cin, kh=geom['kh'], stride=geom.get('stride', 1), pad=geom.get('pad'), c2=geom.get('c2', 0)) \
                    and not (bn == 128 and cout <= 64) and not (bm > 64 and bm > max(M, 64)) and not (bm == 256 and M < 16384):
                out.append(t)
    return out


SPLITK_PART_FLOATS = 16 << 20           # 64 MiB of fp32 partial tiles
_splitk_scratch = {}


def splitk_scratch(device, owner=None):
    """fp32 partial-tile scratch of split-K convs.  One buffer per owner (an Engine: the launches of one engine are
    stream-ordered, engines forked for concurrent clips must not share partials; it lives and dies with the owner);
    owner None = one per device."""
    dev = torch.device(device)
    store = _splitk_scratch if owner is None else owner.__dict__.setdefault('_splitk_part', {})
    key = str(dev)
    if key not in store:
        n = SPLITK_PART_FLOATS if dev.type == 'cuda' else 1024
        store[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return store[key]


def splitk_candidates(M, cout, kpad, tile):
    """Split-K factors worth timing for a conv on a given tile: only when the plain grid leaves CUs idle."""
    if tile == COUT1_TILE or tile == 3 or tile in DMA_TILES or tile in PC_TILES:
        return [1]
    bm, bn, bk = TILES[tile]
    blocks = -(-M // bm) * -(-cout // bn)
    nk = kpad // bk // TILE_WK.get(tile, 1)
    ldp = (cout + 7) & ~7
    out = [1]
    for sk in (2, 3, 4, 6, 8, 9, 12, 16, 18):
        if blocks >= NUM_CU or blocks * sk > 6 * NUM_CU or nk // sk < 2:
            break
        if sk * M * ldp > SPLITK_PART_FLOATS:
            break
        if nk % sk == 0:                                 # equal slices only (conv_igemm.hip)
            out.append(sk)
    return out


def pick_tile(M, cout, cin=64, geom=None):
    """Static, deterministic tile choice for conv geometries that are not in the tuned table (cutie_amd/tiles_gfx950.json):
    the largest tile that still yields >= NUM_CU workgroups, else the one with the most workgroups -- among the LDS-DMA tiles
    when the conv is eligible for them (geom = dict(kh, c2) given and Cin % 64 == 0), else among the register-staged ones."""
    if cout <= 16:
        return 3
    if geom is not None and dma_tiles_enabled() and dma_tile_ok(60, cin=cin, kh=geom['kh'], c2=geom.get('c2', 0)):
        # order fitted on the cold-cache sweeps of the 480p (K = 1, 2, 3) and 1080p (K = 5) geometries (producer / consumer tiles)
        cands = [t for t in (108, 103, 110, 101, 102) if pc_tile_ok(t, cin=cin, kh=geom['kh'], c2=geom.get('c2', 0))]
    else:
        cands = [6, 5, 9] if cin >= 32 else [4, 2]
    if cout <= 64:
        cands = [c for c in cands if ALL_TILES[c][1] <= 64] or [2]
    best, best_blocks = None, -1
    for t in cands:
        bm, bn, _ = ALL_TILES[t]
        blocks = -(-M // bm) * -(-cout // bn)
        if blocks >= NUM_CU:
            return t
        if blocks > best_blocks:
            best, best_blocks = t, blocks
    return best


class Dyn:
    """A named, per-call patched pointer (optionally with a byte offset)."""
    __slots__ = ('name', 'offset')

    def __init__(self, name, offset=0):
        self.name, self.offset = name, offset


def _ptr(t):
    if t is None:
        return 0
    if isinstance(t, int):
        return t
    return t.data_ptr()


class OpList:
    # kinds whose kernels raise their waves' issue priority under F_PRIO (include/cutie_hip.h); CONV and the affinity ops carry a `prio` argument
    PRIO_KINDS = frozenset((UPSAMPLE2X_ADD, AREA_DOWN3, ECA_APPLY, GRU, UP4_SOFTMAX, ATTN_Q2P, ATTN_SELF, ATTN_P2Q, QFFN))

    def __init__(self, scratch_owner=None, touch_next_weights=True, prio=True):
        self.prio = prio                     # the launches of this list belong to the frame's critical path (plans.Plan.prio; $CUTIE_AMD_PRIO=0: nobody's do)
        self.scratch_owner = scratch_owner   # see splitk_scratch
        self.touch_next_weights = touch_next_weights   # see finalize (plans switch it off where weights stay warm between two uses)
        self.recs = []          # (kind, flags, ints, floats, ptrs)
        self.keep = []          # tensors kept alive
        self.wbytes = {}        # conv op index -> bytes of its packed weights
        self.dyn = {}           # name -> [(op index, slot, offset)]
        self.arr = None
        self._words = None      # bind's view of `dyn` and `arr` (rebuilt with the array)

    # ---- generic ------------------------------------------------------------------
    def add(self, kind, flags=0, ints=(), floats=(), ptrs=()):
        idx = len(self.recs)
        if kind in self.PRIO_KINDS and self.prio and PRIO:
            flags |= F_PRIO
        pl = []
        for slot, t in enumerate(ptrs):
            if isinstance(t, Dyn):
                self.dyn.setdefault(t.name, []).append((idx, slot, t.offset))
                pl.append(0)
            else:
                if isinstance(t, torch.Tensor):
                    self.keep.append(t)
                pl.append(_ptr(t))
        self.recs.append((kind, flags, [int(v) for v in ints], [float(v) for v in floats], pl))
        self.arr = self._words = None
        return idx

    def finalize(self):
        arr = np.zeros(len(self.recs), dtype=OP_DTYPE)
        for n, (kind, flags, ints, floats, ptrs) in enumerate(self.recs):
            arr['kind'][n] = kind
            arr['flags'][n] = flags
            arr['i'][n, :len(ints)] = ints
            arr['f'][n, :len(floats)] = floats
            arr['p'][n, :len(ptrs)] = ptrs
        self.arr = arr
        self._words = None
        self.wire_next_weights()
        return arr

    def wire_next_weights(self):
        """CONV p9 / i22, p10 / i23 of the finalized array, from the tiles it carries NOW (plans call this again once the tile table
        has been applied): a producer / consumer conv touches the packed weights of the conv behind it in the list when its own DMA
        is out, and those of the conv after that one when the next conv has no producer waves (conv_pc.hip)."""
        arr = self.arr
        convs = [n for n in range(len(arr)) if arr['kind'][n] == CONV]
        for n in convs:
            arr['p'][n, 9:11] = 0
            arr['i'][n, 22:24] = 0
        if not (WEIGHT_PREFETCH and self.touch_next_weights):
            return
        nxt, nxt2 = None, None                               # (weights, bytes, has producer waves) of the next conv / of the one after it
        for n in reversed(convs):
            i = arr['i'][n]
            pc = int(i[17]) in PC_TILES
            if nxt is not None and pc:
                # the blocks of one XCD share the range: a block that pulls much more than ~48 KB through its CU outlasts the
                # consumers' epilogue (bytes per CU), so small grids touch only the head of large weights
                budget = WEIGHT_PREFETCH_BLOCK * -(-pc_blocks(int(i[17]), int(i[0]), int(i[7]), int(i[8]), int(i[9])) // 8)
                n1 = min(nxt[1], WEIGHT_PREFETCH, budget)
                arr['p'][n, 9], arr['i'][n, 22] = nxt[0], n1
                n2 = min(nxt2[1], WEIGHT_PREFETCH, budget - n1) if (nxt2 is not None and not nxt[2] and WEIGHT_PREFETCH_2) else 0
                if n2 > 0:                                   # the next conv cannot do it for its successor
                    arr['p'][n, 10], arr['i'][n, 23] = nxt2[0], n2
            nxt, nxt2 = (int(arr['p'][n, 2]), self.wbytes[n], pc), nxt

    def patch_ints(self, op, start, values):
        """Overwrite ints [start, start + len(values)) of record `op` -- in the recorded list and, when the array exists, in place (a
        plan whose launches stay the same while a few sizes change: the affinity plans of a bucket between two memory frames)."""
        values = [int(v) for v in values]
        ints = self.recs[op][2]
        end = start + len(values)
        if len(ints) < end:
            ints.extend([0] * (end - len(ints)))
        ints[start:end] = values
        if self.arr is not None:
            self.arr['i'][op, start:end] = values

    def bind(self, **tensors):
        """Patch dynamic pointer slots.  Values: torch tensors or raw ints.  (Host time: this runs five times per frame with a dozen
        names each -- the slots of a name are 64-bit word indices into the descriptor array, written through one memoryview.)"""
        if self.arr is None:
            self.finalize()
        words = self._words
        if words is None:
            # a descriptor is 32 words; its pointer field starts at word 16 (include/cutie_hip.h, asserted against the dtype)
            assert OP_DTYPE.itemsize == 256 and OP_DTYPE.fields['p'][1] == 128
            words = self._words = ({name: tuple((idx * 32 + 16 + slot, off) for (idx, slot, off) in sl) for name, sl in self.dyn.items()},
                                   memoryview(self.arr).cast('B').cast('Q'))
        table, mv = words
        for name, t in tensors.items():
            sl = table.get(name)
            if sl is None:
                continue
            base = 0 if t is None else (t if type(t) is int else t.data_ptr())
            if base:
                for w, off in sl:
                    mv[w] = base + off
            else:
                for w, off in sl:
                    mv[w] = 0

    def run(self, _stream=None, **tensors):
        """_stream: a torch stream other than the current one to launch on (the executor is handed its raw handle; an executor
        without `run_on` -- the recording shims, the interpreter of the tests -- runs inside torch's stream context instead)."""
        if self.arr is None:
            self.finalize()
        if tensors:
            self.bind(**tensors)
        ex = _lib.get_executor()
        if _stream is None:
            ex.run(self.arr)
        elif hasattr(ex, 'run_on'):
            ex.run_on(self.arr, _stream)
        else:
            with torch.cuda.stream(_stream):
                ex.run(self.arr)

    def __len__(self):
        return len(self.recs)

    # ---- builders (argument order mirrors include/cutie_hip.h) -----------------------
    def conv(self, x1, w, y, *, B, H, W, C1, ldx1, OH, OW, ldy, stride=1, pad=0, x2=None, C2=0, ldx2=0,
             res=None, ldr=0, res_bcast=False, relu_in=False, act=ACT_NONE, out_f32=False, tile=None, splitk=1,
             gap_acc=None, zero=None, prio=False, res_group=None):
        """w: PackedConv (weights.py).  res_group = (objects per clip, rows between the clips' residual maps) with res_bcast: clips in lock step
        (include/cutie_hip.h CONV, ABI 4) -- the B objects come in groups, every group adds its own broadcast residual.  gap_acc: int64 [B, Cout] -- the conv adds the per-(object, channel) sums of its stored output
        (fixed point x 2^24) to it (ECA's global average pool without a launch of its own); zero: an int64 tensor cleared by this
        launch (the accumulator of the NEXT conv).  Both need an LDS-DMA tile (the tile choice is restricted accordingly)."""
        flags = (F_RELU_IN if relu_in else 0) | (F_OUT_F32 if out_f32 else 0) | (F_RES_BCAST if res_bcast else 0) | (act << ACT_SHIFT) | F_PLAIN | F_TILE_OFF | \
            (F_PRIO if (prio and PRIO) else 0)
        assert C1 + C2 == w.cin_padded, (C1, C2, w.cin_padded)
        M = B * OH * OW
        side = gap_acc is not None or zero is not None
        if tile is None:
            tile = COUT1_TILE if (cout1_ok(w.cout, C1 + C2, C2, res is not None) and not side) else pick_tile(M, w.cout, C1 + C2, dict(kh=w.kh, c2=C2))
        if side:
            assert (tile in DMA_TILES or tile in PC_TILES) and not out_f32 and w.cout % 8 == 0 and ldy % 8 == 0, 'GAP accumulation needs an LDS-DMA conv (see conv_side_jobs_ok)'
        part = splitk_scratch(w.weight.device, self.scratch_owner)
        self.wbytes[len(self.recs)] = w.weight.numel() * w.weight.element_size()
        return self.add(CONV, flags,
                        [B, H, W, C1, C2, ldx1, ldx2, OH, OW, w.cout, ldy, w.kh, w.kw, stride, pad, ldr, w.kpad, tile, w.cin_real,
                         splitk, part.numel() // 1024, 0 if zero is None else zero.numel()],
                        self._res_group(res_group, res_bcast, B, OH * OW), [x1, x2, w.weight, w.bias, res, y, part, gap_acc, zero])

    @staticmethod
    def _res_group(res_group, res_bcast, B, OHW):
        if res_group is None:
            return []
        kg, stride = int(res_group[0]), int(res_group[1])
        assert res_bcast and kg > 0 and B % kg == 0 and stride >= OHW and stride < (1 << 24), (res_group, B, OHW)
        return [float(kg), float(stride)]

    def maxpool(self, x, y, *, B, H, W, C, relu=False):
        OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        return self.add(MAXPOOL, 1 if relu else 0, [B, H, W, C, OH, OW], [], [x, y])

    STEM_MAX_IMAGES = 12

    def stem(self, image, masks, w, y, *, h0, w0, H, W, pad_left, pad_top, K, mean, std, relu=True, more_images=(), mask_stride=0):
        """IMG_PREP + 7x7 / stride-2 conv (w: PackedConv, Cin padded to 8, Cout 64) + 3x3 / stride-2 max pool (+ ReLU) in one launch.
        more_images (ABI 4, <= 11): further frames of the SAME launch -- frame f reads more_images[f - 1], the masks `mask_stride` floats
        behind frame f - 1's, and writes behind frame f - 1's output (y holds all frames)."""
        assert w.cout == 64 and w.kh == 7 and w.cin_padded == 8 and H % 16 == 0 and W % 16 == 0
        more = list(more_images)
        assert len(more) < self.STEM_MAX_IMAGES
        ints = [h0, w0, H, W, pad_left, pad_top, K, w.kpad] + ([1 + len(more), int(mask_stride)] if more else [])
        return self.add(STEM, 1 if relu else 0, ints, list(mean) + list(std), [image, masks, w.weight, w.bias, y] + more)

    def img_prep(self, image, masks, y, *, h0, w0, H, W, pad_left, pad_top, K, mean, std):
        return self.add(IMG_PREP, 0, [h0, w0, H, W, pad_left, pad_top, K], list(mean) + list(std), [image, masks, y])

    def upsample2x_add(self, g, skip, y, *, B, h, w, C, skip_group=None):
        """skip_group = (objects per clip, pixels between the clips' skip maps): clips in lock step (ABI 4)."""
        ints = [B, h, w, C]
        if skip_group is not None:
            kg, stride = int(skip_group[0]), int(skip_group[1])
            assert kg > 0 and B % kg == 0 and stride >= 4 * h * w
            ints += [kg, stride]
        return self.add(UPSAMPLE2X_ADD, 0, ints, [], [g, skip, y])

    def area_down(self, x, y, *, B, H, W, C, ldx, ldy, r, f32_in=False, Cz=None):
        return self.add(AREA_DOWN, 1 if f32_in else 0, [B, H, W, C, ldx, ldy, r, C if Cz is None else Cz], [], [x, y])

    def area_down3(self, segs):
        """Three area poolings in one launch; segs = 3 x dict(x, y, B, H, W, C, ldx, ldy, r, f32_in=False, Cz=None)."""
        assert len(segs) == 3
        ints, ptrs, flags = [], [], 0
        for q, g in enumerate(segs):
            ints += [g['B'], g['H'], g['W'], g['C'], g['ldx'], g['ldy'], g['r'], g['C'] if g.get('Cz') is None else g['Cz']]
            ptrs += [g['x'], g['y']]
            flags |= (1 << q) if g.get('f32_in') else 0
        return self.add(AREA_DOWN3, flags | AREA_RT, ints, [], ptrs)

    def mask_down(self, masks, pair, m16, *, K, H, W, r=16, pair_channels=8):
        """pair_channels: channel pitch of `pair` (8, or 64 when it is the second source of an LDS-DMA conv: channels 8.. are not written)."""
        assert pair_channels % 8 == 0
        return self.add(MASK_DOWN, 0, [K, H, W, r, pair_channels], [], [masks, pair, m16])

    def gap(self, x, y, *, B, HW, C, scratch=None, partial_only=False):
        if scratch is None:
            scratch = torch.zeros((B, -(-HW // 64), C), dtype=torch.float32, device=y.device)
        self._last_gap_scratch = scratch
        return self.add(GAP, 1 if partial_only else 0, [B, HW, C], [], [x, y, scratch])

    def eca_apply(self, x, gap, wk, r, y, *, B, HW, C, part=None, fixed_sums=None, head=None):
        """part: the GAP partials of the preceding gap(..., partial_only=True) (defaults to the last gap scratch);
        fixed_sums: int64 [B, C] sums accumulated by the producing conv (conv(gap_acc=...)) instead.
        head = (PackedConv w of a Cout = 1, 1x1 conv, out f32 [B, HW]): relu(y) . w + bias computed by the same launch (C = 256)."""
        tail = []
        if head is not None:
            w, out = head
            assert C == 256 and w.cout == 1 and w.kh == 1 and w.cin_padded == 256
            tail = [w.weight, w.bias, out]
        if fixed_sums is not None:
            return self.add(ECA_APPLY, 1, [B, HW, C], [], [x, gap, wk, r, y, fixed_sums] + tail)
        if part is None:
            part = self._last_gap_scratch
        return self.add(ECA_APPLY, 0, [B, HW, C], [], [x, gap, wk, r, y, part] + tail)

    def gru(self, values, h, hb, *, n, C):
        return self.add(GRU, GRU_SCALAR, [n, C], [], [values, h, hb])

    def seg_agg(self, logits, agg, *, K, hw):
        return self.add(SEG_AGG, 0, [K, hw], [], [logits, agg])

    def up4_softmax(self, agg, prob, logits_up, *, P, h, w, from_logits=False, mask_down=None, clips=1):
        """from_logits: `agg` holds the K = P - 1 raw logit planes and the aggregation (SEG_AGG) runs inside the launch (P <= 16).
        mask_down = (m16 f32 [K, hw16], pair bf16 [K, h16, w16, pitch], pitch): the launch also writes MASK_DOWN(prob[1:], r = 16).
        clips > 1 (clips in lock step, ABI 4; the four-pixel forms): every array holds that many clips, P planes out / P - 1 in each."""
        if mask_down is not None:
            assert from_logits and not UP4_SCALAR and P <= 8 and h % 4 == 0 and w % 4 == 0
            m16, pair, pitch = mask_down
            return self.add(UP4_SOFTMAX, 1 | 4 | UP4_RTK | UP4_LANES, [P, h, w, pitch, clips], [], [agg, prob, logits_up, m16, pair])
        assert clips == 1 or (from_logits and not UP4_SCALAR and P <= 8)
        return self.add(UP4_SOFTMAX, (1 | UP4_SCALAR | UP4_RTK) if from_logits else 0, [P, h, w, 0, clips], [], [agg, prob, logits_up])

    def mask_merge(self, inmask, pred, src, planes, *, h0, w0, H, W, pad_left, pad_top, Knew, Kold, nfloat, float_mode):
        return self.add(MASK_MERGE, 1 if float_mode else 0, [h0, w0, H, W, pad_left, pad_top, Knew, Kold, nfloat], [],
                        [inmask, pred, src, planes])

    def agg_softmax(self, planes, prob, *, K, HW):
        return self.add(AGG_SOFTMAX, 0, [K, HW], [], [planes, prob])

    def linear(self, x, w, y, *, M, ldx=None, ldy=None, x_add=None, add_rows=0, res=None, relu=False, add_cols=0,
               ln=None, ln_out=None, eps=1e-5):
        """w: PackedLinear.  ln = (gamma, beta): nn.LayerNorm fused in front (ln_out: where to keep the normalised rows);
        add_cols: x_add feeds only the first add_cols output columns."""
        flags = (1 if relu else 0) | (2 if ln is not None else 0)
        return self.add(LINEAR, flags,
                        [M, w.n, w.kd, w.kd if ldx is None else ldx, w.n if ldy is None else ldy, add_rows, add_cols], [eps],
                        [x, x_add, w.weight, w.bias, res, y, ln[0] if ln else None, ln[1] if ln else None, ln_out])

    def layernorm(self, x, g, b, y, *, M, C):
        return self.add(LAYERNORM, 0, [M, C], [], [x, g, b, y])

    def query_init(self, obj_mem, y, *, rows, C):
        return self.add(QUERY_INIT, 0, [rows, C], [], [obj_mem, y])

    def query_init2(self, obj_mem, query, query_emb, *, rows, w_init, res_init, w_emb, res_emb, zero=None):
        """QUERY_INIT and the two linears that consume it in one launch (C = 256, 16 summaries per object).
        zero = tensor: cleared as a side job (the fixed-point accumulators of the transformer blocks behind; size % 16 bytes == 0)."""
        assert w_init.kd == 256 and w_init.n == 256 and w_emb.kd == 256 and w_emb.n == 256 and rows % 16 == 0
        nz = 0
        if zero is not None:
            nb = zero.numel() * zero.element_size()
            assert nb % 16 == 0 and zero.is_contiguous()
            nz = nb // 16
        return self.add(QUERY_INIT, 1, [rows, 256, nz], [], [obj_mem, query, query_emb, w_init.weight, w_init.bias, res_init,
                                                             w_emb.weight, w_emb.bias, res_emb, zero])

    def aux_mask(self, logits, fg, nfg, *, K, HW):
        self.memset32(nfg, K, 0)
        return self.add(AUX_MASK, 0, [K, HW], [], [logits, fg, nfg])

    @staticmethod
    def _proj(proj):
        """Fused-projection operands of the attention ops: proj = dict(x, W (PackedLinear), emb=None, ln=(gamma, beta) | None,
        ln_out=None, ldx=256) -> (ldx, [ln_out], [W.weight, W.bias, emb, gamma, beta])."""
        ln = proj.get('ln') or (None, None)
        assert proj['W'].kd == 256
        return proj.get('ldx', 256), proj.get('ln_out'), [proj['W'].weight, proj['W'].bias, proj.get('emb'), ln[0], ln[1]]

    QACC_SCALE = 4294967296.0    # fixed-point scale of the accumulators of the query chain (csrc/qchain.hip)

    @staticmethod
    def _proj_extras(flags, ints, ptrs, acc_in, out_proj):
        """acc_in = (acc int64 [K*Q, 256], bias | None): the rows are x + bias + acc / 2^32 (flags&4: p10, p11) -- acc holds the products
        of the launch in front, summed over its blocks in fixed point.
        out_proj = (PackedLinear Wo [256,256], acc int64 [K*Q, 256]): the output projection runs inside the launch, per head, and is
        ADDED to acc (flags&8: p12, p13; cleared beforehand, e.g. by QUERY_INIT); its bias is left to the consumer (acc_in=(acc, Wo.bias))."""
        ints = list(ints) + [0] * (9 - len(ints))
        ptrs = list(ptrs) + [None] * (14 - len(ptrs))
        if acc_in is not None:
            assert acc_in[0].dtype == torch.int64
            flags |= 4
            ptrs[10], ptrs[11] = acc_in[0], acc_in[1]
        if out_proj is not None:
            assert out_proj[0].kd == 256 and out_proj[0].n == 256 and out_proj[1].dtype == torch.int64
            flags |= 8
            ptrs[12], ptrs[13] = out_proj[0].weight, out_proj[1]
        return flags, ints, ptrs

    def attn_q2p(self, q, kv, fg, nfg, y, *, K, Q, HW, C, heads, ldkv, voff, logits=None, proj=None, acc_in=None, out_proj=None, q_pre=None, hstride=None,
                 clip_objects=None):
        """logits given: the foreground mask is derived inside the kernel from the mask_pred logits (AUX_MASK fused; fg / nfg unused).
        proj given (needs logits): q = (LN(x) + emb) Wq^T + b is computed inside the launch from the unprojected rows proj['x'].
        out_proj (needs proj; chain form, see _proj_extras): y is not written; acc_in optional.
        q_pre (chain form, instead of proj): q f32 [K*Q, 256], already projected and scaled by 1/sqrt(32) -- by the ATTN_P2Q launch of the
        previous transformer block (attn_p2q(next_q=...)).
        clip_objects (chain form, ABI 4): objects per clip when the K objects are those of several clips in lock step (i9)."""
        hs = 0 if hstride in (None, C // heads) else hstride   # chain form only: elements between the heads' k (and v) inside a pixel row (i8)
        kg = 0 if clip_objects in (None, K) else int(clip_objects)
        assert kg == 0 or (out_proj is not None and K % kg == 0), 'clips in lock step need the chain form'
        if q_pre is not None:
            assert proj is None and acc_in is None and out_proj is not None and logits is not None
            flags, ints, ptrs = self._proj_extras(3 | 16, [K, Q, HW, C, heads, ldkv, voff, 256], [q_pre, kv, logits, None, None], None, out_proj)
            ints[8] = hs
            return self.add(ATTN_Q2P, flags, ints + [kg], [], ptrs)
        if proj is not None:
            assert logits is not None
            ldx, ln_out, tail = self._proj(proj)
            assert out_proj is not None or acc_in is None, 'the chain form of ATTN_Q2P needs out_proj'
            flags, ints, ptrs = self._proj_extras(3, [K, Q, HW, C, heads, ldkv, voff, ldx], [proj['x'], kv, logits, ln_out, y] + tail, acc_in, out_proj)
            assert hs == 0 or out_proj is not None, 'a head stride needs the chain form'
            ints[8] = hs
            return self.add(ATTN_Q2P, flags, ints + [kg], [], ptrs)
        assert acc_in is None and out_proj is None
        if logits is not None:
            return self.add(ATTN_Q2P, 1, [K, Q, HW, C, heads, ldkv, voff], [], [q, kv, logits, None, y])
        return self.add(ATTN_Q2P, 0, [K, Q, HW, C, heads, ldkv, voff], [], [q, kv, fg, nfg, y])

    def attn_self(self, qk, v, y, *, K, Q, C, heads, ldqk=0, ldv=0, proj=None, acc_in=None, out_proj=None):
        """proj given: q | k | v = packed in-projection of (LN(x) + emb | LN(x) + emb | LN(x)) computed inside the launch.
        out_proj (needs proj; chain form): y is not written; acc_in optional."""
        if proj is not None:
            ldx, ln_out, tail = self._proj(proj)
            assert out_proj is not None or acc_in is None, 'the chain form of ATTN_SELF needs out_proj'
            flags, ints, ptrs = self._proj_extras(2, [K, Q, C, heads, 0, 0, ldx], [proj['x'], None, y, ln_out, None] + tail, acc_in, out_proj)
            return self.add(ATTN_SELF, flags, ints, [], ptrs)
        assert acc_in is None and out_proj is None
        return self.add(ATTN_SELF, 0, [K, Q, C, heads, ldqk, ldv], [], [qk, v, y])

    def attn_p2q(self, q, kq, vq, y, *, K, Q, HW, C, heads, ldq, ldkv=0, proj=None, acc_in=None, next_q=None, out=None):
        """proj given: k | v of the object queries = packed [k | v] projection of (x + emb | x) computed inside the launch.
        acc_in (needs proj): chain form.  next_q = dict(ln=(gamma, beta), W=PackedLinear Wq, q_out, xn_out) (chain form): extra blocks project
        the NEXT transformer block's ATTN_Q2P queries from the same rows: xn_out = LN(x_eff), q_out = ((xn_out + emb) Wq^T + b) / sqrt(32).
        out = dict(Wo=weights.out_proj_blob(...), res=bf16 [K, HW, 256]) (chain form): y = res + Wo . attention + bias -- the 1x1 conv behind
        the attention (read_from_query's out_proj, object_transformer.py:66-70) inside the launch; the attention itself is not stored."""
        if proj is not None:
            ldx, _, tail = self._proj(proj)
            flags, ints, ptrs = self._proj_extras(2, [K, Q, HW, C, heads, ldq, ldkv, ldx], [q, proj['x'], None, y, None] + tail[:3], acc_in, None)
            if out is not None:
                assert acc_in is not None and out['Wo'].dtype == torch.uint8 and out['Wo'].numel() == 256 * 256 * 2 + 256 * 4
                flags |= 32
                ptrs[2], ptrs[4] = out['Wo'], out['res']
            if next_q is not None:
                assert acc_in is not None and next_q['W'].kd == 256 and next_q['W'].n == 256
                flags |= 16
                ptrs = ptrs + [None] * (16 - len(ptrs))
                ptrs[8], ptrs[9] = next_q['ln']
                ptrs[12], ptrs[13], ptrs[14], ptrs[15] = next_q['W'].weight, next_q['W'].bias, next_q['q_out'], next_q['xn_out']
            return self.add(ATTN_P2Q, flags, ints, [], ptrs)
        assert acc_in is None
        return self.add(ATTN_P2Q, 0, [K, Q, HW, C, heads, ldq, ldkv], [], [q, kq, vq, y])

    def qffn(self, x, x_out, acc_out, *, rows, ln, W1, W2, acc_in=None, hid_slice=64):
        """FFN of a transformer block in one launch: acc_out += relu(LN(x_eff) W1^T + b1) W2^T in fixed point, summed over the FF / hid_slice
        blocks of an object; x_eff = x (+ acc_in) is written to x_out.  W2's bias is left to the consumer (acc_in=(acc_out, W2.bias))."""
        FF = W1.n
        assert W1.kd == 256 and W2.kd == FF and W2.n == 256 and FF % hid_slice == 0 and hid_slice in (64, 128) and rows % 16 == 0
        assert acc_out.dtype == torch.int64
        ints = [rows, FF, hid_slice]
        ptrs = [x, x_out, ln[0], ln[1], W1.weight, W1.bias, W2.weight, acc_out] + [None] * 6
        if acc_in is not None:
            assert acc_in[0].dtype == torch.int64
            ptrs[10], ptrs[11] = acc_in[0], acc_in[1]
        return self.add(QFFN, 0, ints, [], ptrs)

    def summarize(self, feat, wl, m16, y, *, K, HW, C, Q, scratch=None, feat_f32=False, ldf=None, ldw=None):
        """feat_f32 / ldf / ldw: the features in fp32 with a row stride of ldf elements, the weight logits with a row stride of ldw (both
        live in one conv output [feature | logits]); default: bf16 [K*HW, C] and f32 [K*HW, Q]."""
        if scratch is None:
            scratch = torch.zeros((K, -(-HW // 128), Q, C + 1), dtype=torch.float32, device=m16.device)
        return self.add(SUMMARIZE, 1 if feat_f32 else 0, [K, HW, C, Q, C if ldf is None else ldf, Q if ldw is None else ldw], [], [feat, wl, m16, y, scratch])

    def add_pe(self, x, pe, y, *, B, n):
        return self.add(ADD_PE, 0, [B, n], [], [x, pe, y])

    def key_prep(self, key, aux, hi, lo, sc, *, n, query):
        return self.add(KEY_PREP, (1 | KEYPREP_LOOP) if query else 0, [n], [], [key, aux, hi, lo, sc])

    AFF_CSTRIDE = 32          # ints between the candidate counters of consecutive queries (one cache line each)

    AFF_DMA = int(os.environ.get('CUTIE_AMD_AFF_DMA', '0'))    # 1: the LDS-DMA kernel (aff_score4_kernel) also for 2 sets per wave
    AFF_NQ = int(os.environ.get('CUTIE_AMD_AFF_NQ', '2'))      # 16-query column sets per wave of AFF_SCORE (1, 2: aff_score_kernel; 4: aff_score4_kernel)

    def aff_score(self, Ahi, Alo, scale, Bhi, Blo, c, out, cand_val, cand_idx, count, *, HW, HWp, ranges, cap, mode, gmax_precedes_tau=False, nq=None, dma=None,
                  frames=1, extra_lds_kb=0, prio=False, banks=None):
        """gmax_precedes_tau (mode 1): `out` (= tau) sits right behind the [HWp, Gld] maxima of pass 0 in memory; the kernel then
        skips every (tile, 16-query set) that cannot hold a candidate.  nq: query column sets per wave (default AFF_NQ; every
        choice computes the same bits).  frames > 1: the query operands of that many frames, HWp rows each (HW real ones), stacked -- every
        per-query array (c, maxima, tau, candidate lists, counters) is then indexed by the stacked row.
        banks = (table, n) (clips in lock step, ABI 4): stacked frame e reads bank e % n -- table: u64 [n, 3] = the (A_hi, A_lo, scale) bases of the
        banks, which share the token ranges (Ahi / Alo / scale are then ignored); needs nq = 2 and HWp % 128 == 0."""
        ranges = [(s, n) for (s, n) in ranges if n > 0]
        assert 1 <= len(ranges) <= 3
        G = sum(-(-n // 16) for _, n in ranges)
        ints = [HW, HWp, len(ranges)]
        for r in range(3):
            ints += list(ranges[r]) if r < len(ranges) else [0, 0]
        ints += [G, cap, mode, self.AFF_NQ if nq is None else nq, 0, int(extra_lds_kb), self.AFF_DMA if dma is None else int(dma)]
        if frames > 1:
            ints[1] = frames * HWp
            ints += [HWp]
        ptrs = [Ahi, Alo, scale, Bhi, Blo, c, out, cand_val, cand_idx, count]
        flags = (1 if (gmax_precedes_tau and mode == 1) else 0) | (F_AFF_PRIO if (prio and PRIO) else 0)
        if banks is not None:
            assert frames > 1 and frames % banks[1] == 0 and ints[12] == 2 and HWp % 128 == 0
            ints += [banks[1]]
            ptrs += [None, banks[0]]
            flags |= 4
        return self.add(AFF_SCORE, flags, ints, [], ptrs)

    def aff_select(self, gmax, tau, *, HW, HWp, G, top_k, clear_count=None, ticks=(), zero=None, frames=1, prio=False):
        """clear_count: pass 1's candidate counters, zeroed here; ticks: up to two (life, n) ranges advanced by one (USAGE_TICK);
        zero = (buffer, n): f32 range cleared instead (excludes ticks: the usage side buffer of a look-ahead read-out).
        frames > 1: stacked queries, see aff_score."""
        if zero is not None:
            assert not ticks
            return self.add(AFF_SELECT, 1 | SELECT_COARSE | (F_AFF_PRIO if (prio and PRIO) else 0), [HW, HWp, G, top_k, zero[1], 0, frames], [], [gmax, tau, clear_count, zero[0], None])
        ticks = list(ticks) + [(None, 0)] * (2 - len(ticks))
        assert len(ticks) == 2
        return self.add(AFF_SELECT, SELECT_COARSE | (F_AFF_PRIO if (prio and PRIO) else 0), [HW, HWp, G, top_k, ticks[0][1], ticks[1][1], frames], [], [gmax, tau, clear_count, ticks[0][0], ticks[1][0]])

    def aff_readout(self, cand_val, cand_idx, count, vptrs, usage, y, overflow, *, HW, cap, top_k, K, CV, frames=1, HWp=0, usage_stride=0, prio=False, banks=1, usage_fx=False):
        """frames > 1: stacked queries (HWp rows per frame, see aff_score); frame f's read-out goes to y[f] ([frames, K, HW, CV]) and its
        usage to usage + f * usage_stride counters.  banks > 1: frame f gathers from bank f % banks -- vptrs: u64 [banks, K].
        usage_fx: the usage counters are unsigned 64-bit fixed point (2^-40) instead of f32 -- integer atomics, sums independent of the order of arrival."""
        assert banks == 1 or frames % banks == 0
        return self.add(AFF_READOUT, (F_AFF_PRIO if (prio and PRIO) else 0) | (1 if usage_fx else 0), [HW, cap, top_k, K, CV, frames, HWp, usage_stride, banks], [], [cand_val, cand_idx, count, vptrs, usage, y, overflow])

    def memset32(self, dst, n, value=0):
        return self.add(MEMSET32, 0, [n, value], [], [dst])

    def bank_write(self, copies=(), fills=()):
        """One launch for the contiguous copies [(src, dst, nbytes)] (<= 6) and 32-bit fills [(dst, words, pattern)] (<= 2) of a memory
        insertion; more of either spill into further launches."""
        copies, fills = list(copies), list(fills)
        idx = None
        while copies or fills:
            c, copies = copies[:6], copies[6:]
            fl, fills = fills[:2], fills[2:]
            ints, ptrs = [0] * 10, [0] * 14
            for s, (src, dst, nbytes) in enumerate(c):
                assert nbytes % 4 == 0 and nbytes > 0
                ints[s], ptrs[2 * s], ptrs[2 * s + 1] = nbytes // 4, src, dst
            for t, (dst, words, pattern) in enumerate(fl):
                ints[6 + t], ints[8 + t], ptrs[12 + t] = words, pattern, dst
            idx = self.add(BANK_WRITE, 0, ints, [], ptrs)
        return idx

    def copy2d(self, src, dst, *, rows, rowbytes, src_stride, dst_stride):
        assert rowbytes % 4 == 0 and src_stride % 4 == 0 and dst_stride % 4 == 0
        return self.add(COPY2D, 0, [rows, rowbytes, src_stride, dst_stride], [], [src, dst])

    def axpy(self, x, y, *, n, a=1.0):
        return self.add(AXPY, 0, [n], [a], [x, y])

    def usage_tick(self, life, n, life2=None, n2=0, use=None, delta=None, n_use=0, delta_fx=False, clear_delta=False):
        """life[:n] += 1, life2[:n2] += 1, use[:n_use] += delta[:n_use] -- one launch (any part may be absent).  delta_fx: delta holds the unsigned
        64-bit fixed-point sums (2^-40) of AFF_READOUT(usage_fx=True); clear_delta (with delta_fx): and is zero again afterwards."""
        assert delta_fx or not clear_delta
        return self.add(USAGE_TICK, (1 if delta_fx else 0) | (2 if clear_delta else 0), [n, n2, n_use], [], [life, life2, use, delta])

    RS_SPLIT = 16             # chunks of the all-pairs rank count (csrc/bank.hip)

    def rank_select(self, use, life, order, *, n, k, scratch=None, gathers=(), zero=None):
        """scratch: i32 [16 * n] partial ranks (allocated here when not given); gathers: up to two (src, dst, row bytes) -- dst[r] =
        src[order[r]] for r < k, done by the launch that scatters the order; zero = (u32 buffer, words) cleared by that launch."""
        if scratch is None:
            scratch = torch.empty((self.RS_SPLIT * n,), dtype=torch.int32, device=use.device)
            self.keep.append(scratch)
        gathers = list(gathers) + [(None, None, 0)] * (2 - len(gathers))
        assert len(gathers) == 2 and all(g[2] % 16 == 0 for g in gathers)
        z = zero if zero is not None else (None, 0)
        return self.add(RANK_SELECT, 0, [n, k, gathers[0][2] // 4, gathers[1][2] // 4, z[1]], [],
                        [use, life, order, scratch, gathers[0][0], gathers[0][1], gathers[1][0], gathers[1][1], z[0]])

    def gather_rows(self, src, order, dst, *, k, rowbytes, src_stride, dst_stride):
        return self.add(GATHER_ROWS, 0, [k, rowbytes, src_stride, dst_stride], [], [src, order, dst])

    @staticmethod
    def consol_lds(n):
        """Row length of the similarity matrix of a consolidation: n rounded up to a multiple of 32."""
        return -(-n // 32) * 32

    def consol_aff(self, ckey, cshr, pkey, psel, S, colmax, *, n, P):
        """S f32 [P, consol_lds(n)] (similarities, -inf in the padding), colmax u32 [P] (cleared by the caller)."""
        return self.add(CONSOL_AFF, 0, [n, P, self.consol_lds(n)], [], [ckey, cshr, pkey, psel, S, colmax])

    @staticmethod
    def consol_scratch_floats(n, P, C, K):
        nchunk = -(-OpList.consol_lds(n) // 256)
        return nchunk * (K * P * C + 2 * P)

    def consol_read(self, S, colmax, vptrs, cshr, scratch, out_shr, *, n, P, C, K, src, dst):
        return self.add(CONSOL_READ, 0, [n, P, C, K, self.consol_lds(n), src, dst], [], [S, colmax, vptrs, cshr, scratch, out_shr])

    def prob_to_id(self, prob, lut, out, *, P, H, W, plane, ldrow):
        """out dtype picks the kernel: uint8 / int32 / int64."""
        code = {torch.uint8: 0, torch.int32: 1, torch.int64: 2}[out.dtype]
        return self.add(PROB_TO_ID, code, [P, H, W, plane, ldrow], [], [prob, lut, out])

    def resize(self, src, dst, *, C, H, W, OH, OW, plane, ldrow, nearest=False):
        return self.add(RESIZE, 1 if nearest else 0, [C, H, W, OH, OW, plane, ldrow], [], [src, dst])

    def flip_w(self, src, dst, *, rows, W, slds=None, dlds=None, alpha=1.0, beta=0.0):
        return self.add(FLIP_W, 0, [rows, W, W if slds is None else slds, W if dlds is None else dlds], [alpha, beta], [src, dst])

    def cast(self, src, dst, *, n, to_f32=False):
        return self.add(CAST, 1 if to_f32 else 0, [n], [], [src, dst])
