"""Per-kernel parity on the MI355X: every op kind of include/cutie_hip.h is run through the HIP kernel (via the
C ABI, cutie_exec) and through the torch interpreter of the same descriptor (tests/mock_exec.py) on identical
seeded inputs.  Tolerances: bf16 outputs within 2 bf16 ulps of the tensor scale (accumulation order differs),
fp32 outputs rtol 2e-3 (bf16 operands, fp32 accumulate), integer / index outputs bit-exact.
"""
import math
import numpy as np
import pytest
import torch

from cutie_amd import _lib, ops as O
from cutie_amd.model.weights import pack_conv, pack_linear
from mock_exec import MockExecutor

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def run_both(build, seed=0):
    """build(dev, g) -> (OpList, {name: output tensor}).  Returns (hip outputs, mock outputs) on CPU."""
    res = []
    for dev, ex in (('cuda', _lib.HipExecutor()), ('cpu', MockExecutor())):
        ol, outs = build(dev, _gen(seed))
        arr = ol.finalize()
        ex.run(arr)
        if dev == 'cuda':
            torch.cuda.synchronize()
        res.append({k: v.detach().cpu().clone() for k, v in outs.items()})
    return res


def check(hip, ref, name='', rtol=None):
    for k in ref:
        a, b = hip[k], ref[k]
        assert a.shape == b.shape, (name, k, a.shape, b.shape)
        if not b.is_floating_point():
            assert torch.equal(a, b), (name, k, 'integer mismatch', int((a != b).sum()))
            continue
        tol = rtol if rtol is not None else (1.6e-2 if b.dtype == BF16 else 2e-3)
        a, b = a.float(), b.float()
        assert torch.isfinite(a).all(), (name, k, 'non-finite output')
        scale = float(b.abs().max().clamp(min=1e-6))
        err = float((a - b).abs().max())
        if err > tol * scale:
            idx = np.unravel_index(int((a - b).abs().argmax()), a.shape)
            raise AssertionError(f'{name}:{k} max|d|={err:.4g} scale={scale:.4g} tol={tol} at {idx} hip={float(a[idx])} ref={float(b[idx])}')


def rnd(g, shape, dtype=BF16, dev='cpu', scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


# ---- CONV ------------------------------------------------------------------------------------------
CONV_CASES = [
    # B,H,W,C1,Cout,k,stride, extras
    dict(B=1, H=30, W=54, C1=512, Cout=256, k=1),
    dict(B=1, H=30, W=54, C1=256, Cout=1024, k=1, res=True, act=O.ACT_RELU),
    dict(B=1, H=60, W=108, C1=128, Cout=128, k=3, stride=2, act=O.ACT_RELU),
    dict(B=1, H=64, W=96, C1=8, Cout=64, k=7, stride=2, pad=3, act=O.ACT_RELU),
    dict(B=3, H=30, W=54, C1=256, Cout=256, k=3, relu_in=True, act=O.ACT_RELU),
    dict(B=3, H=30, W=54, C1=256, Cout=256, k=1, res=True, res_bcast=True),
    dict(B=3, H=30, W=54, C1=256, C2=8, Cout=256, k=1, res=True),
    dict(B=2, H=30, W=54, C1=256, C2=256, Cout=768, k=3, out_f32=True),
    dict(B=3, H=24, W=40, C1=128, Cout=1, k=3, relu_in=True, out_f32=True),
    dict(B=1, H=30, W=54, C1=256, Cout=1, k=3, out_f32=True, act=O.ACT_SQ1),
    dict(B=1, H=30, W=54, C1=256, Cout=64, k=3, out_f32=True, act=O.ACT_SIGMOID),
    dict(B=3, H=30, W=54, C1=256, Cout=16, k=1, out_f32=True),
    dict(B=2, H=17, W=23, C1=64, Cout=96, k=3),                       # ragged M and Cout
    dict(B=1, H=120, W=216, C1=64, Cout=64, k=1, act=O.ACT_RELU),
    dict(B=1, H=9, W=7, C1=32, Cout=40, k=3, stride=2),
    dict(B=2, H=100, W=90, C1=128, Cout=1, k=3, relu_in=True, out_f32=True),             # Cout = 1 on a large map: LDS-patch kernel, ragged tiles
    dict(B=1, H=130, W=131, C1=64, Cout=1, k=3, act=O.ACT_SIGMOID),                       # ... bf16 out, 8 lanes per pixel
]


def _conv_build(c, tile, splitk=1):
    def build(dev, g):
        B, H, W, C1, Cout, k = c['B'], c['H'], c['W'], c['C1'], c['Cout'], c['k']
        C2 = c.get('C2', 0)
        stride, pad = c.get('stride', 1), c.get('pad', (k - 1) // 2)
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        w = torch.randn(Cout, C1 + C2, k, k, generator=g) / math.sqrt((C1 + C2) * k * k)
        b = torch.randn(Cout, generator=g) * 0.1
        pc = pack_conv(w, b, dev, segs=[(C1, C1)] + ([(C2, C2)] if C2 else []))
        x1 = rnd(g, (B, H, W, C1), dev=dev)
        x2 = rnd(g, (B, H, W, C2), dev=dev) if C2 else None
        res = rnd(g, (1 if c.get('res_bcast') else B, OH, OW, Cout), dev=dev) if c.get('res') else None
        y = torch.zeros((B, OH, OW, Cout), dtype=F32 if c.get('out_f32') else BF16, device=dev)
        ol = O.OpList()
        ol.conv(x1, pc, y, B=B, H=H, W=W, C1=C1, ldx1=C1, OH=OH, OW=OW, ldy=Cout, stride=stride, pad=pad, x2=x2, C2=C2,
                ldx2=C2, res=res, ldr=Cout, res_bcast=c.get('res_bcast', False), relu_in=c.get('relu_in', False),
                act=c.get('act', O.ACT_NONE), out_f32=c.get('out_f32', False), tile=tile, splitk=splitk)
        return ol, {'y': y}
    return build


@pytest.mark.parametrize('ci', range(len(CONV_CASES)))
def test_conv_auto_tile(ci):
    hip, ref = run_both(_conv_build(CONV_CASES[ci], None), seed=ci)
    check(hip, ref, f'conv[{ci}]')


def _k_tiles(c, tile):
    cin = c['C1'] + c.get('C2', 0)
    return -(-(cin * c['k'] * c['k']) // 128) * 128 // O.TILES[tile][2]


@pytest.mark.parametrize('tile', sorted(O.TILES))
@pytest.mark.parametrize('ci', [0, 2, 4, 6, 7, 12, 14])
def test_conv_every_tile(ci, tile):
    c = CONV_CASES[ci]
    if tile in O.TILES and O.TILES[tile][2] > 32 and c['C1'] + c.get('C2', 0) < 32:
        pytest.skip('BK > 32 needs Cin >= 32')
    if _k_tiles(c, tile) % O.TILE_WK.get(tile, 1):
        pytest.skip('the K groups of this tile do not divide the K tiles')
    hip, ref = run_both(_conv_build(c, tile), seed=100 + ci)
    check(hip, ref, f'conv[{ci}] tile{tile}')


LAYER_CASES = [
    dict(B=3, H=30, W=54, C1=256, Cout=256, k=3, relu_in=True, act=O.ACT_RELU),          # CAResBlock conv (480p, 3 objects)
    dict(B=1, H=30, W=54, C1=1024, Cout=256, k=1),                                        # pix_feat_proj
    dict(B=1, H=120, W=216, C1=64, Cout=64, k=3, act=O.ACT_RELU),                        # ResNet layer1 3x3
    dict(B=1, H=60, W=108, C1=256, Cout=512, k=1, stride=2),                              # 1x1 stride-2 downsample
    dict(B=2, H=17, W=23, C1=64, Cout=96, k=3, res=True),                                 # ragged M and Cout, residual
    dict(B=2, H=18, W=22, C1=128, Cout=72, k=3, stride=2, relu_in=True),                  # 3x3 stride 2
    dict(B=3, H=30, W=54, C1=256, C2=256, Cout=768, k=3),                                 # GRU transform: two sources
    dict(B=3, H=9, W=7, C1=128, C2=128, Cout=40, k=1, res=True, res_bcast=True, out_f32=True, relu_in=True),
    dict(B=1, H=5, W=6, C1=128, Cout=64, k=3),                                            # M smaller than any tile
]


DMA_CASES = LAYER_CASES + [
    dict(B=1, H=30, W=54, C1=256, Cout=1024, k=1, res=True, act=O.ACT_RELU),             # ResNet conv3 + residual
    dict(B=1, H=36, W=40, C1=128, Cout=128, k=3, stride=2, act=O.ACT_RELU),              # 3x3 stride 2: halo on two sides only
    dict(B=3, H=30, W=54, C1=256, Cout=768, k=1, res=True),                               # transformer pixel projections
    dict(B=2, H=21, W=19, C1=64, C2=192, Cout=130, k=3, relu_in=True, out_f32=True),     # unequal sources, ragged everything
    dict(B=1, H=120, W=216, C1=128, Cout=128, k=3, relu_in=True, act=O.ACT_RELU),        # decoder 3x3 at stride 4 (large M)
    dict(B=1, H=16, W=16, C1=2048, Cout=64, k=1),                                         # long K loop (32 tiles)
    dict(B=1, H=12, W=20, C1=64, Cout=64, k=1),                                           # one K tile: shorter than the ring
    dict(B=1, H=12, W=20, C1=128, Cout=64, k=1, act=O.ACT_SIGMOID, out_f32=True),        # two K tiles
    dict(B=2, H=21, W=19, C1=128, C2=256, Cout=130, k=3, relu_in=True, out_f32=True),    # two sources, both multiples of 128
    dict(B=1, H=23, W=31, C1=384, Cout=72, k=3, stride=2, act=O.ACT_RELU),               # 3 K tiles of 128 per tap, stride 2
]


@pytest.mark.parametrize('tile', sorted(O.DMA_TILES))
@pytest.mark.parametrize('ci', range(len(DMA_CASES)))
def test_conv_dma_tiles(ci, tile):
    """conv_dma_kernel (tiles 60..: LDS-DMA staging, halo by out-of-range buffer offsets) against the interpreter."""
    c = DMA_CASES[ci]
    if not O.dma_tile_ok(tile, cin=c['C1'] + c.get('C2', 0), kh=c['k'], c2=c.get('C2', 0)):
        assert O.DMA_TILES[tile][2] == 128
        pytest.skip('128-channel K tile: sources are not multiples of 128')
    hip, ref = run_both(_conv_build(c, tile), seed=400 + ci)
    check(hip, ref, f'dma[{ci}] tile{tile}')


NARROW_CASES = [            # 3x3 / stride 1 / pad 1 on narrow maps (written for the strip-resident kernel of round 2; kept as cases of the halo tiles)
    dict(B=3, H=30, W=54, C1=256, Cout=256, k=3, relu_in=True, act=O.ACT_RELU),          # CAResBlock conv (480p, 3 objects)
    dict(B=1, H=30, W=54, C1=256, Cout=64, k=3, out_f32=True, act=O.ACT_SIGMOID),        # key projection e_proj
    dict(B=2, H=17, W=23, C1=64, Cout=96, k=3, res=True),                                 # ragged M and Cout, residual, one slice
    dict(B=3, H=9, W=7, C1=128, Cout=40, k=3, res=True, res_bcast=True, out_f32=True),   # tiny map: several objects inside a strip
    dict(B=1, H=40, W=60, C1=256, Cout=256, k=3),
    dict(B=2, H=21, W=19, C1=64, C2=192, Cout=130, k=3, relu_in=True, out_f32=True),     # two sources
    dict(B=3, H=30, W=54, C1=256, C2=256, Cout=768, k=3),                                 # sensory-update transform conv
    dict(B=5, H=5, W=7, C1=256, Cout=256, k=3),                                           # M = 175: a tile holds whole objects
]


PC_CASES = DMA_CASES + NARROW_CASES[3:5] + [NARROW_CASES[7]] + [
    dict(B=2, H=13, W=37, C1=128, Cout=136, k=3, res=True, act=O.ACT_RELU),               # ragged patches in both directions, Cout % 8 == 0 but not % 32
    dict(B=1, H=8, W=16, C1=64, Cout=64, k=3, relu_in=True),                              # exactly one 8 x 16 patch
    dict(B=1, H=33, W=17, C1=192, Cout=100, k=3, out_f32=True, act=O.ACT_SQ1),            # three slices, ragged Cout (not % 4)
    dict(B=3, H=30, W=54, C1=256, Cout=256, k=1, res=True, res_bcast=True, act=O.ACT_RELU),
]


@pytest.mark.parametrize('tile', sorted(O.PC_TILES))
@pytest.mark.parametrize('ci', range(len(PC_CASES)))
def test_conv_pc_tiles(ci, tile):
    """conv_pc_kernel (tiles 100..: producer / consumer waves; 120..: 3x3 halo patch resident in LDS; epilogue straight from the
    accumulators with the permuted weight-row fetch) against the interpreter."""
    c = PC_CASES[ci]
    if not O.pc_tile_ok(tile, cin=c['C1'] + c.get('C2', 0), kh=c['k'], stride=c.get('stride', 1), pad=c.get('pad', (c['k'] - 1) // 2), c2=c.get('C2', 0)):
        pytest.skip('halo tiles: 3x3 / stride 1 / pad 1 only')
    hip, ref = run_both(_conv_build(c, tile), seed=1100 + ci)
    check(hip, ref, f'pc conv[{ci}] tile{tile}')


@pytest.mark.parametrize('splitk', [2, 3, 4, 9])
@pytest.mark.parametrize('tile', [0, 5, 7, 8, 11, 13, 16, 20, 22])
@pytest.mark.parametrize('ci', [0, 2, 4, 5, 6, 7, 12, 14])
def test_conv_split_k(ci, tile, splitk):
    """grid.z K slices + reduce launch: same result as the single-pass kernel for every epilogue variant
    (residual / broadcast residual / 2-source / f32 out / ragged M and Cout / stride 2)."""
    c = CONV_CASES[ci]
    cin = c['C1'] + c.get('C2', 0)
    nk = _k_tiles(c, tile) // O.TILE_WK.get(tile, 1)
    if _k_tiles(c, tile) % O.TILE_WK.get(tile, 1) or splitk > nk or nk % splitk:
        pytest.skip('the slice count must divide the K tiles')
    if splitk * c['B'] * c['H'] * c['W'] * ((c['Cout'] + 7) & ~7) > O.SPLITK_PART_FLOATS:
        pytest.skip('partials exceed the scratch')
    hip, ref = run_both(_conv_build(c, tile, splitk), seed=500 + ci)
    check(hip, ref, f'conv[{ci}] tile{tile} splitk{splitk}')


def test_conv_split_k_repeatable():
    """Two launches of the same split-K conv give bit-identical outputs (slices are summed in slice order)."""
    dev = 'cuda'
    c = CONV_CASES[4]
    g = torch.Generator().manual_seed(7)
    ol, outs = _conv_build(c, 8, 3)(dev, g)
    arr = ol.finalize()
    ex = _lib.HipExecutor()
    ex.run(arr)
    torch.cuda.synchronize()
    a = outs['y'].clone()
    ex.run(arr)
    torch.cuda.synchronize()
    assert torch.equal(a, outs['y'])


def test_conv_split_k_rejects_bad_factor():
    dev = 'cuda'
    g = torch.Generator().manual_seed(7)
    ol, _ = _conv_build(CONV_CASES[0], 8, 3)(dev, g)        # 512/128 = 4 K tiles: 3 slices do not divide them
    arr = ol.finalize()
    with pytest.raises(Exception):
        _lib.HipExecutor().run(arr)


@pytest.mark.parametrize('ci', [8, 9])
def test_conv_cout1_kernel(ci):
    hip, ref = run_both(_conv_build(CONV_CASES[ci], O.COUT1_TILE), seed=300 + ci)
    check(hip, ref, f'conv cout1 [{ci}]')


ROWS_CASES = [CONV_CASES[15], CONV_CASES[16],
              dict(B=3, H=120, W=216, C1=128, Cout=1, k=3, relu_in=True, out_f32=True),      # the decoder's logits head at 480p, 3 objects
              dict(B=1, H=65, W=64, C1=256, Cout=1, k=3, act=O.ACT_RELU)]                    # 32 lanes per pixel, ragged rows


@pytest.mark.parametrize('ci', range(len(ROWS_CASES)))
def test_conv_cout1_rows_kernel(ci):
    """Cout = 1, 3x3 on maps of >= 4096 pixels: conv_cout1_rows_kernel (a thread walks 4 output rows of a column) against the
    interpreter: ragged rows and columns, 8 / 16 / 32 lanes per pixel, ReLU on the input, every output form."""
    hip, ref = run_both(_conv_build(ROWS_CASES[ci], O.COUT1_TILE), seed=700 + ci)
    check(hip, ref, f'conv cout1 rows [{ci}]')


@pytest.mark.parametrize('ci', [0, 2])
def test_conv_cout1_tile_kernel_matches_rows_kernel(ci, monkeypatch):
    """Cin = 128: conv_cout1_tile_kernel (input tile in LDS, dot products on MFMA since round 5) against conv_cout1_rows_kernel (flags &
    CUTIE_F_TILE_OFF: fp32 VALU, tap by tap): the same bf16 products summed in fp32 in another order -- equal to fp32 rounding (bf16 outputs:
    at most one ulp), ragged tiles and image borders included."""
    outs = []
    for off in (0, 128):
        monkeypatch.setattr(O, 'F_TILE_OFF', off)
        ol, t = _conv_build(ROWS_CASES[ci], O.COUT1_TILE)('cuda', torch.Generator().manual_seed(41))
        ol.run()
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in t.items()})
    for k in outs[0]:
        a, b = outs[0][k].float(), outs[1][k].float()
        tol = 1e-5 if outs[0][k].dtype == F32 else 8e-3          # (fp32 outputs: summation order; bf16 outputs: one ulp where the rounding flips)
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= tol * float(b.abs().max().clamp(min=1.0)), (k, float((a - b).abs().max()))


def test_conv_cout1_1x1_relu_in():
    c = dict(B=3, H=30, W=54, C1=256, Cout=1, k=1, relu_in=True, out_f32=True)
    hip, ref = run_both(_conv_build(c, O.COUT1_TILE), seed=9)
    check(hip, ref, 'conv cout1 1x1')


def test_conv_strided_channel_slices():
    """ldx / ldy larger than C (reading / writing channel slices of wider NHWC buffers)."""
    def build(dev, g):
        B, H, W = 2, 12, 20
        big = rnd(g, (B, H, W, 96), dev=dev)
        w = torch.randn(48, 64, 1, 1, generator=g) / 8
        pc = pack_conv(w, None, dev)
        out = torch.zeros((B, H, W, 80), dtype=BF16, device=dev)
        ol = O.OpList()
        ol.conv(big.view(-1)[16:], pc, out.view(-1)[8:], B=B, H=H, W=W, C1=64, ldx1=96, OH=H, OW=W, ldy=80)
        return ol, {'out': out}
    hip, ref = run_both(build)
    check(hip, ref, 'conv slices')


# ---- elementwise family ------------------------------------------------------------------------------
def test_maxpool():
    for relu in (False, True):
        def build(dev, g):
            x = rnd(g, (2, 24, 36, 64), dev=dev)
            y = torch.zeros((2, 12, 18, 64), dtype=BF16, device=dev)
            ol = O.OpList()
            ol.maxpool(x, y, B=2, H=24, W=36, C=64, relu=relu)
            return ol, {'y': y}
        check(*run_both(build), name=f'maxpool relu={relu}', rtol=1e-6)


@pytest.mark.parametrize('geo', [(480, 854, 480, 864, 5, 0, 3), (480, 854, 480, 864, 5, 0, 0), (100, 120, 112, 128, 4, 6, 2), (16, 16, 16, 16, 0, 0, 1),
                                 (30, 43, 32, 48, 2, 1, 5), (1080, 1920, 1088, 1920, 0, 4, 0)])
def test_stem_kernel(geo):
    """STEM (IMG_PREP + 7x7 / stride-2 conv + 3x3 / stride-2 max pool in one launch, csrc/stem.hip) against the interpreter and against
    the three launches it replaces: frame borders, pad geometry, ragged pooled tiles, mask / others planes of K objects, no masks."""
    h0, w0, H, W, pl, pt, K = geo

    def build(dev, g):
        img = torch.rand((3, h0, w0), generator=g).to(dev)
        masks = None
        if K:
            masks = torch.rand((K, H, W), generator=g)
            masks = (masks * (torch.rand((K, H, W), generator=g) > 0.5)).to(dev)
        Kk = max(K, 1)
        wt = torch.randn((64, 8, 7, 7), generator=g) / math.sqrt(147)
        wt[:, 5:] = 0
        if not K:
            wt[:, 3:] = 0
        pc = pack_conv(wt, torch.randn(64, generator=g) * 0.1, dev)
        mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
        y = torch.zeros((Kk, H // 4, W // 4, 64), dtype=BF16, device=dev)
        y3 = torch.zeros((Kk, H // 4, W // 4, 64), dtype=BF16, device=dev)
        x8 = torch.zeros((Kk, H, W, 8), dtype=BF16, device=dev)
        cv = torch.zeros((Kk, H // 2, W // 2, 64), dtype=BF16, device=dev)
        ol = O.OpList()
        ol.keep += [pc.weight]
        for relu, out in ((True, y),):
            ol.stem(img, masks, pc, out, h0=h0, w0=w0, H=H, W=W, pad_left=pl, pad_top=pt, K=Kk, mean=mean, std=std, relu=relu)
        ol.img_prep(img, masks, x8, h0=h0, w0=w0, H=H, W=W, pad_left=pl, pad_top=pt, K=Kk, mean=mean, std=std)
        ol.conv(x8, pc, cv, B=Kk, H=H, W=W, C1=8, ldx1=8, OH=H // 2, OW=W // 2, ldy=64, stride=2, pad=3)
        ol.maxpool(cv, y3, B=Kk, H=H // 2, W=W // 2, C=64, relu=True)
        return ol, {'y': y, 'y3': y3}
    hip, ref = run_both(build, seed=sum(geo))
    check(hip, ref, name=f'stem {geo}')
    d = float((hip['y'].float() - hip['y3'].float()).abs().max())
    assert d <= 1.6e-2 * max(1.0, float(hip['y3'].float().abs().max())), d


def test_img_prep():
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    for K in (0, 3):
        def build(dev, g):
            img = torch.rand((3, 30, 43), generator=g).to(dev)
            H, W = 32, 48
            masks = torch.rand((K, H, W), generator=g).to(dev) if K else None
            y = torch.zeros((max(K, 1), H, W, 8), dtype=BF16, device=dev)
            ol = O.OpList()
            ol.img_prep(img, masks, y, h0=30, w0=43, H=H, W=W, pad_left=2, pad_top=1, K=max(K, 1), mean=mean, std=std)
            return ol, {'y': y}
        check(*run_both(build), name=f'img_prep K={K}')


def test_upsample2x_add():
    def build(dev, g):
        x = rnd(g, (3, 7, 9, 128), dev=dev)
        skip = rnd(g, (1, 14, 18, 128), dev=dev)
        y = torch.zeros((3, 14, 18, 128), dtype=BF16, device=dev)
        ol = O.OpList()
        ol.upsample2x_add(x, skip, y, B=3, h=7, w=9, C=128)
        return ol, {'y': y}
    check(*run_both(build), name='upsample2x_add')


def test_area_down():
    def build(dev, g):
        x = rnd(g, (2, 16, 24, 128), dev=dev)
        y = torch.zeros((2, 4, 6, 128), dtype=BF16, device=dev)
        lg = torch.randn((2, 16, 24), generator=g).to(dev)
        y2 = torch.ones((2, 4, 6, 8), dtype=BF16, device=dev)
        ol = O.OpList()
        ol.area_down(x, y, B=2, H=16, W=24, C=128, ldx=128, ldy=128, r=4)
        ol.area_down(lg, y2, B=2, H=16, W=24, C=1, ldx=1, ldy=8, r=4, f32_in=True, Cz=8)
        return ol, {'y': y, 'y2': y2}
    check(*run_both(build), name='area_down')


def test_mask_down():
    def build(dev, g):
        m = torch.rand((3, 64, 96), generator=g).to(dev)
        pair = torch.ones((3, 4, 6, 8), dtype=BF16, device=dev)
        m16 = torch.zeros((3, 4, 6), dtype=F32, device=dev)
        ol = O.OpList()
        ol.mask_down(m, pair, m16, K=3, H=64, W=96)
        return ol, {'pair': pair, 'm16': m16}
    check(*run_both(build), name='mask_down', rtol=1e-2)


def test_area_down3():
    """Three area poolings in one launch, written side by side into one concat buffer (the sensory update's inputs)."""
    def build(dev, g):
        K, h, w = 3, 6, 10
        p8 = rnd(g, (K, 2 * h, 2 * w, 128), dev=dev)
        p4 = rnd(g, (K, 4 * h, 4 * w, 128), dev=dev)
        lg = (torch.randn((K, 4 * h, 4 * w), generator=g) * 3).to(dev)
        CT = 128 + 128 + 64
        cat3 = torch.zeros((K, h, w, CT), dtype=BF16, device=dev)
        cat1 = torch.zeros((K, h, w, CT), dtype=BF16, device=dev)
        ol = O.OpList()
        segs = lambda y: [dict(x=p8, y=y, B=K, H=2 * h, W=2 * w, C=128, ldx=128, ldy=CT, r=2),
                          dict(x=p4, y=y.view(-1)[128:], B=K, H=4 * h, W=4 * w, C=128, ldx=128, ldy=CT, r=4),
                          dict(x=lg, y=y.view(-1)[256:], B=K, H=4 * h, W=4 * w, C=1, ldx=1, ldy=CT, r=4, f32_in=True, Cz=8)]
        ol.area_down3(segs(cat3))
        for sgm in segs(cat1):
            sgm = dict(sgm)
            ol.area_down(sgm.pop('x'), sgm.pop('y'), **sgm)
        return ol, {'cat3': cat3, 'cat1': cat1}
    hip, ref = run_both(build)
    check(hip, ref, 'area_down3')
    assert torch.equal(hip['cat3'].view(torch.int16), hip['cat1'].view(torch.int16))


def test_gap_eca():
    def build(dev, g):
        B, HW, C = 3, 1620, 256
        x = rnd(g, (B, HW, C), dev=dev)
        r = rnd(g, (B, HW, C), dev=dev)
        gap = torch.zeros((B, C), dtype=F32, device=dev)
        wk = (torch.randn(5, generator=g) * 0.6).to(dev)
        y = torch.zeros((B, HW, C), dtype=BF16, device=dev)
        ol = O.OpList()
        ol.gap(x, gap, B=B, HW=HW, C=C, partial_only=True)
        ol.eca_apply(x, gap, wk, r, y, B=B, HW=HW, C=C)
        return ol, {'gap': gap, 'y': y}
    hip, ref = run_both(build)
    check(hip, ref, 'gap/eca')


@pytest.mark.parametrize('tile', [66, 67, 63, 61, 70, 68, 82, 85, 86, 100, 102, 103, 105, 108, 110, 120, 121, 123, 125, 129, 131, 134])
@pytest.mark.parametrize('geo', [(3, 30, 54), (5, 5, 7), (2, 9, 16)])
def test_conv_gap_accumulation(tile, geo):
    """ECA's average pool riding on the convs of a CAResBlock: conv1 clears the accumulator, conv2 adds the fixed-point channel sums of
    its stored output (tiles straddling 1, 2 or several objects), ECA_APPLY turns them into means.  Against the interpreter, and the
    sums against the stored tensor EXACTLY: every stored value goes to fixed point (2^-20) before anything is summed, so the accumulator is the
    same integer whatever the tile (how it groups rows into fragments and waves) and whatever the order of the atomics -- fp32 partial
    sums (rounds 3-6) differed between two tiles of one K-order class about once per 35 M outputs, enough for a clip in lock step to leave
    its own run after a few dozen frames (tools/lockstep_soak.py)."""
    B, H, W = geo
    C = 256

    def build(dev, g):
        w1 = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)
        w2 = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)
        pc1, pc2 = pack_conv(w1, torch.randn(C, generator=g) * 0.1, dev, segs=[(C, C)]), pack_conv(w2, torch.randn(C, generator=g) * 0.1, dev, segs=[(C, C)])
        x = rnd(g, (B, H, W, C), dev=dev)
        t1, t2, y = (torch.zeros((B, H, W, C), dtype=BF16, device=dev) for _ in range(3))
        sums = torch.full((B, C), 12345, dtype=torch.int64, device=dev)        # stale contents: conv1 must clear them
        gap = torch.zeros((B, C), dtype=F32, device=dev)
        wk = (torch.randn(5, generator=g) * 0.6).to(dev)
        ol = O.OpList()
        kw = dict(B=B, H=H, W=W, C1=C, ldx1=C, OH=H, OW=W, ldy=C, pad=1, tile=tile)
        ol.conv(x, pc1, t1, relu_in=True, act=O.ACT_RELU, zero=sums, **kw)
        ol.conv(t1, pc2, t2, gap_acc=sums, **kw)
        ol.eca_apply(t2, gap, wk, x, y, B=B, HW=H * W, C=C, fixed_sums=sums)
        return ol, {'t2': t2, 'sums': sums, 'gap': gap, 'y': y}
    hip, ref = run_both(build, seed=31)
    check({k: hip[k] for k in ('t2', 'gap', 'y')}, {k: ref[k] for k in ('t2', 'gap', 'y')}, f'conv gap tile{tile}')
    exact = torch.round(hip['t2'].float().reshape(B, H * W, C).double() * 1048576.0).to(torch.int64).sum(1) * 16
    assert torch.equal(hip['sums'].cpu(), exact.cpu()), 'sums of the stored tensor: %d of %d accumulators differ' % (int((hip['sums'].cpu() != exact.cpu()).sum()), exact.numel())


@pytest.mark.parametrize('tile', [100, 103, 105, 110, 120, 123, 131, 134])
@pytest.mark.parametrize('C3', [8, 40, 256])
def test_conv_next_weights_touch_changes_nothing(tile, C3, monkeypatch):
    """A producer / consumer conv reads (and discards) the packed weights of the next conv(s) of the list on its way out (p9 / i22,
    p10 / i23): the outputs are bit-identical without, with whole tensors, and with ranges cut at 1000 bytes."""
    B, H, W, C = 2, 30, 54, 256

    def run(pf):
        monkeypatch.setattr(O, 'WEIGHT_PREFETCH', pf)
        g = _gen(77)
        w1 = pack_conv(torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9), torch.randn(C, generator=g) * 0.1, 'cuda', segs=[(C, C)])
        w2 = pack_conv(torch.randn(C, C, 1, 1, generator=g) / math.sqrt(C), None, 'cuda', segs=[(C, C)])
        w3 = pack_conv(torch.randn(C3, C, 3, 3, generator=g) / math.sqrt(C * 9), torch.randn(C3, generator=g) * 0.1, 'cuda', segs=[(C, C)])
        x = rnd(g, (B, H, W, C), dev='cuda')
        t1, t2 = (torch.zeros((B, H, W, C), dtype=BF16, device='cuda') for _ in range(2))
        t3 = torch.zeros((B, H, W, max(C3, 8)), dtype=BF16, device='cuda')
        ol = O.OpList()
        kw = dict(B=B, H=H, W=W, C1=C, ldx1=C, OH=H, OW=W)
        ol.conv(x, w1, t1, ldy=C, pad=1, tile=tile, **kw)
        ol.conv(t1, w2, t2, ldy=C, pad=0, tile=61, **kw)                     # no producer waves: its successor's weights ride on conv 1
        ol.conv(t2, w3, t3, ldy=max(C3, 8), pad=1, **kw)
        arr = ol.finalize()
        assert bool(arr['p'][0, 9]) == bool(pf) and bool(arr['p'][0, 10]) == bool(pf)
        ex = _lib.HipExecutor()
        for _ in range(3):
            ex.run(arr)
        torch.cuda.synchronize()
        return [t.cpu().clone() for t in (t1, t2, t3)]
    off, on, small = run(0), run(8 << 20), run(1000)
    for a, b, c in zip(off, on, small):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert float(off[2].float().abs().max()) > 0


def test_bank_write():
    """BANK_WRITE: six copies of very different lengths (one word ... 200k words, odd tails) and two fills in one launch; a second op
    with fewer segments; untouched neighbours stay untouched."""
    def build(dev, g):
        sizes = [1, 1620, 103680, 207361, 255, 1024]
        srcs = [torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64).to(torch.int32).to(dev) for n in sizes]
        dsts = [torch.full((n + 8,), 7, dtype=torch.int32, device=dev) for n in sizes]
        f0, f1 = torch.full((1620 + 8,), 7, dtype=torch.int32, device=dev), torch.full((5 + 8,), 7, dtype=torch.int32, device=dev)
        ol = O.OpList()
        ol.bank_write([(srcs[k], dsts[k].view(-1)[4:], 4 * sizes[k]) for k in range(6)], [(f0.view(-1)[4:], 1620, 0), (f1.view(-1)[4:], 5, 0x33D6BF95)])
        e0 = torch.full((40,), 7, dtype=torch.int32, device=dev)
        ol.bank_write([(srcs[1], e0.view(-1)[4:], 4 * 30)], [])
        e1 = torch.full((40,), 7, dtype=torch.int32, device=dev)
        ol.bank_write([], [(e1.view(-1)[4:], 30, -1)])
        outs = {f'd{k}': dsts[k] for k in range(6)}
        outs.update(f0=f0, f1=f1, e0=e0, e1=e1)
        for k in range(6):
            outs[f's{k}'] = srcs[k]
        return ol, outs
    hip, ref = run_both(build, seed=5)
    check(hip, ref, 'bank_write')
    for k, n in enumerate([1, 1620, 103680, 207361, 255, 1024]):
        assert torch.equal(hip[f'd{k}'][4:4 + n], hip[f's{k}']) and int((hip[f'd{k}'][:4] != 7).sum()) == 0 and int((hip[f'd{k}'][4 + n:] != 7).sum()) == 0
    assert int((hip['f0'][4:1624] != 0).sum()) == 0 and int((hip['f1'][4:9] != 0x33D6BF95).sum()) == 0 and int((hip['f1'][9:] != 7).sum()) == 0


def test_gru():
    def build(dev, g):
        n, C = 500, 256
        v = (torch.randn((n, 3 * C), generator=g) * 2).to(dev)
        h = torch.randn((n, C), generator=g).to(dev)
        hb = torch.zeros((n, C), dtype=BF16, device=dev)
        ol = O.OpList()
        ol.gru(v, h, hb, n=n, C=C)
        return ol, {'h': h, 'hb': hb}
    hip, ref = run_both(build)
    check({'h': hip['h']}, {'h': ref['h']}, 'gru', rtol=1e-4)
    check({'hb': hip['hb']}, {'hb': ref['hb']}, 'gru shadow')          # bf16: within one ulp of the rounding point


def test_seg_epilogue():
    def build(dev, g):
        K, h, w = 3, 12, 20
        lg = (torch.randn((K, h, w), generator=g) * 3).to(dev)
        agg = torch.zeros((K + 1, h, w), dtype=F32, device=dev)
        prob = torch.zeros((K + 1, 4 * h, 4 * w), dtype=F32, device=dev)
        lup = torch.zeros((K + 1, 4 * h, 4 * w), dtype=F32, device=dev)
        ol = O.OpList()
        ol.seg_agg(lg, agg, K=K, hw=h * w)
        ol.up4_softmax(agg, prob, lup, P=K + 1, h=h, w=w)
        return ol, {'agg': agg, 'prob': prob, 'lup': lup}
    check(*run_both(build), name='seg epilogue', rtol=2e-4)

    # the fused forms (<= 8 planes: four pixels per thread; <= 16: one): bit-identical to the two launches, also on one-column and odd maps
    for K, h, w in ((3, 12, 20), (9, 12, 20), (1, 7, 1), (7, 5, 3), (3, 30, 54), (2, 1, 9)):
        def build2(dev, g):
            lg = (torch.randn((K, h, w), generator=g) * 3).to(dev)
            agg = torch.zeros((K + 1, h, w), dtype=F32, device=dev)
            o = [torch.zeros((K + 1, 4 * h, 4 * w), dtype=F32, device=dev) for _ in range(4)]
            ol = O.OpList()
            ol.seg_agg(lg, agg, K=K, hw=h * w)
            ol.up4_softmax(agg, o[0], o[1], P=K + 1, h=h, w=w)
            ol.up4_softmax(lg, o[2], o[3], P=K + 1, h=h, w=w, from_logits=True)
            return ol, {'prob': o[0], 'lup': o[1], 'prob_f': o[2], 'lup_f': o[3]}
        hip, ref = run_both(build2)
        check(hip, ref, name=f'seg epilogue fused K={K}', rtol=2e-4)
        assert torch.equal(hip['prob'], hip['prob_f']) and torch.equal(hip['lup'], hip['lup_f'])


def test_mask_merge_and_agg():
    for fmode in (False, True):
        def build(dev, g):
            h0, w0, H, W = 30, 43, 32, 48
            Kold, Knew = 2, 4
            if fmode:
                inmask = torch.rand((5, h0, w0), generator=g).to(dev)
                src = torch.tensor([-1, 3, 1, 4], dtype=torch.int32).to(dev)
            else:
                inmask = torch.randint(0, 6, (h0, w0), generator=g).to(torch.int32).to(dev)
                src = torch.tensor([-1, 5, 2, 4], dtype=torch.int32).to(dev)
            pred = torch.softmax(torch.randn((Kold + 1, H, W), generator=g), 0).to(dev)
            planes = torch.zeros((Knew, H, W), dtype=F32, device=dev)
            prob = torch.zeros((Knew + 1, H, W), dtype=F32, device=dev)
            ol = O.OpList()
            ol.mask_merge(inmask, pred, src, planes, h0=h0, w0=w0, H=H, W=W, pad_left=2, pad_top=1, Knew=Knew, Kold=Kold,
                          nfloat=5 if fmode else 0, float_mode=fmode)
            ol.agg_softmax(planes, prob, K=Knew, HW=H * W)
            return ol, {'planes': planes, 'prob': prob}
        check(*run_both(build), name=f'mask_merge float={fmode}', rtol=2e-4)


def test_linear_layernorm_queryinit():
    def build(dev, g):
        M, Kd, N = 48, 256, 2048
        x = torch.randn((M, Kd), generator=g).to(dev)
        xa = torch.randn((M, Kd), generator=g).to(dev)
        pl = pack_linear(torch.randn((N, Kd), generator=g) / 16, torch.randn(N, generator=g) * 0.1, dev)
        y = torch.zeros((M, N), dtype=F32, device=dev)
        pl2 = pack_linear(torch.randn((Kd, N), generator=g) / 45, torch.randn(Kd, generator=g) * 0.1, dev)
        res = torch.randn((M, Kd), generator=g).to(dev)
        y2 = torch.zeros((M, Kd), dtype=F32, device=dev)
        ln = torch.zeros((M, Kd), dtype=F32, device=dev)
        gw, gb = torch.rand(Kd, generator=g).to(dev) + 0.5, torch.randn(Kd, generator=g).to(dev) * 0.1
        om = torch.rand((M, Kd + 1), generator=g).to(dev) + 0.1
        qi = torch.zeros((M, Kd), dtype=F32, device=dev)
        ol = O.OpList()
        ol.linear(x, pl, y, M=M, x_add=xa, add_rows=M, relu=True)
        ol.linear(y, pl2, y2, M=M, res=res)
        ol.layernorm(y2, gw, gb, ln, M=M, C=Kd)
        ol.query_init(om, qi, rows=M, C=Kd)
        return ol, {'y': y, 'y2': y2, 'ln': ln, 'qi': qi}
    check(*run_both(build), name='linear/ln', rtol=2e-3)


def test_linear_broadcast_add_rows():
    def build(dev, g):
        M, Kd, N = 32, 256, 512
        x = torch.randn((M, Kd), generator=g).to(dev)
        xa = torch.randn((16, Kd), generator=g).to(dev)
        pl = pack_linear(torch.randn((N, Kd), generator=g) / 16, None, dev)
        y = torch.zeros((M, N), dtype=F32, device=dev)
        ol = O.OpList()
        ol.linear(x, pl, y, M=M, x_add=xa, add_rows=16)
        return ol, {'y': y}
    check(*run_both(build), name='linear add_rows', rtol=2e-3)


@pytest.mark.parametrize('M,Kd,N,ld', [(48, 256, 256, 256), (16, 256, 768, 800), (80, 2048, 256, 256), (5, 384, 40, 48), (48, 264, 256, 256)])
def test_linear_shapes(M, Kd, N, ld):
    """Ragged M / N, strided output, K split over the 4 waves (MFMA kernel for Kd % 128 == 0, butterfly kernel otherwise)."""
    def build(dev, g):
        x = torch.randn((M, Kd), generator=g).to(dev)
        pl = pack_linear(torch.randn((N, Kd), generator=g) / math.sqrt(Kd), torch.randn(N, generator=g) * 0.1, dev)
        res = torch.randn((M, N), generator=g).to(dev)
        y = torch.zeros((M, ld), dtype=F32, device=dev)
        ol = O.OpList()
        ol.linear(x, pl, y, M=M, res=res, ldy=ld)
        return ol, {'y': y}
    check(*run_both(build, seed=M + N), name=f'linear {M}x{Kd}x{N}', rtol=2e-3)


def test_linear_fused_layernorm_and_add_cols():
    """LayerNorm fused in front of the projection (normalised rows kept as a side output), positional term feeding only
    the first add_cols outputs (merged q|k|v)."""
    def build(dev, g):
        M, Kd, N = 48, 256, 768
        x = (torch.randn((M, Kd), generator=g) * 3 + 1).to(dev)
        xa = torch.randn((M, Kd), generator=g).to(dev)
        pl = pack_linear(torch.randn((N, Kd), generator=g) / 16, torch.randn(N, generator=g) * 0.1, dev)
        gw, gb = (torch.rand(Kd, generator=g) + 0.5).to(dev), (torch.randn(Kd, generator=g) * 0.1).to(dev)
        y = torch.zeros((M, N), dtype=F32, device=dev)
        xn = torch.zeros((M, Kd), dtype=F32, device=dev)
        y2 = torch.zeros((M, N), dtype=F32, device=dev)
        ol = O.OpList()
        ol.linear(x, pl, y, M=M, x_add=xa, add_rows=M, add_cols=512, ln=(gw, gb), ln_out=xn)
        ol.linear(x, pl, y2, M=M, relu=True, ln=(gw, gb))
        return ol, {'y': y, 'xn': xn, 'y2': y2}
    check(*run_both(build), name='linear ln', rtol=2e-3)


# ---- attention -----------------------------------------------------------------------------------------
def _aux_inputs(g, K, HW, mode):
    lg = torch.randn((K, HW), generator=g) * 2
    if mode == 'nofg':
        lg[0] = -20.0          # object 0 never foreground  -> fg queries of object 0 get un-blocked
    if mode == 'allfg':
        lg[:] = -20.0
        lg[1] = 20.0           # object 1 foreground everywhere -> its bg queries get un-blocked
    return lg


@pytest.mark.parametrize('mode', ['mixed', 'nofg', 'allfg'])
def test_aux_mask_and_q2p(mode):
    def build(dev, g):
        K, Q, HW, C, heads = 3, 16, 1620, 256, 8
        lg = _aux_inputs(g, K, HW, mode).to(dev)
        fg = torch.zeros((K, HW), dtype=torch.uint8, device=dev)
        nfg = torch.zeros((K,), dtype=torch.int32, device=dev)
        q = torch.randn((K, Q, C), generator=g).to(dev)
        kv = rnd(g, (K, HW, 3 * C), dev=dev)
        y = torch.zeros((K, Q, C), dtype=F32, device=dev)
        ol = O.OpList()
        ol.aux_mask(lg, fg, nfg, K=K, HW=HW)
        ol.attn_q2p(q, kv, fg, nfg, y, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C)
        return ol, {'fg': fg, 'nfg': nfg, 'y': y}
    check(*run_both(build), name=f'aux_mask/q2p {mode}', rtol=3e-3)

    def build_fused(dev, g):                       # the form the frame uses: mask derived inside ATTN_Q2P, bit-identical to the two-op form
        K, Q, HW, C, heads = 3, 16, 1620, 256, 8
        lg = _aux_inputs(g, K, HW, mode).to(dev)
        fg = torch.zeros((K, HW), dtype=torch.uint8, device=dev)
        nfg = torch.zeros((K,), dtype=torch.int32, device=dev)
        q = torch.randn((K, Q, C), generator=g).to(dev)
        kv = rnd(g, (K, HW, 3 * C), dev=dev)
        y, y2 = (torch.zeros((K, Q, C), dtype=F32, device=dev) for _ in range(2))
        ol = O.OpList()
        ol.aux_mask(lg, fg, nfg, K=K, HW=HW)
        ol.attn_q2p(q, kv, fg, nfg, y, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C)
        ol.attn_q2p(q, kv, None, None, y2, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=lg)
        return ol, {'y': y, 'y2': y2}
    hip, ref = run_both(build_fused)
    check(hip, ref, name=f'q2p fused aux mask {mode}', rtol=3e-3)
    assert torch.equal(hip['y'], hip['y2'])


def test_attn_self_and_p2q():
    def build(dev, g):
        K, Q, HW, C, heads = 3, 16, 700, 256, 8
        qk = torch.randn((K, Q, 2 * C), generator=g).to(dev)
        v = torch.randn((K, Q, C), generator=g).to(dev)
        y = torch.zeros((K, Q, C), dtype=F32, device=dev)
        qp = rnd(g, (K, HW, 3 * C), dev=dev)
        kq = torch.randn((K, Q, C), generator=g).to(dev)
        y2 = torch.zeros((K, HW, C), dtype=BF16, device=dev)
        # the same operands packed the way the merged projections produce them: [q | k | v] rows and [k | v] rows
        qkv = torch.cat([qk.cpu(), v.cpu()], -1).contiguous().to(dev)
        kv = torch.cat([kq.cpu(), v.cpu()], -1).contiguous().to(dev)
        y3 = torch.zeros((K, Q, C), dtype=F32, device=dev)
        y4 = torch.zeros((K, HW, C), dtype=BF16, device=dev)
        ol = O.OpList()
        ol.attn_self(qk, v, y, K=K, Q=Q, C=C, heads=heads)
        ol.attn_p2q(qp.view(-1)[2 * C:], kq, v, y2, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C)
        ol.attn_self(qkv, qkv.view(-1)[2 * C:], y3, K=K, Q=Q, C=C, heads=heads, ldqk=3 * C, ldv=3 * C)
        ol.attn_p2q(qp.view(-1)[2 * C:], kv, kv.view(-1)[C:], y4, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C, ldkv=2 * C)
        return ol, {'y': y, 'y2': y2, 'y3': y3, 'y4': y4}
    hip, ref = run_both(build)
    check(hip, ref, name='attn self/p2q', rtol=None)
    assert torch.equal(hip['y'], hip['y3']) and torch.equal(hip['y2'], hip['y4'])


@pytest.mark.parametrize('K', [1, 3])
def test_attention_with_fused_projections(K):
    """The three attentions of a transformer block with their small projections computed inside the launch (LayerNorm + query
    embedding + packed in-projection on MFMA from LDS-staged rows) against the LINEAR + attention pairs they replace: same results to
    fp32 rounding, and both against the interpreter."""
    def build(dev, g):
        Q, HW, C, heads = 16, 1620, 256, 8
        M = K * Q
        x = torch.randn((M, C), generator=g).to(dev)
        emb = (torch.randn((M, C), generator=g) * 0.5).to(dev)
        gam, bet = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
        mk = lambda n: pack_linear(torch.randn((n, C), generator=g) / 16, torch.randn(n, generator=g) * 0.1, dev)
        Wq, Wqkv, Wkv = mk(C), mk(3 * C), mk(2 * C)
        lg = _aux_inputs(g, K, HW, 'mixed').to(dev)
        kvq = rnd(g, (K, HW, 3 * C), dev=dev)
        z = lambda *shape, dt=F32: torch.zeros(shape, dtype=dt, device=dev)
        ol = O.OpList()
        ol.keep += [Wq.weight, Wqkv.weight, Wkv.weight]
        out = {}
        # read_from_pixel
        qp, xn, att = z(M, C), z(M, C), z(M, C)
        ol.linear(x, Wq, qp, M=M, x_add=emb, add_rows=M, ln=(gam, bet), ln_out=xn)
        ol.attn_q2p(qp, kvq, None, None, att, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=lg)
        xn_f, att_f = z(M, C), z(M, C)
        ol.attn_q2p(None, kvq, None, None, att_f, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=lg,
                    proj=dict(x=x, W=Wq, emb=emb, ln=(gam, bet), ln_out=xn_f))
        out.update(xn=xn, att=att, xn_f=xn_f, att_f=att_f)
        # self attention
        qkv, y, sa = z(M, 3 * C), z(M, C), z(M, C)
        ol.linear(x, Wqkv, qkv, M=M, x_add=emb, add_rows=M, add_cols=2 * C, ln=(gam, bet), ln_out=y)
        ol.attn_self(qkv, qkv.view(-1)[2 * C:], sa, K=K, Q=Q, C=C, heads=heads, ldqk=3 * C, ldv=3 * C)
        y_f, sa_f = z(M, C), z(M, C)
        ol.attn_self(None, None, sa_f, K=K, Q=Q, C=C, heads=heads, proj=dict(x=x, W=Wqkv, emb=emb, ln=(gam, bet), ln_out=y_f))
        out.update(y=y, sa=sa, y_f=y_f, sa_f=sa_f)
        # read_from_query
        kv2, pa, pa_f = z(M, 2 * C), z(K, HW, C, dt=BF16), z(K, HW, C, dt=BF16)
        ol.linear(x, Wkv, kv2, M=M, x_add=emb, add_rows=M, add_cols=C)
        ol.attn_p2q(kvq.view(-1)[2 * C:], kv2, kv2.view(-1)[C:], pa, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C, ldkv=2 * C)
        ol.attn_p2q(kvq.view(-1)[2 * C:], None, None, pa_f, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C, proj=dict(x=x, W=Wkv, emb=emb))
        out.update(pa=pa, pa_f=pa_f)
        return ol, out
    hip, ref = run_both(build, seed=77)
    check(hip, ref, name='fused projections', rtol=3e-3)
    for a, b, tol in (('xn', 'xn_f', 1e-5), ('att', 'att_f', 2e-3), ('y', 'y_f', 1e-5), ('sa', 'sa_f', 2e-3)):
        d = float((hip[a] - hip[b]).abs().max())
        assert d <= tol * max(1.0, float(hip[a].abs().max())), (a, d)
    assert float((hip['pa'].float() - hip['pa_f'].float()).abs().max()) <= 2e-2 * float(hip['pa'].float().abs().max())


@pytest.mark.parametrize('K,HW,hid_slice,qnext,inter', [(1, 1620, 64, 1, 0), (3, 1620, 64, 1, 1), (3, 1620, 128, 0, 0), (2, 700, 64, 1, 1), (2, 8040, 64, 1, 0), (1, 37, 128, 1, 1),
                                                       (9, 300, 64, 1, 0), (3, 1620, 64, 0, 1), (5, 2500, 64, 1, 1)])
def test_query_chain_in_four_launches(K, HW, hid_slice, qnext, inter):
    """The query side of a transformer block as the frame runs it (csrc/qchain.hip: ATTN_Q2P with its out-projection summed into a
    fixed-point accumulator -> ATTN_SELF adding it, + out-projection -> QFFN -> ATTN_P2Q) against the seven-launch sequence it replaces
    (LINEAR out-projections, linear1, linear2), with a second block behind it so that ATTN_Q2P's accumulator input is covered as well;
    both against the interpreter.  HW = 8040 (1080p): the pixel loop of ATTN_Q2P runs past its prefetched chunks; HW = 37: ragged single
    chunk; K = 9: the mask logits are read late (more than 8 objects); qnext: the second block's ATTN_Q2P gets its queries from the
    first block's ATTN_P2Q launch (extra blocks) instead of projecting them itself; inter: k | v of the pixels interleaved per head
    (head stride 64, v 32 behind k) as the frame's pixel projection leaves them."""
    def build(dev, g):
        Q, C, heads, FF = 16, 256, 8, 2048
        M = K * Q
        x0 = torch.randn((M, C), generator=g).to(dev)
        emb = (torch.randn((M, C), generator=g) * 0.5).to(dev)
        lnp = lambda: ((torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev))
        mk = lambda n, kd=C: pack_linear(torch.randn((n, kd), generator=g) / (kd ** 0.5), torch.randn(n, generator=g) * 0.1, dev)
        z = lambda *shape, dt=F32: torch.zeros(shape, dtype=dt, device=dev)
        lg = _aux_inputs(g, K, HW, 'mixed').to(dev)
        ol = O.OpList()
        out = {}
        blocks = []
        for b in range(2):
            blocks.append(dict(Wq=mk(C), Wo1=mk(C), Wqkv=mk(3 * C), Wo2=mk(C), W1=mk(FF), W2=mk(C, FF), Wkv=mk(2 * C),
                               ln1=lnp(), ln2=lnp(), ln3=lnp(), kvq=rnd(g, (K, HW, 3 * C), dev=dev)))
            ol.keep += [v.weight for v in blocks[-1].values() if hasattr(v, 'weight')]
        # seven launches per block
        x = x0
        for b, B in enumerate(blocks):
            qp, xn, att, x1 = z(M, C), z(M, C), z(M, C), z(M, C)
            ol.attn_q2p(None, B['kvq'], None, None, att, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=lg,
                        proj=dict(x=x, W=B['Wq'], emb=emb, ln=B['ln1'], ln_out=xn))
            ol.linear(att, B['Wo1'], x1, M=M, res=xn)
            y, sa, x2 = z(M, C), z(M, C), z(M, C)
            ol.attn_self(None, None, sa, K=K, Q=Q, C=C, heads=heads, proj=dict(x=x1, W=B['Wqkv'], emb=emb, ln=B['ln2'], ln_out=y))
            ol.linear(sa, B['Wo2'], x2, M=M, res=y)
            hid, x3 = z(M, FF), z(M, C)
            ol.linear(x2, B['W1'], hid, M=M, relu=True, ln=B['ln3'])
            ol.linear(hid, B['W2'], x3, M=M, res=x2)
            pa = z(K, HW, C, dt=BF16)
            ol.attn_p2q(B['kvq'].view(-1)[2 * C:], None, None, pa, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C, proj=dict(x=x3, W=B['Wkv'], emb=emb))
            out.update({f'x2_{b}': x2, f'pa_{b}': pa, f'y_{b}': y})
            x = x3
        # four launches per block; the products that mix blocks travel as fixed-point accumulators (cleared here by the allocation,
        # in the frame by QUERY_INIT)
        x, acc = x0, None
        zi = lambda: torch.zeros((M, C), dtype=torch.int64, device=dev)
        q_pre = xn_pre = None
        for b, B in enumerate(blocks):
            xn, y, x2 = (xn_pre if xn_pre is not None else z(M, C)), z(M, C), z(M, C)
            a1, a2, a3 = zi(), zi(), zi()
            kvc, lay = B['kvq'], dict(voff=C)
            if inter:                                  # [k | v | q2] -> [k_0 v_0 | k_1 v_1 | ... | q2]
                t = B['kvq'].cpu()
                kv2 = torch.stack([t[..., :C].reshape(K, HW, heads, 32), t[..., C:2 * C].reshape(K, HW, heads, 32)], 3).reshape(K, HW, 2 * C)
                kvc, lay = torch.cat([kv2, t[..., 2 * C:]], -1).contiguous().to(dev), dict(voff=32, hstride=64)
            if q_pre is not None:
                ol.attn_q2p(None, kvc, None, None, None, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, logits=lg, q_pre=q_pre, out_proj=(B['Wo1'], a1), **lay)
            else:
                ol.attn_q2p(None, kvc, None, None, None, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, logits=lg,
                            proj=dict(x=x, W=B['Wq'], emb=emb, ln=B['ln1'], ln_out=xn), acc_in=acc, out_proj=(B['Wo1'], a1), **lay)
            ol.attn_self(None, None, None, K=K, Q=Q, C=C, heads=heads, proj=dict(x=xn, W=B['Wqkv'], emb=emb, ln=B['ln2'], ln_out=y),
                         acc_in=(a1, B['Wo1'].bias), out_proj=(B['Wo2'], a2))
            ol.qffn(y, x2, a3, rows=M, ln=B['ln3'], W1=B['W1'], W2=B['W2'], acc_in=(a2, B['Wo2'].bias), hid_slice=hid_slice)
            acc = (a3, B['W2'].bias)
            pa = z(K, HW, C, dt=BF16)
            next_q = None
            q_pre = xn_pre = None
            if qnext and b + 1 < len(blocks):
                q_pre, xn_pre = z(M, C), z(M, C)
                next_q = dict(ln=blocks[b + 1]['ln1'], W=blocks[b + 1]['Wq'], q_out=q_pre, xn_out=xn_pre)
                out.update({f'qpre_{b + 1}': q_pre, f'xnpre_{b + 1}': xn_pre})
            ol.attn_p2q(B['kvq'].view(-1)[2 * C:], None, None, pa, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C,
                        proj=dict(x=x2, W=B['Wkv'], emb=emb), acc_in=acc, next_q=next_q)
            out.update({f'x2c_{b}': x2, f'pac_{b}': pa, f'yc_{b}': y})
            x = x2
        return ol, out
    hip, ref = run_both(build, seed=31 + K)
    f32k = [k_ for k_ in ref if ref[k_].dtype == torch.float32]
    check({k_: hip[k_] for k_ in f32k}, {k_: ref[k_] for k_ in f32k}, name='query chain', rtol=4e-3)
    check({k_: hip[k_] for k_ in ref if k_ not in f32k}, {k_: ref[k_] for k_ in ref if k_ not in f32k}, name='query chain (bf16 outputs)')
    for b in range(2):
        for a_, c_ in ((f'x2_{b}', f'x2c_{b}'), (f'y_{b}', f'yc_{b}')):
            d = float((hip[a_] - hip[c_]).abs().max())
            assert d <= 2e-3 * max(1.0, float(hip[a_].abs().max())), (a_, d)
        assert float((hip[f'pa_{b}'].float() - hip[f'pac_{b}'].float()).abs().max()) <= 2e-2 * float(hip[f'pa_{b}'].float().abs().max())


@pytest.mark.parametrize('K,HW,qnext,tile', [(1, 1620, 1, 110), (3, 1620, 1, 63), (2, 700, 0, 100), (1, 37, 1, 102), (5, 2500, 1, 110), (2, 8040, 0, 105), (9, 300, 1, 65)])
def test_p2q_with_the_output_projection_inside(K, HW, qnext, tile):
    """ATTN_P2Q flags&32 (csrc/qchain.hip: p2q_out_kernel -- pixel + out_proj(attention) in the attention launch) against the two
    launches it replaces, ATTN_P2Q (chain form) and the 1x1 conv with the residual on a tile of the 'stream' K-order class: bit-identical
    pixels and, with qnext, bit-identical projected queries for the next block; both forms against the interpreter."""
    from cutie_amd.model.weights import linear_as_conv, out_proj_blob
    if not _lib.has_diag_kernels():
        pytest.skip('p2q_out_kernel is a measured-and-lost variant: only in the diagnostic library (make -C cutie_amd/csrc DIAG=1)')
    assert O.korder_class(tile) == 'stream'

    def build(dev, g):
        Q, C, heads = 16, 256, 8
        M = K * Q
        x = torch.randn((M, C), generator=g).to(dev)
        emb = (torch.randn((M, C), generator=g) * 0.5).to(dev)
        acc = torch.round(torch.randn((M, C), generator=g).double() * 0.3 * 4294967296.0).to(torch.int64).to(dev)
        abias = (torch.randn(C, generator=g) * 0.1).to(dev)
        mk = lambda n, kd=C: pack_linear(torch.randn((n, kd), generator=g) / (kd ** 0.5), torch.randn(n, generator=g) * 0.1, dev)
        Wkv, Wq = mk(2 * C), mk(C)
        ln = ((torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev))
        Wo = linear_as_conv(torch.randn((C, C), generator=g) / 16, torch.randn(C, generator=g) * 0.1, dev)
        blob = out_proj_blob(Wo)
        kvq = rnd(g, (K, HW, 3 * C), dev=dev)
        pixel = rnd(g, (K, HW, C), dev=dev)
        z = lambda *shape, dt=F32: torch.zeros(shape, dtype=dt, device=dev)
        ol = O.OpList()
        ol.keep += [Wkv.weight, Wq.weight, Wo.weight, blob]
        out = {}
        for form in ('two', 'one'):
            nq = None
            if qnext:
                out['q_' + form], out['xn_' + form] = z(M, C), z(M, C)
                nq = dict(ln=ln, W=Wq, q_out=out['q_' + form], xn_out=out['xn_' + form])
            pf = z(K, HW, C, dt=BF16)
            kw = dict(K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C, proj=dict(x=x, W=Wkv, emb=emb), acc_in=(acc, abias), next_q=nq)
            if form == 'two':
                pa = z(K, HW, C, dt=BF16)
                ol.attn_p2q(kvq.view(-1)[2 * C:], None, None, pa, **kw)
                ol.conv(pa, Wo, pf, B=K, H=1, W=HW, C1=C, ldx1=C, OH=1, OW=HW, ldy=C, res=pixel, ldr=C, tile=tile)
            else:
                ol.attn_p2q(kvq.view(-1)[2 * C:], None, None, pf, out=dict(Wo=blob, res=pixel), **kw)
            out['pf_' + form] = pf
        return ol, out
    hip, ref = run_both(build, seed=57 + K)
    check({k_: hip[k_] for k_ in ref if ref[k_].dtype == torch.float32}, {k_: ref[k_] for k_ in ref if ref[k_].dtype == torch.float32}, name='p2q + out', rtol=4e-3)
    check({k_: hip[k_] for k_ in ref if ref[k_].dtype != torch.float32}, {k_: ref[k_] for k_ in ref if ref[k_].dtype != torch.float32}, name='p2q + out (bf16 outputs)')
    assert torch.equal(hip['pf_one'].view(torch.int16), hip['pf_two'].view(torch.int16)), int((hip['pf_one'].view(torch.int16) != hip['pf_two'].view(torch.int16)).sum())
    if qnext:
        assert torch.equal(hip['q_one'], hip['q_two']) and torch.equal(hip['xn_one'], hip['xn_two'])


def test_query_chain_rejects_bad_forms():
    """flags 4 / 8 without the fused projection, QFFN with a ragged hidden size: refused by the library, nothing launched."""
    ex = _lib.HipExecutor()
    ol = O.OpList()
    y = torch.zeros((16, 256), device='cuda')
    ol.add(O.ATTN_SELF, 4, [1, 16, 256, 8, 0, 0], [], [y, y, y])
    with pytest.raises(RuntimeError, match="chain form"):
        ex.run(ol.finalize())
    ol = O.OpList()
    ol.add(O.QFFN, 0, [16, 300, 64], [], [y, y, y, y, y, y, y, y])
    with pytest.raises(RuntimeError, match="qffn"):
        ex.run(ol.finalize())


@pytest.mark.parametrize('mode', ['mixed', 'nofg', 'allfg'])
@pytest.mark.parametrize('qpre', [0, 1])
def test_q2p_chain_mask_modes(mode, qpre):
    """Chain form of ATTN_Q2P on the degenerate masks (an object without foreground: its foreground queries attend everywhere; an
    object that is foreground everywhere: its background queries do); qpre: with the queries handed in already projected."""
    def build(dev, g):
        K, Q, HW, C, heads = 3, 16, 1620, 256, 8
        M = K * Q
        lg = _aux_inputs(g, K, HW, mode).to(dev)
        x = torch.randn((M, C), generator=g).to(dev)
        emb = (torch.randn((M, C), generator=g) * 0.5).to(dev)
        gam, bet = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
        Wq = pack_linear(torch.randn((C, C), generator=g) / 16, torch.randn(C, generator=g) * 0.1, dev)
        Wo = pack_linear(torch.randn((C, C), generator=g) / 16, torch.randn(C, generator=g) * 0.1, dev)
        kvq = rnd(g, (K, HW, 3 * C), dev=dev)
        xn = torch.zeros((M, C), dtype=F32, device=dev)
        att = torch.zeros((M, C), dtype=F32, device=dev)
        x1 = torch.zeros((M, C), dtype=F32, device=dev)
        acc = torch.zeros((M, C), dtype=torch.int64, device=dev)
        ol = O.OpList()
        ol.keep += [Wq.weight, Wo.weight]
        if qpre:                                       # the projection by a LINEAR launch, scaled the way ATTN_P2Q's extra blocks leave it
            qp, qs = torch.zeros((M, C), dtype=F32, device=dev), torch.zeros((M, C), dtype=F32, device=dev)
            ol.linear(x, Wq, qp, M=M, x_add=emb, add_rows=M, ln=(gam, bet), ln_out=xn)
            ol.axpy(qp, qs, n=M * C, a=1.0 / 32 ** 0.5)
            ol.attn_q2p(None, kvq, None, None, None, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=lg, q_pre=qs, out_proj=(Wo, acc))
        else:
            ol.attn_q2p(None, kvq, None, None, None, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=lg,
                        proj=dict(x=x, W=Wq, emb=emb, ln=(gam, bet), ln_out=xn), out_proj=(Wo, acc))
        # the unfused pair for comparison: attention output, then the LINEAR (without its bias: the accumulator does not hold it)
        ol.attn_q2p(None, kvq, None, None, att, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=lg,
                    proj=dict(x=x, W=Wq, emb=emb, ln=(gam, bet), ln_out=None))
        return ol, {'acc': acc, 'att': att, 'xn': xn, '_Wo': Wo.weight}
    hip, ref = run_both(build, seed=5)
    fa, fb = hip['acc'].double() / O.OpList.QACC_SCALE, ref['acc'].double() / O.OpList.QACC_SCALE
    assert float((fa - fb).abs().max()) <= 3e-3 * max(1.0, float(fb.abs().max())), float((fa - fb).abs().max())
    # and against the kernel's own attention output pushed through Wo on the host
    want = hip['att'].double() @ hip['_Wo'].double()[:256, :256].t()
    assert float((fa - want).abs().max()) <= 3e-3 * max(1.0, float(want.abs().max()))
    check({'xn': hip['xn'], 'att': hip['att']}, {'xn': ref['xn'], 'att': ref['att']}, name='q2p chain ' + mode, rtol=3e-3)


@pytest.mark.parametrize('K', [1, 3, 5])
def test_query_init_with_its_linears(K):
    def build(dev, g):
        Q, C = 16, 256
        M = K * Q
        om = (torch.rand((M, C + 1), generator=g) + 0.1).to(dev)
        mk = lambda: pack_linear(torch.randn((C, C), generator=g) / 16, torch.randn(C, generator=g) * 0.1, dev)
        Wi, We = mk(), mk()
        ri, re = torch.randn((M, C), generator=g).to(dev), torch.randn((M, C), generator=g).to(dev)
        z = lambda: torch.zeros((M, C), dtype=F32, device=dev)
        vals, q1, e1, q2, e2 = z(), z(), z(), z(), z()
        ol = O.OpList()
        ol.keep += [Wi.weight, We.weight]
        ol.query_init(om, vals, rows=M, C=C)
        ol.linear(vals, Wi, q1, M=M, res=ri)
        ol.linear(vals, We, e1, M=M, res=re)
        ol.query_init2(om, q2, e2, rows=M, w_init=Wi, res_init=ri, w_emb=We, res_emb=re)
        return ol, {'q1': q1, 'e1': e1, 'q2': q2, 'e2': e2}
    hip, ref = run_both(build, seed=5)
    check(hip, ref, name='query_init2', rtol=2e-3)
    assert float((hip['q1'] - hip['q2']).abs().max()) < 1e-4 and float((hip['e1'] - hip['e2']).abs().max()) < 1e-4


def test_summarize_add_pe():
    def build(dev, g):
        K, HW, C, Q = 3, 1620, 256, 16
        feat = rnd(g, (K, HW, C), dev=dev)
        wl = torch.randn((K, HW, Q), generator=g).to(dev)
        m16 = torch.rand((K, HW), generator=g).to(dev)
        y = torch.zeros((K, Q, C + 1), dtype=F32, device=dev)
        pe = rnd(g, (HW * C,), dev=dev)
        z = torch.zeros((K, HW * C), dtype=BF16, device=dev)
        ol = O.OpList()
        ol.summarize(feat, wl, m16, y, K=K, HW=HW, C=C, Q=Q)
        ol.add_pe(feat, pe, z, B=K, n=HW * C)
        return ol, {'y': y, 'z': z}
    check(*run_both(build), name='summarize/add_pe', rtol=2e-3)


# ---- affinity pipeline ---------------------------------------------------------------------------------------
def _affinity_build(HW, ranges, slots, K, top_k, with_usage, dup=False, skip=True, nq=None, cluster=False, dma=None):
    def build(dev, g):
        CV, cap = 256, 1024
        HWp = -(-HW // 64) * 64
        mkey = torch.randn((slots, 64), generator=g) * 0.8
        if cluster:                                # runs of 16 nearly identical tokens: a query's best TILES are full of candidates
            mkey = mkey[torch.arange(slots) // 16] + torch.randn((slots, 64), generator=g) * 1e-5
        if dup:                                    # duplicated memory frames -> exact score ties
            half = slots // 2
            mkey[half:2 * half] = mkey[:half]
        mshr = torch.rand((slots,), generator=g) * 2 + 1
        qkey = torch.randn((HW, 64), generator=g) * 0.8
        qsel = torch.rand((HW, 64), generator=g)
        mkey, mshr, qkey, qsel = mkey.to(dev), mshr.to(dev), qkey.to(dev), qsel.to(dev)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        Ahi, Alo, scale = z((slots + 16, 128), BF16), z((slots + 16, 128), BF16), z((slots + 16,), F32)
        Bhi, Blo, cq = z((HWp, 128), BF16), z((HWp, 128), BF16), z((HWp,), F32)
        G = sum(-(-n // 16) for _, n in ranges if n > 0)
        Gld = -(-G // 64) * 64
        gbuf = z((HWp * Gld + HWp,), F32)                  # pass-0 maxima with the thresholds right behind them (pass 1 skips tiles)
        gmax, tau = gbuf[:HWp * Gld].view(HWp, Gld), gbuf[HWp * Gld:HWp * Gld + HW]
        cval, cidx, count, ovf = z((HW, cap), F32), z((HW, cap), torch.int32), z((HW * 32,), torch.int32), z((1,), torch.int32)
        vals = [rnd(g, (slots + 16, CV), dev=dev) for _ in range(K)]
        vptrs = torch.tensor([v.data_ptr() for v in vals], dtype=torch.int64).to(dev)
        usage = z((slots + 16,), F32) if with_usage else None
        y = z((K, HW, CV), BF16)
        ol = O.OpList()
        ol.keep += vals
        ol.key_prep(mkey, mshr, Ahi, Alo, scale, n=slots, query=False)
        ol.key_prep(qkey, qsel, Bhi, Blo, cq, n=HW, query=True)
        life = torch.arange(slots + 16, dtype=F32).to(dev)
        common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=cap)
        if skip:                                   # the frame's form: stale counters, cleared by the selection launch, which also ticks
            count += 5
        else:
            ol.memset32(count, HW * 32, 0)
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, gmax, None, None, None, mode=0, nq=nq, dma=dma, **common)
        if skip:
            ol.aff_select(gmax, tau, HW=HW, HWp=HWp, G=G, top_k=top_k, clear_count=count,
                          ticks=[(life[8:], slots // 2), (life, 5)])
        else:
            ol.aff_select(gmax, tau, HW=HW, HWp=HWp, G=G, top_k=top_k)
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, tau, cval, cidx, count, mode=1, gmax_precedes_tau=skip, nq=nq, dma=dma, **common)
        ol.aff_readout(cval, cidx, count, vptrs, usage, y, ovf, HW=HW, cap=cap, top_k=top_k, K=K, CV=CV)
        outs = {'Ahi': Ahi, 'Alo': Alo, 'scale': scale, 'Bhi': Bhi, 'Blo': Blo, 'cq': cq, 'tau': tau, 'y': y, 'ovf': ovf,
                'gmax': gmax[:HW, :G], 'count': count.view(HW, 32)[:, 0], 'life': life}
        if with_usage:
            outs['usage'] = usage
        # dense fp32 reference of the reference algorithm (memory_utils.py) for the oracle-level check
        outs['_mkey'], outs['_mshr'], outs['_qkey'], outs['_qsel'] = mkey, mshr, qkey, qsel
        for i, v in enumerate(vals):
            outs[f'_v{i}'] = v
        return ol, outs
    return build


@pytest.mark.parametrize('case', [
    dict(HW=1620, ranges=[(0, 1620)], slots=1620, K=3, top_k=30, usage=False),
    dict(HW=1620, ranges=[(0, 300), (1000, 1620), (3000, 4000)], slots=7100, K=2, top_k=30, usage=True),
    dict(HW=48, ranges=[(0, 48)], slots=48, K=3, top_k=30, usage=True),               # G < top_k
    dict(HW=100, ranges=[(0, 1003)], slots=1003, K=1, top_k=5, usage=False),           # ragged tail tile
    dict(HW=1620, ranges=[(0, 2000), (2100, 1620), (4000, 8097)], slots=12200, K=3, top_k=30, usage=True),   # the bench's size class; half-empty last 256-query block
    dict(HW=700, ranges=[(16, 37), (64, 5)], slots=100, K=2, top_k=30, usage=False),   # three tiles, two of them ragged (G < top_k)
])
@pytest.mark.parametrize('skip', [True, False])
@pytest.mark.parametrize('nq', [1, 2, 4, 12])          # 12: 2 sets per wave on the LDS-DMA kernel
def test_affinity_pipeline(case, skip, nq):
    build = _affinity_build(case['HW'], case['ranges'], case['slots'], case['K'], case['top_k'], case['usage'], skip=skip, nq=nq % 10, dma=nq >= 10)
    hip, ref = run_both(build, seed=7)
    exact = ['Ahi', 'Alo', 'Bhi', 'Blo']
    for k in exact:
        assert torch.equal(hip[k].view(torch.int16), ref[k].view(torch.int16)), k
    G = sum(-(-n // 16) for _, n in case['ranges'])
    keys = ('scale', 'cq', 'gmax') + (('tau',) if G >= case['top_k'] else ())
    check({k: hip[k] for k in keys}, {k: ref[k] for k in keys}, 'aff', rtol=1e-5)
    if G < case['top_k']:
        assert bool(torch.isinf(hip['tau']).all()) and bool((hip['tau'] < 0).all())      # "take everything"
    assert int(hip['ovf']) == 0
    assert torch.equal(hip['life'], ref['life'])
    check({'y': hip['y']}, {'y': ref['y']}, 'aff readout')
    if case['usage']:
        check({'usage': hip['usage']}, {'usage': ref['usage']}, 'aff usage', rtol=1e-4)
    # and against the reference algorithm in dense fp32 (what the oracle computes)
    from oracle.net import get_similarity, topk_softmax
    HW, K = case['HW'], case['K']
    slots = torch.cat([torch.arange(s, s + n) for s, n in case['ranges']])
    sim = get_similarity(hip['_mkey'][slots].t().float(), hip['_mshr'][slots].float(), hip['_qkey'].t().float(), hip['_qsel'].t().float())
    aff, usage = topk_softmax(sim, case['top_k'])
    for o in range(K):
        dense = (hip[f'_v{o}'][slots].float().t() @ aff).t()             # [HW,CV]
        err = float((hip['y'][o].float() - dense).abs().max())
        assert err < 2e-2 * float(dense.abs().max()), ('dense oracle', o, err)
    if case['usage']:
        u = torch.zeros_like(hip['usage'])
        u[slots] = usage
        assert float((hip['usage'] - u).abs().max()) < 1e-3


@pytest.mark.parametrize('case', [
    dict(HW=1620, F=5, ranges=[(0, 2000), (2100, 1620), (4000, 8097)], slots=12200, K=3, top_k=30),     # a memory cycle of the bench clip: 5 x 1664 query rows
    dict(HW=1620, F=2, ranges=[(0, 1620)], slots=1620, K=1, top_k=30),
    dict(HW=100, F=3, ranges=[(0, 1003)], slots=1003, K=2, top_k=5),                                   # ragged tail tile, 128 rows per frame
    dict(HW=48, F=4, ranges=[(0, 48)], slots=48, K=3, top_k=30),                                       # G < top_k: "take everything"
])
@pytest.mark.parametrize('nq', [2, 4])
def test_affinity_batched_frames_match_one_frame_plans(case, nq):
    """One read-out per bank version (MemoryManager._affinity_batch): the stacked query operands of F frames through ONE
    score / select / score / read-out sequence against the F one-frame sequences of the frame's plan -- identical thresholds, candidate
    counts and read-out bits per frame, per-frame usage side buffers (cleared by the selection launch) equal up to the order of the
    float atomics; and both against the descriptor interpreter."""
    HW, F, ranges, slots, K, top_k = case['HW'], case['F'], case['ranges'], case['slots'], case['K'], case['top_k']

    def build(dev, g):
        CV, cap = 256, 1024
        HWp = -(-HW // 64) * 64
        mkey = (torch.randn((slots, 64), generator=g) * 0.8).to(dev)
        mshr = (torch.rand((slots,), generator=g) * 2 + 1).to(dev)
        qkey = (torch.randn((F, HW, 64), generator=g) * 0.8).to(dev)
        qsel = torch.rand((F, HW, 64), generator=g).to(dev)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        Ahi, Alo, scale = z((slots + 16, 128), BF16), z((slots + 16, 128), BF16), z((slots + 16,), F32)
        Bhi, Blo, cq = z((F, HWp, 128), BF16), z((F, HWp, 128), BF16), z((F, HWp), F32)
        G = sum(-(-n // 16) for _, n in ranges if n > 0)
        Gld = -(-G // 64) * 64
        vals = [rnd(g, (slots + 16, CV), dev=dev) for _ in range(K)]
        vptrs = torch.tensor([v.data_ptr() for v in vals], dtype=torch.int64).to(dev)
        nslots = slots + 16
        ovf = z((1,), torch.int32)
        ol = O.OpList()
        ol.keep += vals
        ol.key_prep(mkey, mshr, Ahi, Alo, scale, n=slots, query=False)
        for f in range(F):
            ol.key_prep(qkey[f], qsel[f], Bhi[f], Blo[f], cq[f], n=HW, query=True)
        # (a) frame by frame, as MemoryManager._affinity(ahead=True) issues it
        y1, use1 = z((F, K, HW, CV), BF16), z((F, nslots), F32) + 3.0
        tau1, cnt1 = z((F, HW), F32), z((F, HW), torch.int32)
        keep = []
        for f in range(F):
            gbuf = z((HWp * Gld + HWp,), F32)
            gmax, tau = gbuf[:HWp * Gld], gbuf[HWp * Gld:]
            cval, cidx, count = z((HW, cap), F32), z((HW, cap), torch.int32), z((HW * 32,), torch.int32) + 7
            keep += [gbuf, cval, cidx, count]
            common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=cap, nq=nq)
            ol.aff_score(Ahi, Alo, scale, Bhi[f], Blo[f], cq[f], gmax, None, None, None, mode=0, **common)
            ol.aff_select(gmax, tau, HW=HW, HWp=HWp, G=G, top_k=top_k, clear_count=count, zero=(use1[f], nslots))
            ol.aff_score(Ahi, Alo, scale, Bhi[f], Blo[f], cq[f], tau, cval, cidx, count, mode=1, gmax_precedes_tau=True, **common)
            ol.aff_readout(cval, cidx, count, vptrs, use1[f], y1[f], ovf, HW=HW, cap=cap, top_k=top_k, K=K, CV=CV)
            ol.copy2d(tau, tau1[f], rows=1, rowbytes=4 * HW, src_stride=4 * HW, dst_stride=4 * HW)
            ol.copy2d(count, cnt1[f], rows=HW, rowbytes=4, src_stride=128, dst_stride=4)
        # (b) the F frames stacked
        rows = F * HWp
        yb, useb = z((F, K, HW, CV), BF16), z((F, nslots), F32) + 3.0
        gbuf = z((rows * Gld + rows,), F32)
        gmax, tau = gbuf[:rows * Gld], gbuf[rows * Gld:]
        cval, cidx, count = z((rows, cap), F32), z((rows, cap), torch.int32), z((rows * 32,), torch.int32) + 7
        common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=cap, nq=nq, frames=F)
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, gmax, None, None, None, mode=0, **common)
        ol.aff_select(gmax, tau, HW=HW, HWp=HWp, G=G, top_k=top_k, clear_count=count, zero=(useb, F * nslots), frames=F)
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, tau, cval, cidx, count, mode=1, gmax_precedes_tau=True, **common)
        ol.aff_readout(cval, cidx, count, vptrs, useb, yb, ovf, HW=HW, cap=cap, top_k=top_k, K=K, CV=CV, frames=F, HWp=HWp, usage_stride=nslots)
        ol.keep += keep + [gbuf, cval, cidx, count]
        return ol, {'y1': y1, 'yb': yb, 'use1': use1, 'useb': useb, 'tau1': tau1, 'taub': tau.view(F, HWp)[:, :HW],
                    'cnt1': cnt1, 'cntb': count.view(F, HWp, 32)[:, :HW, 0], 'cnt_pad': count.view(F, HWp, 32)[:, HW:, 0], 'ovf': ovf}

    hip, ref = run_both(build, seed=5)
    assert torch.equal(hip['yb'].view(torch.int16), hip['y1'].view(torch.int16))
    assert torch.equal(hip['taub'], hip['tau1'])
    assert torch.equal(hip['cntb'], hip['cnt1'])
    assert int(hip['cnt_pad'].abs().max() if hip['cnt_pad'].numel() else 0) == 0      # padding rows: counters cleared, nothing appended
    assert int(hip['ovf']) == 0
    assert torch.allclose(hip['useb'], hip['use1'], rtol=1e-5, atol=1e-6)
    assert float(hip['useb'].sum()) > 0.99 * case['F'] * case['HW']                   # (every query's weights sum to one; the + 3.0 was cleared)
    # against the interpreter: the one-frame plans are held to it by test_affinity_pipeline; at 5 x 1620 random queries a borderline
    # top-k choice (MFMA against torch fp32 summation order) flips one element in a few runs, so only the small cases are compared
    if case['HW'] <= 100:
        check({'yb': hip['yb']}, {'yb': ref['yb']}, 'aff batched')
        check({'useb': hip['useb']}, {'useb': ref['useb']}, 'aff batched usage', rtol=1e-4)


def test_affinity_usage_in_fixed_point_does_not_depend_on_the_launch_shape():
    """AFF_READOUT flags&1 (what the product runs): usage accumulated in unsigned 64-bit fixed point (2^-40).  Integer atomics commute: the F
    one-frame read-outs and the stacked one give the SAME counters bit for bit (the f32 form agrees to ~1e-6 only: its last bits follow the
    arrival order of the blocks -- which, through a near-tie of the consolidation's usage ranking, let a clip in lock step leave its own
    run after a few hundred frames, tools/lockstep_soak.py); USAGE_TICK flags&1 / &2 adds them to fp32 counters with one rounding and clears them;
    against the f32 form and the interpreter within float accuracy."""
    HW, F, ranges, slots, K, top_k = 1620, 4, [(0, 2000), (2100, 1620), (4000, 8097)], 12200, 2, 30

    def build(dev, g):
        CV, cap = 256, 1024
        HWp = -(-HW // 64) * 64
        mkey = (torch.randn((slots, 64), generator=g) * 0.8).to(dev)
        mkey = (mkey[torch.arange(slots) // 8] + torch.randn((slots, 64), generator=g).to(dev) * 1e-3)      # clustered: popular tokens, long lists
        mshr = (torch.rand((slots,), generator=g) * 2 + 1).to(dev)
        qkey = (torch.randn((F, HW, 64), generator=g) * 0.8).to(dev)
        qsel = torch.rand((F, HW, 64), generator=g).to(dev)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        nslots = slots + 16
        Ahi, Alo, scale = z((nslots, 128), BF16), z((nslots, 128), BF16), z((nslots,), F32)
        Bhi, Blo, cq = z((F, HWp, 128), BF16), z((F, HWp, 128), BF16), z((F, HWp), F32)
        G = sum(-(-n // 16) for _, n in ranges if n > 0)
        Gld = -(-G // 64) * 64
        vals = [rnd(g, (nslots, CV), dev=dev) for _ in range(K)]
        vptrs = torch.tensor([v.data_ptr() for v in vals], dtype=torch.int64).to(dev)
        ovf = z((1,), torch.int32)
        ol = O.OpList()
        ol.keep += vals
        ol.key_prep(mkey, mshr, Ahi, Alo, scale, n=slots, query=False)
        for f in range(F):
            ol.key_prep(qkey[f], qsel[f], Bhi[f], Blo[f], cq[f], n=HW, query=True)
        y = z((F, K, HW, CV), BF16)
        fx1, f32_1 = z((F, nslots), torch.int64) + 7, z((F, nslots), F32) + 3.0
        keep = []
        for f in range(F):                                      # one frame at a time: fixed-point and f32 counters
            for usage, fx in ((fx1[f], True), (f32_1[f], False)):
                gbuf = z((HWp * Gld + HWp,), F32)
                gmax, tau = gbuf[:HWp * Gld], gbuf[HWp * Gld:]
                cval, cidx, count = z((HW, cap), F32), z((HW, cap), torch.int32), z((HW * 32,), torch.int32)
                keep += [gbuf, cval, cidx, count]
                common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=cap, nq=2)
                ol.aff_score(Ahi, Alo, scale, Bhi[f], Blo[f], cq[f], gmax, None, None, None, mode=0, **common)
                ol.aff_select(gmax, tau, HW=HW, HWp=HWp, G=G, top_k=top_k, clear_count=count, zero=(usage, (2 if fx else 1) * nslots))
                ol.aff_score(Ahi, Alo, scale, Bhi[f], Blo[f], cq[f], tau, cval, cidx, count, mode=1, gmax_precedes_tau=True, **common)
                ol.aff_readout(cval, cidx, count, vptrs, usage, y[f], ovf, HW=HW, cap=cap, top_k=top_k, K=K, CV=CV, usage_fx=fx)
        rows = F * HWp                                          # the F frames stacked, fixed point
        fxb = z((F, nslots), torch.int64) + 7
        gbuf = z((rows * Gld + rows,), F32)
        gmax, tau = gbuf[:rows * Gld], gbuf[rows * Gld:]
        cval, cidx, count = z((rows, cap), F32), z((rows, cap), torch.int32), z((rows * 32,), torch.int32)
        common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=cap, nq=4, frames=F)
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, gmax, None, None, None, mode=0, **common)
        ol.aff_select(gmax, tau, HW=HW, HWp=HWp, G=G, top_k=top_k, clear_count=count, zero=(fxb, 2 * F * nslots), frames=F)
        ol.aff_score(Ahi, Alo, scale, Bhi, Blo, cq, tau, cval, cidx, count, mode=1, gmax_precedes_tau=True, **dict(common, nq=2, dma=True))
        yb = z((F, K, HW, CV), BF16)
        ol.aff_readout(cval, cidx, count, vptrs, fxb, yb, ovf, HW=HW, cap=cap, top_k=top_k, K=K, CV=CV, frames=F, HWp=HWp, usage_stride=nslots, usage_fx=True)
        # the bank's fp32 counters: += every frame's fixed-point sums, one rounding each; the side counters are zero afterwards
        use = torch.arange(nslots, dtype=F32).to(dev) * 0.25
        fxc = fxb.clone() if dev == 'cpu' else None
        fx_copy = z((F, nslots), torch.int64)
        ol.copy2d(fxb, fx_copy, rows=F, rowbytes=8 * nslots, src_stride=8 * nslots, dst_stride=8 * nslots)
        for f in range(F):
            ol.usage_tick(None, 0, None, 0, use=use, delta=fxb[f], n_use=nslots, delta_fx=True, clear_delta=True)
        ol.keep += keep + [gbuf, cval, cidx, count]
        return ol, {'fx1': fx1, 'fxb': fx_copy, 'fx_after': fxb, 'f32': f32_1, 'use': use, 'ovf': ovf, 'y': y, 'yb': yb}

    hip, ref = run_both(build, seed=13)
    assert int(hip['ovf']) == 0
    assert torch.equal(hip['yb'].view(torch.int16), hip['y'].view(torch.int16))
    assert torch.equal(hip['fxb'], hip['fx1'])                                        # bit for bit, whatever the launch shape
    assert int(hip['fx_after'].abs().max()) == 0
    as_f = hip['fxb'].double() * 2.0 ** -40
    assert float(as_f.sum()) > 0.99 * F * HW and float(as_f.max()) > 8.0             # (popular tokens: many queries add to one counter)
    assert torch.allclose(as_f.float(), hip['f32'], rtol=1e-5, atol=1e-5)
    want = torch.arange(hip['use'].shape[0], dtype=torch.float64) * 0.25
    for f in range(F):
        want = (want.float() + as_f[f].float()).double()                              # one rounding per read-out
    assert torch.equal(hip['use'], want.float())
    check({'use': hip['use']}, {'use': ref['use']}, 'usage in fixed point', rtol=1e-4)


@pytest.mark.parametrize('case', [
    dict(HW=1620, F=2, NB=4, ranges=[(0, 2000), (2100, 1620), (4000, 8097)], slots=12200, K=3, top_k=30),   # four bench clips, two frames each
    dict(HW=1620, F=1, NB=2, ranges=[(0, 1620)], slots=1620, K=1, top_k=30),
    dict(HW=100, F=3, NB=3, ranges=[(0, 1003)], slots=1003, K=2, top_k=5),                                 # ragged tail tile, 128 rows per frame
    dict(HW=120, F=2, NB=2, ranges=[(0, 48)], slots=48, K=3, top_k=30),                                    # G < top_k: "take everything"
])
@pytest.mark.parametrize('dma', [False, True])
def test_affinity_frames_of_several_banks_in_one_pass(case, dma):
    """Clips in lock step (ABI 4: AFF_SCORE flags&4 + bank table, AFF_READOUT i8): the stacked frames of NB clips, frame-major (entry
    e = frame * NB + clip reads bank e % NB), through ONE score / select / score / read-out sequence against the one-frame sequence of
    every entry on its own bank -- identical thresholds, candidate counts and read-out bits, per-entry usage buffers equal up to the
    order of the float atomics; the small cases also against the descriptor interpreter."""
    HW, F, NB, ranges, slots, K, top_k = (case[k] for k in ('HW', 'F', 'NB', 'ranges', 'slots', 'K', 'top_k'))
    E = F * NB

    def build(dev, g):
        CV, cap = 256, 1024
        HWp = -(-HW // 64) * 64
        assert HWp % 128 == 0
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        nslots = slots + 16
        G = sum(-(-n // 16) for _, n in ranges if n > 0)
        Gld = -(-G // 64) * 64
        ol = O.OpList()
        banks, vals, vp = [], [], []
        for b in range(NB):
            mkey = (torch.randn((slots, 64), generator=g) * 0.8).to(dev)
            mshr = (torch.rand((slots,), generator=g) * 2 + 1).to(dev)
            Ahi, Alo, scale = z((nslots, 128), BF16), z((nslots, 128), BF16), z((nslots,), F32)
            ol.key_prep(mkey, mshr, Ahi, Alo, scale, n=slots, query=False)
            banks.append((Ahi, Alo, scale))
            vb = [rnd(g, (nslots, CV), dev=dev) for _ in range(K)]
            vals += vb
            vp.append(torch.tensor([v.data_ptr() for v in vb], dtype=torch.int64).to(dev))
        ol.keep += vals + [t for b in banks for t in b] + vp
        table = torch.tensor([[t.data_ptr() for t in b] for b in banks], dtype=torch.int64).to(dev)
        vptrs_all = torch.cat(vp)
        qkey = (torch.randn((E, HW, 64), generator=g) * 0.8).to(dev)
        qsel = torch.rand((E, HW, 64), generator=g).to(dev)
        Bhi, Blo, cq = z((E, HWp, 128), BF16), z((E, HWp, 128), BF16), z((E, HWp), F32)
        for e in range(E):
            ol.key_prep(qkey[e], qsel[e], Bhi[e], Blo[e], cq[e], n=HW, query=True)
        ovf = z((1,), torch.int32)
        # (a) entry by entry on its own bank
        y1, use1 = z((E, K, HW, CV), BF16), z((E, nslots), F32) + 3.0
        tau1, cnt1 = z((E, HW), F32), z((E, HW), torch.int32)
        keep = []
        for e in range(E):
            Ahi, Alo, scale = banks[e % NB]
            gbuf = z((HWp * Gld + HWp,), F32)
            gmax, tau = gbuf[:HWp * Gld], gbuf[HWp * Gld:]
            cval, cidx, count = z((HW, cap), F32), z((HW, cap), torch.int32), z((HW * 32,), torch.int32) + 7
            keep += [gbuf, cval, cidx, count]
            common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=cap, nq=2)
            ol.aff_score(Ahi, Alo, scale, Bhi[e], Blo[e], cq[e], gmax, None, None, None, mode=0, **common)
            ol.aff_select(gmax, tau, HW=HW, HWp=HWp, G=G, top_k=top_k, clear_count=count, zero=(use1[e], nslots))
            ol.aff_score(Ahi, Alo, scale, Bhi[e], Blo[e], cq[e], tau, cval, cidx, count, mode=1, gmax_precedes_tau=True, **common)
            ol.aff_readout(cval, cidx, count, vp[e % NB], use1[e], y1[e], ovf, HW=HW, cap=cap, top_k=top_k, K=K, CV=CV)
            ol.copy2d(tau, tau1[e], rows=1, rowbytes=4 * HW, src_stride=4 * HW, dst_stride=4 * HW)
            ol.copy2d(count, cnt1[e], rows=HW, rowbytes=4, src_stride=128, dst_stride=4)
        # (b) all entries stacked, one bank per entry
        rows = E * HWp
        yb, useb = z((E, K, HW, CV), BF16), z((E, nslots), F32) + 3.0
        gbuf = z((rows * Gld + rows,), F32)
        gmax, tau = gbuf[:rows * Gld], gbuf[rows * Gld:]
        cval, cidx, count = z((rows, cap), F32), z((rows, cap), torch.int32), z((rows * 32,), torch.int32) + 7
        common = dict(HW=HW, HWp=HWp, ranges=ranges, cap=cap, nq=2, dma=dma, frames=E, banks=(table, NB))
        ol.aff_score(None, None, None, Bhi, Blo, cq, gmax, None, None, None, mode=0, **common)
        ol.aff_select(gmax, tau, HW=HW, HWp=HWp, G=G, top_k=top_k, clear_count=count, zero=(useb, E * nslots), frames=E)
        ol.aff_score(None, None, None, Bhi, Blo, cq, tau, cval, cidx, count, mode=1, gmax_precedes_tau=True, **common)
        ol.aff_readout(cval, cidx, count, vptrs_all, useb, yb, ovf, HW=HW, cap=cap, top_k=top_k, K=K, CV=CV, frames=E, HWp=HWp, usage_stride=nslots, banks=NB)
        ol.keep += keep + [gbuf, cval, cidx, count, table, vptrs_all]
        return ol, {'y1': y1, 'yb': yb, 'use1': use1, 'useb': useb, 'tau1': tau1, 'taub': tau.view(E, HWp)[:, :HW],
                    'cnt1': cnt1, 'cntb': count.view(E, HWp, 32)[:, :HW, 0], 'ovf': ovf}

    hip, ref = run_both(build, seed=11)
    assert torch.equal(hip['taub'], hip['tau1'])
    assert torch.equal(hip['cntb'], hip['cnt1'])
    assert torch.equal(hip['yb'].view(torch.int16), hip['y1'].view(torch.int16))
    assert int(hip['ovf']) == 0
    assert torch.allclose(hip['useb'], hip['use1'], rtol=1e-5, atol=1e-6)
    assert float(hip['useb'].sum()) > 0.99 * E * HW
    if HW <= 120:
        check({'yb': hip['yb']}, {'yb': ref['yb']}, 'aff several banks')
        check({'useb': hip['useb']}, {'useb': ref['useb']}, 'aff several banks usage', rtol=1e-4)


@pytest.mark.parametrize('nq', [2, 4])
def test_affinity_long_candidate_lists(nq):
    """Homogeneous memory (runs of 16 nearly identical tokens): the tile-maximum threshold of pass 1 lets hundreds of candidates
    through per query; AFF_READOUT cuts such lists to the entries >= the top_k-th value before it ranks them -- same top-k, same order
    as the interpreter's full sort and as the dense reference."""
    from oracle.net import get_similarity, topk_softmax
    build = _affinity_build(700, [(0, 3000), (3200, 1000)], 4300, 2, 30, True, nq=nq, cluster=True)
    hip, ref = run_both(build, seed=11)
    print('candidates per query: hip max', int(hip['count'].max()), 'mean', float(hip['count'].float().mean()), '| interpreter max', int(ref['count'].max()),
          '| tau finite', bool(torch.isfinite(hip['tau']).all()), 'overflow', int(hip['ovf']))
    assert int(hip['count'].max()) > 64 and int(hip['ovf']) == 0, (int(hip['count'].max()), int(ref['count'].max()))
    check({'y': hip['y']}, {'y': ref['y']}, 'aff long lists')
    check({'usage': hip['usage']}, {'usage': ref['usage']}, 'aff long lists usage', rtol=1e-4)
    slots = torch.cat([torch.arange(0, 3000), torch.arange(3200, 4200)])
    sim = get_similarity(hip['_mkey'][slots].t().float(), hip['_mshr'][slots].float(), hip['_qkey'].t().float(), hip['_qsel'].t().float())
    aff, _ = topk_softmax(sim, 30)
    for o in range(2):
        dense = (hip[f'_v{o}'][slots].float().t() @ aff).t()
        assert float((hip['y'][o].float() - dense).abs().max()) < 2e-2 * float(dense.abs().max())


def test_affinity_exact_ties_are_deterministic():
    """Duplicated memory tokens give exactly tied scores; ties resolve to the lower slot, like the interpreter."""
    build = _affinity_build(200, [(0, 800)], 800, 1, 30, False, dup=True)
    hip, ref = run_both(build, seed=3)
    check({'y': hip['y']}, {'y': ref['y']}, 'aff ties')


# ---- bank / long-term kernels ---------------------------------------------------------------------------------
def test_bank_misc():
    def build(dev, g):
        n = 3000
        src = torch.randn((n, 64), generator=g).to(dev)
        dst = torch.zeros((n, 80), dtype=F32, device=dev)
        a = torch.randn((n,), generator=g).to(dev)
        b = torch.randn((n,), generator=g).to(dev)
        life = torch.rand((n,), generator=g).to(dev)
        ms = torch.zeros((n,), dtype=torch.int32, device=dev)
        c1 = torch.randn((n,), generator=g).to(dev)
        c2 = torch.zeros((n,), dtype=BF16, device=dev)
        c3 = torch.zeros((n,), dtype=F32, device=dev)
        ol = O.OpList()
        ol.copy2d(src, dst, rows=n, rowbytes=256, src_stride=256, dst_stride=320)
        ol.axpy(a, b, n=n, a=0.5)
        ol.usage_tick(life, n)
        ol.memset32(ms, n, 869711765)
        ol.cast(c1, c2, n=n)
        ol.cast(c2, c3, n=n, to_f32=True)
        return ol, {'dst': dst, 'b': b, 'life': life, 'ms': ms, 'c2': c2, 'c3': c3}
    check(*run_both(build), name='bank misc', rtol=1e-6)


@pytest.mark.parametrize('n,k', [(5000, 700), (8100, 128), (9872, 7872), (270, 16), (17, 17)])
def test_rank_select_gather(n, k):
    """torch.topk(usage, k) order (ties -> lower index) by split all-pairs counting; row gathers through the order, by GATHER_ROWS and
    as side jobs of the scattering launch; the cleared side buffer."""
    def build(dev, g):
        use = torch.rand((n,), generator=g)
        use[::7] = 0.0                                       # ties
        life = torch.rand((n,), generator=g) + 0.5
        life[::7] = 1.0
        use, life = use.to(dev), life.to(dev)
        order = torch.zeros((k,), dtype=torch.int32, device=dev)
        src = torch.randn((n, 64), generator=g).to(dev)
        src2 = torch.randn((n, 8), generator=g).to(dev)
        dst = torch.zeros((k, 64), dtype=F32, device=dev)
        side1, side2 = torch.zeros((k, 64), dtype=F32, device=dev), torch.zeros((k, 8), dtype=F32, device=dev)
        cleared = torch.full((37,), 5, dtype=torch.int32, device=dev)
        ol = O.OpList()
        ol.rank_select(use, life, order, n=n, k=k, gathers=[(src, side1, 256), (src2, side2, 32)], zero=(cleared, 30))
        ol.gather_rows(src, order, dst, k=k, rowbytes=256, src_stride=256, dst_stride=256)
        return ol, {'order': order, 'dst': dst, 'side1': side1, 'side2': side2, 'cleared': cleared, '_use': use, '_life': life}
    hip, ref = run_both(build)
    for name in ('order', 'dst', 'side1', 'side2', 'cleared'):
        assert torch.equal(hip[name], ref[name]), name
    assert torch.equal(hip['side1'], hip['dst'])
    u = hip['_use'] / hip['_life']
    assert torch.equal(u[hip['order'].long()], torch.topk(u, k).values)      # the reference's torch.topk values, in its order


@pytest.mark.parametrize('n,P,K', [(2000, 128, 2), (8100, 128, 3), (270, 16, 1), (1000, 136, 1)])
def test_consolidation_kernels(n, P, K):
    """CONSOL_AFF + CONSOL_READ (memory_manager.py:347-356): similarities on fp32 MFMA, softmax over the candidates, prototypes of every
    object's values (bf16 MFMA on split weights) and of the shrinkage -- against the interpreter's dense fp32 form."""
    def build(dev, g):
        C, src, dst = 256, 40, 3
        ck = (torch.randn((n, 64), generator=g) * 0.8).to(dev)
        cs = (torch.rand((n,), generator=g) * 2 + 1).to(dev)
        pk = (torch.randn((P, 64), generator=g) * 0.8).to(dev)
        pk[:P // 2] = ck.cpu()[torch.arange(P // 2) * (n // P)].to(dev)          # prototypes ARE candidates (as in a consolidation): peaked columns
        pe = torch.rand((P, 64), generator=g).to(dev)
        ldS = O.OpList.consol_lds(n)
        S = torch.full((P, ldS), 7.0, dtype=F32, device=dev)
        colmax = torch.zeros((P,), dtype=torch.int32, device=dev)
        banks = [rnd(g, (src + n + 8, C), dev=dev) for _ in range(K)]
        before = [b_.clone() for b_ in banks]
        vptrs = torch.tensor([b_.data_ptr() for b_ in banks], dtype=torch.int64).to(dev)
        part = torch.zeros((O.OpList.consol_scratch_floats(n, P, C, K),), dtype=F32, device=dev)
        oshr = torch.zeros((P,), dtype=F32, device=dev)
        ol = O.OpList()
        ol.keep += banks
        ol.consol_aff(ck, cs, pk, pe, S, colmax, n=n, P=P)
        ol.consol_read(S, colmax, vptrs, cs, part, oshr, n=n, P=P, C=C, K=K, src=src, dst=dst)
        outs = {'S': S[:, :n], 'pad': S[:, n:], 'oshr': oshr}
        for o in range(K):
            outs[f'proto{o}'] = banks[o][dst:dst + P]
            outs[f'rest{o}'] = banks[o][dst + P:]
            outs[f'_rest_before{o}'] = before[o][dst + P:]
        return ol, outs
    hip, ref = run_both(build)
    assert bool(torch.isinf(hip['pad']).all()) and bool((hip['pad'] < 0).all())
    check({'S': hip['S']}, {'S': ref['S']}, 'consol sim', rtol=2e-5)
    check({'oshr': hip['oshr']}, {'oshr': ref['oshr']}, 'consol shrinkage', rtol=1e-4)
    for o in range(K):
        check({'p': hip[f'proto{o}']}, {'p': ref[f'proto{o}']}, f'consol values {o}', rtol=6e-3)      # bf16 results: half an ulp of rounding
        assert torch.equal(hip[f'rest{o}'].view(torch.int16), hip[f'_rest_before{o}'].view(torch.int16))      # nothing else in the bank is touched


@pytest.mark.parametrize('dt', [torch.uint8, torch.int32, torch.int64])
def test_prob_to_id(dt):
    """argmax + id remap on the un-padded, strided view that InferenceCore.step returns; ties -> first plane."""
    def build(dev, g):
        P, Hp, Wp, H, W = 4, 48, 64, 45, 59
        full = torch.rand((P, Hp, Wp), generator=g)
        full[1, 5:9] = full[2, 5:9]                               # exact ties between planes 1 and 2
        full = full.to(dev)
        prob = full[:, 2:2 + H, 3:3 + W]
        lut = torch.tensor([0, 7, 3, 200], dtype=torch.int32).to(dev)
        out = torch.zeros((H, W), dtype=dt, device=dev)
        ol = O.OpList()
        ol.prob_to_id(prob, lut, out, P=P, H=H, W=W, plane=prob.stride(0), ldrow=prob.stride(1))
        return ol, {'out': out}
    hip, ref = run_both(build)
    assert torch.equal(hip['out'], ref['out'])


@pytest.mark.parametrize('nearest', [False, True])
@pytest.mark.parametrize('shape', [((3, 37, 53), (24, 35)), ((4, 30, 54), (480, 854)), ((1, 97, 61), (48, 31))])
def test_resize(shape, nearest):
    """F.interpolate(size=...) bilinear align_corners=False / nearest-exact, up- and down-sampling, strided source view."""
    (C, H, W), (OH, OW) = shape
    def build(dev, g):
        full = torch.rand((C, H + 3, W + 5), generator=g).to(dev)
        src = full[:, 1:1 + H, 2:2 + W]
        out = torch.zeros((C, OH, OW), dtype=F32, device=dev)
        ol = O.OpList()
        ol.resize(src, out, C=C, H=H, W=W, OH=OH, OW=OW, plane=src.stride(0), ldrow=src.stride(1), nearest=nearest)
        return ol, {'out': out}
    hip, ref = run_both(build)
    if nearest:
        assert torch.equal(hip['out'], ref['out'])
    else:
        check(hip, ref, name='resize', rtol=1e-5)


def test_flip_w():
    def build(dev, g):
        src = torch.rand((3, 17, 45), generator=g).to(dev)
        dst = torch.zeros((3, 17, 45), dtype=F32, device=dev)
        acc = torch.rand((4, 20, 64), generator=g).to(dev)
        other = torch.rand((4, 20, 64), generator=g).to(dev)
        ol = O.OpList()
        ol.flip_w(src, dst, rows=3 * 17, W=45)
        ol.flip_w(other, acc, rows=4 * 20, W=64, alpha=0.5, beta=0.5)          # average of a pass and a flipped pass
        return ol, {'dst': dst, 'acc': acc}
    hip, ref = run_both(build)
    assert torch.equal(hip['dst'], ref['dst'])
    check(hip, ref, name='flip_w', rtol=1e-6)


@pytest.mark.parametrize('K,h,w', [(3, 120, 216), (1, 24, 32), (5, 8, 12), (2, 4, 4), (7, 4, 28), (3, 272, 480)])
def test_up4_softmax_with_mask_down(K, h, w):
    """UP4_SOFTMAX flags&4: the launch also writes MASK_DOWN(prob[1:], r = 16) -- bit-identical to the MASK_DOWN launch on the stored
    probabilities (same per-lane sums, same wave reduction), probabilities identical to the plain form."""
    g = torch.Generator().manual_seed(5)
    lg = (torch.randn((K, h, w), generator=g) * 3).cuda()
    H, W = 4 * h, 4 * w
    assert H % 16 == 0 and W % 16 == 0
    hw16 = (H // 16) * (W // 16)
    prob_a, prob_b = torch.zeros((K + 1, H, W), device='cuda'), torch.zeros((K + 1, H, W), device='cuda')
    pair_a, pair_b = torch.zeros((K, hw16, 64), dtype=BF16, device='cuda'), torch.zeros((K, hw16, 64), dtype=BF16, device='cuda')
    m_a, m_b = torch.zeros((K, hw16), device='cuda'), torch.zeros((K, hw16), device='cuda')
    ol = O.OpList()
    ol.up4_softmax(lg, prob_a, None, P=K + 1, h=h, w=w, from_logits=True, mask_down=(m_a, pair_a, 64))
    ol.up4_softmax(lg, prob_b, None, P=K + 1, h=h, w=w, from_logits=True)
    ol.mask_down(prob_b[1:], pair_b, m_b, K=K, H=H, W=W, pair_channels=64)
    ol.run()
    torch.cuda.synchronize()
    assert torch.equal(prob_a, prob_b)
    assert torch.equal(m_a, m_b)
    assert torch.equal(pair_a.view(torch.int16), pair_b.view(torch.int16))


# ---- round 4: loads that used to come in one after the other (tools/isa_waits.py).  Every change keeps the arithmetic and its order,
# ---- so the old form (still selectable) and an exact host emulation must agree to the bit.
@pytest.mark.parametrize('K', [1, 2, 3, 5, 7])
def test_up4_softmax_compile_time_object_count(K, monkeypatch):
    """UP4_SOFTMAX: one kernel instantiation per object count (all 6 x K source logits in flight together) against the kernels with a
    run-time K (flags&8), plain and with the MASK_DOWN side job."""
    h, w = 12, 20
    g = torch.Generator().manual_seed(11 + K)
    lg = (torch.randn((K, h, w), generator=g) * 3).cuda()
    H, W = 4 * h, 4 * w
    hw16 = (H // 16) * (W // 16)
    res = []
    for rtk, lanes in ((0, 0), (8, 0), (0, 16), (8, 16)):    # lanes = 16: per-lane aggregation instead of the wave's shared 6 x 6 source pixels
        monkeypatch.setattr(O, 'UP4_RTK', rtk)
        monkeypatch.setattr(O, 'UP4_LANES', lanes)
        prob, lup, prob_m = (torch.zeros((K + 1, H, W), device='cuda') for _ in range(3))
        pair, m16 = torch.zeros((K, hw16, 64), dtype=BF16, device='cuda'), torch.zeros((K, hw16), device='cuda')
        ol = O.OpList()
        ol.up4_softmax(lg, prob, lup, P=K + 1, h=h, w=w, from_logits=True)
        ol.up4_softmax(lg, prob_m, None, P=K + 1, h=h, w=w, from_logits=True, mask_down=(m16, pair, 64))
        fl = [int(f) for f in ol.finalize()['flags']]
        assert all((f & 8) == rtk for f in fl) and (fl[1] & 16) == lanes
        ol.run()
        torch.cuda.synchronize()
        res.append((prob, lup, prob_m, pair.view(torch.int16), m16))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)
    assert torch.equal(res[0][0], res[0][2])


def test_area_down3_compile_time_ratio(monkeypatch):
    """AREA_DOWN3 with the pooling ratio as a compile-time constant (r = 2, 4: all taps in flight) against the run-time loops (flags&8)."""
    K, h, w = 3, 30, 54
    g = torch.Generator().manual_seed(3)
    p8 = rnd(g, (K, 2 * h, 2 * w, 256), dev='cuda')
    p4 = rnd(g, (K, 4 * h, 4 * w, 256), dev='cuda')
    lg = (torch.randn((K, 4 * h, 4 * w), generator=g) * 3).cuda()
    CT = 256 + 256 + 64
    outs = []
    for rt in (0, 8):
        monkeypatch.setattr(O, 'AREA_RT', rt)
        cat = torch.zeros((K, h, w, CT), dtype=BF16, device='cuda')
        ol = O.OpList()
        ol.area_down3([dict(x=p8, y=cat, B=K, H=2 * h, W=2 * w, C=256, ldx=256, ldy=CT, r=2),
                       dict(x=p4, y=cat.view(-1)[256:], B=K, H=4 * h, W=4 * w, C=256, ldx=256, ldy=CT, r=4),
                       dict(x=lg, y=cat.view(-1)[512:], B=K, H=4 * h, W=4 * w, C=1, ldx=1, ldy=CT, r=4, f32_in=True, Cz=8)])
        assert (int(ol.finalize()['flags'][0]) & 8) == rt
        ol.run()
        torch.cuda.synchronize()
        outs.append(cat.view(torch.int16).clone())
    assert torch.equal(outs[0], outs[1])
    assert float(outs[0].float().abs().max()) > 0


def test_key_prep_query_constant_is_the_channel_ordered_sum(monkeypatch):
    """KEY_PREP (query side): c_j = sum_i e_i k_i^2 is summed in channel order i = 0..63 with separate multiplies and adds (the row's lanes
    hand the running sum on) -- emulated exactly in numpy float32."""
    g = torch.Generator().manual_seed(9)
    for HW, loop in ((1620, 0), (100, 0), (7, 0), (1620, 2), (33, 2)):          # loop = 2: the one-lane-per-row form (A/B switch)
        monkeypatch.setattr(O, 'KEYPREP_LOOP', loop)
        HWp = -(-HW // 64) * 64
        qkey = torch.randn((HW, 64), generator=g) * 0.8
        qsel = torch.rand((HW, 64), generator=g)
        Bhi, Blo, cq = (torch.zeros((HWp, 128), dtype=BF16, device='cuda'), torch.zeros((HWp, 128), dtype=BF16, device='cuda'),
                        torch.zeros((HWp,), dtype=F32, device='cuda'))
        ol = O.OpList()
        kd, sd = qkey.cuda(), qsel.cuda()
        ol.key_prep(kd, sd, Bhi, Blo, cq, n=HW, query=True)
        ol.run()
        torch.cuda.synchronize()
        k, e = qkey.numpy().astype(np.float32), qsel.numpy().astype(np.float32)
        c = np.zeros((HW,), np.float32)
        for i in range(64):
            c = (c + ((e[:, i] * k[:, i]).astype(np.float32) * k[:, i]).astype(np.float32)).astype(np.float32)
        got = cq.cpu().numpy()
        assert np.array_equal(got[:HW].view(np.int32), c.view(np.int32)), float(np.abs(got[:HW] - c).max())
        assert not got[HW:].any()


def test_summarize_partials_are_added_in_chunk_order():
    """SUMMARIZE's final sum takes the per-chunk partials eight at a time; the order of the additions is the chunk order (13 chunks at
    480p, 64 at 1080p: one and eight rounds), so the result equals a sequential float32 sum of the scratch buffer."""
    g = torch.Generator().manual_seed(4)
    for HW in (1620, 8160, 100):
        K, C, Q = 2, 256, 16
        feat = rnd(g, (K, HW, C), dev='cuda')
        wl = (torch.randn((K, HW, Q), generator=g)).cuda()
        m16 = torch.rand((K, HW), generator=g).cuda()
        nchunk, n = (HW + 127) // 128, Q * (C + 1)
        y = torch.zeros((K, Q, C + 1), device='cuda')
        scratch = torch.zeros((K, nchunk, n), device='cuda')
        ol = O.OpList()
        ol.summarize(feat, wl, m16, y, K=K, HW=HW, C=C, Q=Q, scratch=scratch)
        ol.run()
        torch.cuda.synchronize()
        part = scratch.cpu().numpy()
        acc = np.zeros((K, n), np.float32)
        for c in range(nchunk):
            acc = (acc + part[:, c]).astype(np.float32)
        assert np.array_equal(y.cpu().numpy().reshape(K, n).view(np.int32), acc.view(np.int32))


def test_gru_four_channels_per_thread(monkeypatch):
    """GRU with four channels per thread (16-byte accesses) against the one-channel form (flags&1): same expression per element."""
    g = torch.Generator().manual_seed(21)
    n, C = 4860, 256
    v0 = (torch.randn((n, 3 * C), generator=g) * 2).cuda()
    h0 = torch.randn((n, C), generator=g).cuda()
    outs = []
    for scalar in (0, 1):
        monkeypatch.setattr(O, 'GRU_SCALAR', scalar)
        h, hb = h0.clone(), torch.zeros((n, C), dtype=BF16, device='cuda')
        ol = O.OpList()
        ol.gru(v0, h, hb, n=n, C=C)
        assert (int(ol.finalize()['flags'][0]) & 1) == scalar      # (bit 8 = CUTIE_F_PRIO, set by the builder)
        ol.run()
        torch.cuda.synchronize()
        outs.append((h, hb.view(torch.int16)))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], h0)


@pytest.mark.parametrize('G', [30, 64, 257, 700, 1040, 1290, 1800, 2300, 2900, 3700, 4096])
def test_aff_select_values_per_lane_follow_the_tile_count(G, monkeypatch):
    """AFF_SELECT holds ceil(G / 64) values per lane in steps of 4 (12 sizes) instead of 16 | 32 | 64 (flags&2): the same exact top_k-th
    largest tile maximum, also with ties and with -inf entries."""
    g = torch.Generator().manual_seed(G)
    HW, top_k = 203, 30
    HWp, Gld = -(-HW // 64) * 64, -(-G // 64) * 64
    gm = torch.randn((HWp, Gld), generator=g) * 20
    nt = gm[:, 1::7].shape[1]
    gm[:, 0:7 * nt:7] = gm[:, 1::7]                               # exact ties
    gm[5] = float('-inf'); gm[6, : G // 2] = float('-inf')
    gm = gm.cuda()
    taus = []
    for coarse in (0, 2):
        monkeypatch.setattr(O, 'SELECT_COARSE', coarse)
        tau = torch.full((HW,), 7.0, device='cuda')
        ol = O.OpList()
        ol.aff_select(gm, tau, HW=HW, HWp=HWp, G=G, top_k=top_k)
        assert (int(ol.finalize()['flags'][0]) & 2) == coarse
        ol.run()
        torch.cuda.synchronize()
        taus.append(tau.cpu())
    ref = torch.topk(gm[:HW, :G].float().cpu(), top_k, dim=1).values[:, -1] if G >= top_k else torch.full((HW,), float('-inf'))
    assert torch.equal(taus[0], taus[1])
    assert torch.equal(taus[0], ref)


# ---- round 6 (ABI 4): clips in lock step -- every grouped launch against the per-clip launches it replaces, bit for bit ------------------------
@pytest.mark.parametrize('tile', [None, 2, 66, 68, 100, 103, 105, 110])
@pytest.mark.parametrize('k', [1, 3])
def test_conv_residual_groups(tile, k):
    """CONV with CUTIE_F_RES_BCAST and f0 / f1: the B = G x Kg objects come in G groups, group q adds ITS residual map (f1 rows apart) --
    equal to G launches of Kg objects with a plain broadcast residual each, on every kernel family (igemm, conv_dma, conv_pc)."""
    G, Kg, H, W, C, Cout, frames = 3, 2, 13, 17, 64, 72, 4
    g = _gen(3)
    w = torch.randn(Cout, C, k, k, generator=g) / math.sqrt(C * k * k)
    pc = pack_conv(w, torch.randn(Cout, generator=g) * 0.1, 'cuda')
    x = rnd(g, (G * Kg, H, W, C), dev='cuda')
    res = rnd(g, (G, frames, H, W, Cout), dev='cuda')             # group q's map = res[q, 1]: maps `frames` x H x W rows apart
    y, yref = (torch.zeros((G * Kg, H, W, Cout), dtype=BF16, device='cuda') for _ in range(2))
    ol = O.OpList()
    kw = dict(H=H, W=W, C1=C, ldx1=C, OH=H, OW=W, ldy=Cout, ldr=Cout, res_bcast=True, act=O.ACT_RELU, tile=tile, pad=(k - 1) // 2)
    ol.conv(x, pc, y, B=G * Kg, res=res[0, 1], res_group=(Kg, frames * H * W), **kw)
    for q in range(G):
        ol.conv(x[q * Kg:], pc, yref[q * Kg:], B=Kg, res=res[q, 1], **kw)
    ol.run()
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), yref.view(torch.int16))
    # and the interpreter's reading of the grouped descriptor
    hip, ref = run_both(lambda dev, gg: _grouped_conv(dev, gg, tile, k), seed=9)
    check(hip, ref, 'grouped residual')


def _grouped_conv(dev, g, tile, k):
    G, Kg, H, W, C, Cout = 2, 3, 9, 11, 64, 40
    pc = pack_conv(torch.randn(Cout, C, k, k, generator=g) / math.sqrt(C * k * k), torch.randn(Cout, generator=g) * 0.1, dev)
    x, res = rnd(g, (G * Kg, H, W, C), dev=dev), rnd(g, (G, 2, H, W, Cout), dev=dev)
    y = torch.zeros((G * Kg, H, W, Cout), dtype=BF16, device=dev)
    ol = O.OpList()
    ol.conv(x, pc, y, B=G * Kg, H=H, W=W, C1=C, ldx1=C, OH=H, OW=W, ldy=Cout, ldr=Cout, res=res, res_bcast=True, res_group=(Kg, 2 * H * W), tile=tile, pad=(k - 1) // 2)
    return ol, {'y': y}


def test_upsample2x_add_skip_groups():
    """UPSAMPLE2X_ADD i4 / i5: one skip map per group of objects."""
    G, Kg, h, w, C, frames = 3, 2, 7, 9, 128, 3
    g = _gen(4)
    x = rnd(g, (G * Kg, h, w, C), dev='cuda')
    skip = rnd(g, (G, frames, 2 * h, 2 * w, C), dev='cuda')
    y, yref = (torch.zeros((G * Kg, 2 * h, 2 * w, C), dtype=BF16, device='cuda') for _ in range(2))
    ol = O.OpList()
    ol.upsample2x_add(x, skip[0, 2], y, B=G * Kg, h=h, w=w, C=C, skip_group=(Kg, frames * 4 * h * w))
    for q in range(G):
        ol.upsample2x_add(x[q * Kg:], skip[q, 2], yref[q * Kg:], B=Kg, h=h, w=w, C=C)
    ol.run()
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), yref.view(torch.int16))

    def build(dev, gg):
        xx, sk = rnd(gg, (4, 5, 6, 64), dev=dev), rnd(gg, (2, 10, 12, 64), dev=dev)
        yy = torch.zeros((4, 10, 12, 64), dtype=BF16, device=dev)
        o = O.OpList()
        o.upsample2x_add(xx, sk, yy, B=4, h=5, w=6, C=64, skip_group=(2, 120))
        return o, {'y': yy}
    check(*run_both(build), name='upsample2x_add groups')


@pytest.mark.parametrize('K,h,w,md', [(3, 120, 216, True), (2, 8, 12, True), (3, 24, 32, False), (7, 4, 28, True)])
def test_up4_softmax_clips(K, h, w, md):
    """UP4_SOFTMAX i4: C clips per launch (grid.y) = C one-clip launches, with and without the MASK_DOWN side results."""
    C = 3
    g = _gen(6)
    lg = (torch.randn((C, K, h, w), generator=g) * 3).cuda()
    H, W = 4 * h, 4 * w
    hw16 = (H // 16) * (W // 16)
    mk = lambda: (torch.zeros((C, K + 1, H, W), device='cuda'), torch.zeros((C * K, hw16, 64), dtype=BF16, device='cuda'), torch.zeros((C * K, hw16), device='cuda'))
    (pa, ra, ma), (pb, rb, mb) = mk(), mk()
    ol = O.OpList()
    ol.up4_softmax(lg, pa, None, P=K + 1, h=h, w=w, from_logits=True, clips=C, mask_down=(ma, ra, 64) if md else None)
    for c in range(C):
        ol.up4_softmax(lg[c], pb[c], None, P=K + 1, h=h, w=w, from_logits=True, mask_down=(mb[c * K:], rb[c * K:], 64) if md else None)
    ol.run()
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(ra.view(torch.int16), rb.view(torch.int16))
    if md:                                               # the interpreter's reading of the clips form
        def build(dev, gg):
            l2 = (torch.randn((2, 2, 8, 8), generator=gg) * 3).to(dev)
            p2 = torch.zeros((2, 3, 32, 32), device=dev)
            m2, r2 = torch.zeros((4, 4), device=dev), torch.zeros((4, 4, 8), dtype=BF16, device=dev)
            o = O.OpList()
            o.up4_softmax(l2, p2, None, P=3, h=8, w=8, from_logits=True, clips=2, mask_down=(m2, r2, 8))
            return o, {'prob': p2, 'm16': m2, 'pair': r2}
        check(*run_both(build), name='up4 clips', rtol=2e-3)


@pytest.mark.parametrize('qpre', [0, 1])
@pytest.mark.parametrize('Kg', [1, 3, 5])
def test_q2p_chain_clip_objects(Kg, qpre):
    """ATTN_Q2P i9 (chain form): the foreground masks are decided among the objects of ONE clip -- a launch over G clips of Kg objects adds
    to the accumulator exactly what G launches of Kg objects add."""
    G, Q, HW, C, heads = 2, 16, 700, 256, 8
    K = G * Kg
    M = K * Q
    g = _gen(11)
    lg = torch.cat([_aux_inputs(g, Kg, HW, 'mixed') for _ in range(G)], 0).cuda()
    x, emb = torch.randn((M, C), generator=g).cuda(), (torch.randn((M, C), generator=g) * 0.5).cuda()
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    Wq = pack_linear(torch.randn((C, C), generator=g) / 16, torch.randn(C, generator=g) * 0.1, 'cuda')
    Wo = pack_linear(torch.randn((C, C), generator=g) / 16, torch.randn(C, generator=g) * 0.1, 'cuda')
    kvq = rnd(g, (K, HW, 3 * C), dev='cuda')
    qs = (torch.randn((M, C), generator=g) * 0.2).cuda()
    acc_a, acc_b = (torch.zeros((M, C), dtype=torch.int64, device='cuda') for _ in range(2))
    xn_a, xn_b = (torch.zeros((M, C), dtype=F32, device='cuda') for _ in range(2))
    ol = O.OpList()
    ol.keep += [Wq.weight, Wo.weight]
    common = dict(Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C)

    def launch(k0, k1, acc, xn, clip_objects):
        if qpre:
            ol.attn_q2p(None, kvq[k0:], None, None, None, K=k1 - k0, logits=lg[k0:], q_pre=qs[k0 * Q:], out_proj=(Wo, acc[k0 * Q:]), clip_objects=clip_objects, **common)
        else:
            ol.attn_q2p(None, kvq[k0:], None, None, None, K=k1 - k0, logits=lg[k0:], proj=dict(x=x[k0 * Q:], W=Wq, emb=emb[k0 * Q:], ln=(gam, bet), ln_out=xn[k0 * Q:]),
                        out_proj=(Wo, acc[k0 * Q:]), clip_objects=clip_objects, **common)
    launch(0, K, acc_a, xn_a, Kg)
    for q in range(G):
        launch(q * Kg, (q + 1) * Kg, acc_b, xn_b, None)
    ol.run()
    torch.cuda.synchronize()
    assert torch.equal(acc_a, acc_b) and torch.equal(xn_a, xn_b)
    # ... and NOT what one launch over all K objects as ONE clip adds (the masks differ as soon as a clip has a second object)
    if Kg > 1:
        acc_c = torch.zeros_like(acc_a)
        ol2 = O.OpList()
        ol2.keep += [Wq.weight, Wo.weight]
        ol2.attn_q2p(None, kvq, None, None, None, K=K, logits=lg, q_pre=qs, out_proj=(Wo, acc_c), **common) if qpre else \
            ol2.attn_q2p(None, kvq, None, None, None, K=K, logits=lg, proj=dict(x=x, W=Wq, emb=emb, ln=(gam, bet), ln_out=xn_b), out_proj=(Wo, acc_c), **common)
        ol2.run()
        torch.cuda.synchronize()
        assert not torch.equal(acc_a, acc_c)


@pytest.mark.parametrize('geo', [(480, 854, 480, 864, 5, 0, 0, 12), (100, 120, 112, 128, 4, 6, 2, 3), (30, 43, 32, 48, 2, 1, 3, 5)])
def test_stem_several_frames_per_launch(geo):
    """STEM i8 / i9 (ABI 4): the frames of an encoder window (no masks) or the clips of a lock-step group (K masks each, a fixed stride apart)
    in ONE launch -- bit-identical to one launch per frame; and the interpreter's reading of the descriptor."""
    h0, w0, H, W, pl, pt, K, NI = geo
    g = _gen(sum(geo))
    imgs = [torch.rand((3, h0, w0), generator=g).cuda() for _ in range(NI)]
    Kk = max(K, 1)
    masks = None
    if K:
        masks = torch.rand((NI, K + 1, H, W), generator=g)
        masks = (masks * (torch.rand((NI, K + 1, H, W), generator=g) > 0.5)).cuda()      # clip f's object planes at [f, 1:]
    wt = torch.randn((64, 8, 7, 7), generator=g) / math.sqrt(147)
    wt[:, 5 if K else 3:] = 0
    pc = pack_conv(wt, torch.randn(64, generator=g) * 0.1, 'cuda')
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    ya, yb = (torch.zeros((NI * Kk, H // 4, W // 4, 64), dtype=BF16, device='cuda') for _ in range(2))
    kw = dict(h0=h0, w0=w0, H=H, W=W, pad_left=pl, pad_top=pt, K=Kk, mean=mean, std=std, relu=True)
    ol = O.OpList()
    ol.stem(imgs[0], masks[0, 1:] if K else None, pc, ya, more_images=imgs[1:], mask_stride=(K + 1) * H * W, **kw)
    for f in range(NI):
        ol.stem(imgs[f], masks[f, 1:] if K else None, pc, yb[f * Kk:], **kw)
    ol.run()
    torch.cuda.synchronize()
    assert torch.equal(ya.view(torch.int16), yb.view(torch.int16))

    def build(dev, gg):
        im = [torch.rand((3, 20, 30), generator=gg).to(dev) for _ in range(3)]
        mk = torch.rand((3, 3, 32, 32), generator=gg).to(dev)
        w2 = torch.randn((64, 8, 7, 7), generator=gg) / math.sqrt(147)
        w2[:, 5:] = 0
        p2 = pack_conv(w2, torch.randn(64, generator=gg) * 0.1, dev)
        yy = torch.zeros((3 * 2, 8, 8, 64), dtype=BF16, device=dev)
        o = O.OpList()
        o.keep += [p2.weight]
        o.stem(im[0], mk[0, 1:], p2, yy, h0=20, w0=30, H=32, W=32, pad_left=1, pad_top=6, K=2, mean=mean, std=std, more_images=im[1:], mask_stride=3 * 32 * 32)
        return o, {'y': yy}
    check(*run_both(build, seed=3), name='stem frames')
