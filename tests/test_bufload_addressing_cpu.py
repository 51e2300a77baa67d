"""The integer logic of the experimental buffer-load conv kernel (cutie_amd/csrc/conv_bufload.hip: per-thread chunk offsets,
tap validity masks, wave-uniform tap / channel / source state, shifted resource base, out-of-range zero fill, weight offsets,
loop-invariant LDS indices) emulated on the CPU against conv2d -- tools/emulate_bufload_addressing.py holds the emulation and
more cases; the kernel itself is opt-in (CUTIE_AMD_EXPERIMENTAL_TILES=1) until it has been run on a GPU."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location(
    'emulate_bufload_addressing', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools', 'emulate_bufload_addressing.py'))
emu = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(emu)


def test_lds_indices_match_the_default_kernel():
    emu.check_lds_indices()


@pytest.mark.parametrize('case', [
    dict(B=1, H=6, W=7, C1=64, C2=0, Cout=40, k=3, stride=1, pad=1, BM=32, BN=64, BK=64),           # halo, Kpad > K tail tile
    dict(B=1, H=7, W=6, C1=64, C2=0, Cout=72, k=3, stride=2, pad=1, BM=64, BN=128, BK=64),          # stride 2, Cout tail
    dict(B=1, H=5, W=6, C1=64, C2=64, Cout=64, k=1, stride=1, pad=0, ldx1=80, ldx2=72, BM=32, BN=64, BK=64),   # two sources, ld > C
])
def test_addressing_reproduces_conv2d(case):
    err, scale = emu.emulate(**case, NT=256)
    assert err < 1e-4 * max(1.0, scale), (case, err)


def test_eligibility_mirrors_the_launch_checks():
    from cutie_amd import ops as O
    assert O.bufload_tile_ok(50, cin=256, kh=3) and O.bufload_tile_ok(53, cin=256, kh=1)
    assert not O.bufload_tile_ok(50, cin=264, kh=1, c2=8)            # sensory_compress: the 8-channel pair straddles a tile
    assert not O.bufload_tile_ok(53, cin=64, kh=3)                   # BK 128 > Cin
    assert not O.bufload_tile_ok(50, cin=64, kh=7)                   # 49 taps do not fit the validity mask
    assert O.bufload_tile_ok(50, cin=512, kh=3, c2=256) and not O.bufload_tile_ok(54, cin=320, kh=3, c2=64)
    os.environ.pop('CUTIE_AMD_EXPERIMENTAL_TILES', None)
    geom = dict(kh=3, stride=1, pad=1, W=54, c2=0)
    assert not set(O.tile_candidates(4860, 256, 256, 2304, geom=geom)) & set(O.EXPERIMENTAL_TILES)      # off by default
    os.environ['CUTIE_AMD_EXPERIMENTAL_TILES'] = '1'
    try:
        assert {50, 51, 52, 53, 54, 55, 56} <= set(O.tile_candidates(4860, 256, 256, 2304, geom=geom))
        assert O.splitk_candidates(4860, 256, 2304, 50) == [1]
    finally:
        del os.environ['CUTIE_AMD_EXPERIMENTAL_TILES']
