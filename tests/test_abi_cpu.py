"""CPU-side checks of the C-ABI boundary: the shared library loads without a GPU and exports every symbol
declared in include/cutie_hip.h; the Python descriptor mirror (cutie_amd/ops.py) matches the header."""
import ctypes
import os
import re

import numpy as np

from cutie_amd import _lib, ops as O

HEADER = os.path.join(os.path.dirname(__file__), '..', 'include', 'cutie_hip.h')


def test_library_loads_and_exports_declared_symbols():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    src = open(HEADER).read()
    declared = re.findall(r'^\s*(?:int|void\*?|float|const char\*)\s+\*?(cutie_\w+)\s*\(', src, re.M)
    assert set(declared) == set(_lib.EXPORTED_SYMBOLS), (declared, _lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cutie_hip_abi_version() == _lib.ABI_VERSION
    assert lib.cutie_op_struct_size() == O.OP_DTYPE.itemsize == 256


def test_descriptor_layout_matches_header():
    src = open(HEADER).read()
    assert int(re.search(r'#define CUTIE_OP_NI (\d+)', src).group(1)) == O.NI
    assert int(re.search(r'#define CUTIE_OP_NF (\d+)', src).group(1)) == O.NF
    assert int(re.search(r'#define CUTIE_OP_NP (\d+)', src).group(1)) == O.NP
    off = {n: O.OP_DTYPE.fields[n][1] for n in O.OP_DTYPE.names}
    assert off == {'kind': 0, 'flags': 4, 'i': 8, 'f': 8 + 4 * O.NI, 'p': 8 + 4 * O.NI + 4 * O.NF}
    # enum values: first is explicit (= 1), the rest count up
    body = src[src.index('enum {'):src.index('CUTIE_OP__COUNT')]
    names = re.findall(r'^\s*(CUTIE_OP_[A-Z0-9_]+)\s*(?:=\s*(\d+))?\s*,', body, re.M)
    val = 0
    for name, explicit in names:
        val = int(explicit) if explicit else val + 1
        assert getattr(O, name[len('CUTIE_OP_'):]) == val, name
    assert len(names) == 42
    for flag in ('F_RELU_IN', 'F_OUT_F32', 'F_RES_BCAST', 'ACT_SHIFT', 'ACT_RELU', 'ACT_SIGMOID', 'ACT_SQ1'):
        assert int(re.search(r'#define CUTIE_%s\s+(\d+)' % flag, src).group(1)) == getattr(O, flag)


def test_unknown_op_is_rejected_without_a_gpu():
    lib = _lib.load()
    arr = np.zeros(1, dtype=O.OP_DTYPE)
    arr['kind'][0] = 999
    assert lib.cutie_exec(arr.ctypes.data, 1, None) != 0
    assert b'unknown op' in lib.cutie_hip_last_error()


def test_product_has_no_cpu_fallback():
    import pytest, torch
    from cutie_amd.config import default_config
    from cutie_amd.model.cutie import CUTIE
    _lib.set_executor_for_testing(None)
    if torch.cuda.is_available():
        pytest.skip('GPU present: covered by tests/test_gpu_parity.py::test_product_requires_hip_library')
    net = CUTIE(default_config())
    with pytest.raises(_lib.HipLibraryError):
        net.encode_image(torch.zeros(1, 3, 32, 32))
