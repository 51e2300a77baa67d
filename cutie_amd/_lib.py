"""ctypes binding of libcutie_hip.so (C ABI: include/cutie_hip.h).

The product path has NO CPU fallback: if the shared library is missing, stale, or no HIP device is
present, ``get_executor()`` raises.  (tests/mock_exec.py can inject a torch interpreter of the op
descriptors to check the host-side wiring on CPU; that interpreter lives under tests/ only.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CUTIE_AMD_LIB') or os.path.join(_HERE, 'libcutie_hip.so')     # ($CUTIE_AMD_LIB: diagnostic builds, tools/ablate_conv.sh)
ABI_VERSION = 4
OP_STRUCT_SIZE = 256

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C cutie_amd/csrc`).  cutie_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    lib.cutie_exec.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.cutie_exec.restype = ctypes.c_int
    lib.cutie_exec_one.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.cutie_exec_one.restype = ctypes.c_int
    lib.cutie_graph_capture.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.cutie_graph_capture.restype = ctypes.c_void_p
    lib.cutie_graph_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.cutie_graph_launch.restype = ctypes.c_int
    lib.cutie_graph_destroy.argtypes = [ctypes.c_void_p]
    lib.cutie_graph_destroy.restype = None
    lib.cutie_time_ops.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.cutie_time_ops.restype = ctypes.c_float
    lib.cutie_hip_last_error.argtypes = []
    lib.cutie_hip_last_error.restype = ctypes.c_char_p
    lib.cutie_hip_abi_version.restype = ctypes.c_int
    lib.cutie_op_struct_size.restype = ctypes.c_int
    lib.cutie_hip_build_flags.restype = ctypes.c_int
    if lib.cutie_hip_abi_version() != ABI_VERSION or lib.cutie_op_struct_size() != OP_STRUCT_SIZE:
        raise HipLibraryError(f'{LIB_PATH} is stale (ABI {lib.cutie_hip_abi_version()}, '
                              f'op size {lib.cutie_op_struct_size()}); rebuild it.')
    _lib = lib
    return lib


EXPORTED_SYMBOLS = ['cutie_exec', 'cutie_exec_one', 'cutie_graph_capture', 'cutie_graph_launch',
                    'cutie_graph_destroy', 'cutie_time_ops', 'cutie_hip_last_error',
                    'cutie_hip_abi_version', 'cutie_op_struct_size', 'cutie_hip_build_flags']


def has_diag_kernels():
    """Was the loaded library built with -DCUTIE_DIAG (make DIAG=1: measured-and-lost kernel variants present)?"""
    return bool(load().cutie_hip_build_flags() & 1)


class HipExecutor:
    """Runs op-descriptor arrays on the current torch HIP stream."""
    is_mock = False

    def __init__(self):
        import torch
        self.lib = load()
        if not torch.cuda.is_available():
            raise HipLibraryError('no HIP device visible: cutie_amd runs on MI355X only (no CPU fallback).')
        self._torch = torch
        self._raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        self.graph_stats = [0, 0]          # plans run launch by launch | replayed as a graph
        self._warned = False

    def stream(self):
        """The current torch HIP stream of the current device as a raw handle.  (Asked once per cutie_exec call: the private getter
        returns the handle without building a torch.cuda.Stream object -- ~0.3 us instead of ~3 -- and is what torch's own compiled
        code paths call; without it the public API is used.)"""
        raw = self._raw_stream
        if raw is not None:
            return raw(self._torch.cuda.current_device())
        return self._torch.cuda.current_stream().cuda_stream

    def run(self, arr):
        rc = self.lib.cutie_exec(arr.ctypes.data, len(arr), self.stream())
        if rc != 0:
            raise RuntimeError('cutie_exec failed: ' + self.lib.cutie_hip_last_error().decode())

    def run_on(self, arr, stream):
        """`run` on a given torch stream (of the current device) without making it torch's current stream first."""
        rc = self.lib.cutie_exec(arr.ctypes.data, len(arr), stream.cuda_stream)
        if rc != 0:
            raise RuntimeError('cutie_exec failed: ' + self.lib.cutie_hip_last_error().decode())

    # ---- HIP-graph replay of launch plans -------------------------------------------------------------------------------------
    # A plan is ~15-60 launches at 2.5-4.6 us of host time each; the frame's ~160 launches cost the host 0.65-1.1 ms, about as long as
    # the device needs for them, so a second stream (look-ahead encoder, deferred memorising) could not be fed fast enough to overlap
    # (profiles/r03_host.md).  A plan whose bound pointers repeat (the caching allocator hands the same blocks back in steady state)
    # is captured the second time it is seen and replayed as ONE graph launch from then on.
    GRAPH_CACHE_ENTRIES = 24

    def run_cached(self, arr, cache, head=0):
        """cache: dict owned by the plan (pointer signature -> 0 seen once | graph handle | -1 capture failed).  The first `head` launches
        (they read caller-owned pointers that differ every frame, e.g. the image) always run as launches."""
        if head:
            self.run(arr[:head])
            arr = arr[head:]
        key = arr['p'].tobytes()
        ent = cache.get(key)
        if ent is None:
            if len(cache) >= self.GRAPH_CACHE_ENTRIES:
                old = cache.pop(next(iter(cache)))
                if old not in (0, -1):
                    self.lib.cutie_graph_destroy(old)
            cache[key] = 0
            self.graph_stats[0] += 1
            return self.run(arr)
        if ent == 0:
            ent = self.lib.cutie_graph_capture(arr.ctypes.data, len(arr), self.stream()) or -1
            cache[key] = ent
            if ent == -1 and not self._warned:
                self._warned = True
                import warnings
                warnings.warn('HIP graph capture of a launch plan failed (%s); such plans keep running launch by launch'
                              % self.lib.cutie_hip_last_error().decode())
        if ent == -1:
            self.graph_stats[0] += 1
            return self.run(arr)
        self.graph_stats[1] += 1
        if self.lib.cutie_graph_launch(ent, self.stream()) != 0:
            raise RuntimeError('cutie_graph_launch failed: ' + self.lib.cutie_hip_last_error().decode())

    def time_ops(self, arr, iters):
        ms = self.lib.cutie_time_ops(arr.ctypes.data, len(arr), iters, self.stream())
        if ms < 0:
            raise RuntimeError('cutie_time_ops failed: ' + self.lib.cutie_hip_last_error().decode())
        return ms


_executor = None


def get_executor():
    global _executor
    if _executor is None:
        _executor = HipExecutor()
    return _executor


def set_executor_for_testing(ex):
    """Test hook (tests/mock_exec.py): inject an interpreter of the descriptors.  Never used by the product."""
    global _executor
    _executor = ex
