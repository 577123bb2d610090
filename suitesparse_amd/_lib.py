"""ctypes binding of libgpuqrengine_amd.so (the C ABI in include/gpuqrengine_amd.h).

Plumbing for tests and benchmarks; the product is the shared library.  There
is no fallback: if the library is missing, loading raises.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPUQR_LIB_PATH") or os.path.join(_HERE, "lib", "libgpuqrengine_amd.so")

GPUQR_SUCCESS, GPUQR_OUTOFMEMORY, GPUQR_GPUERROR = 0, 1, 2


class SEntry(C.Structure):
    """gpuqr_sentry: F[findex] = value (16 bytes)."""
    _fields_ = [("findex", C.c_int64), ("value", C.c_double)]


class FrontC(C.Structure):
    """gpuqr_front, field for field."""
    _fields_ = [
        ("fids", C.c_int64), ("pids", C.c_int64), ("fidg", C.c_int64), ("pidg", C.c_int64),
        ("fm", C.c_int64), ("fn", C.c_int64), ("fp", C.c_int64), ("rank", C.c_int64),
        ("is_col_major", C.c_int32), ("pad0", C.c_int32), ("ldn", C.c_int64),
        ("F", C.c_void_p), ("gpuF", C.c_void_p), ("cpuR", C.c_void_p), ("Stair", C.c_void_p),
        ("is_sparse", C.c_int32), ("is_staged", C.c_int32), ("push_only", C.c_int32), ("nc", C.c_int32),
        ("gpuS", C.c_void_p), ("Scount", C.c_int64),
        ("cm", C.c_int32), ("cn", C.c_int32), ("pn", C.c_int32), ("pad1", C.c_int32),
        ("gpuRjmap", C.c_void_p), ("gpuRimap", C.c_void_p), ("gpuC", C.c_void_p),
        ("gpuRpack", C.c_void_p), ("cpuRpack", C.c_void_p),
        ("rhs_col", C.c_int32), ("rhs_n", C.c_int32),
        ("cpuRimap", C.c_void_p),
    ]


class Stats(C.Structure):
    """gpuqr_stats / QREngineStats."""
    _fields_ = [("kernelTime", C.c_float), ("numLaunches", C.c_int64), ("flopsActual", C.c_int64)]


_lib = None


def lib():
    """Load the shared library once; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libgpuqrengine_amd.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`"
            % LIB_PATH)
    # A PyTorch wheel brings its own copy of the HIP runtime (torch/lib/libamdhip64.so), the library is
    # linked against the system's.  Whichever is loaded first serves the whole process, and PyTorch finds
    # no device when it is not its own ("No HIP GPUs are available").  The tests, the benchmark and the
    # multi-GPU driver use both, so PyTorch goes first here whenever it is installed; a C caller, or a
    # Python process that sets GPUQR_NO_TORCH, never loads it.
    if "torch" not in sys.modules and os.environ.get("GPUQR_NO_TORCH") is None:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    L.gpuqr_factor_dense.argtypes = [C.c_size_t, C.POINTER(FrontC), C.c_int64, C.POINTER(Stats)]
    L.gpuqr_factor_dense.restype = C.c_int
    L.gpuqr_factor_sparse.argtypes = [C.c_size_t, C.POINTER(FrontC), C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.gpuqr_factor_sparse.restype = C.c_int
    L.gpuqr_find_staircase.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    L.gpuqr_find_staircase.restype = C.POINTER(C.c_int64)
    L.gpuqr_max_queue_size.argtypes = [C.c_size_t]
    L.gpuqr_max_queue_size.restype = C.c_size_t
    L.gpuqr_front_flops.argtypes = [C.c_int64, C.c_int64, C.c_void_p]
    L.gpuqr_front_flops.restype = C.c_int64
    L.gpuqr_device_count.restype = C.c_int
    L.gpuqr_version.restype = C.c_int
    L.gpuqr_set_mode.argtypes = [C.c_int]
    L.gpuqr_set_mode.restype = None
    _lib = L
    return L
