"""ctypes loader of the CPU oracle (oracle/qr_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


class SEntry(C.Structure):
    _fields_ = [("findex", C.c_int64), ("value", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_max_queue_size.argtypes = [C.c_size_t]
        _lib.oracle_max_queue_size.restype = C.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def max_queue_size(mem):
    return int(lib().oracle_max_queue_size(mem))


def find_staircase(F):
    F = np.ascontiguousarray(F, dtype=np.float64)
    fm, fn = F.shape
    st = np.zeros(fn, dtype=np.int64)
    lib().oracle_find_staircase(_p(F), C.c_int64(fm), C.c_int64(fn), _p(st))
    return st


def front_qr(F, Stair=None):
    """Householder QR of a row-major front; returns the reduced front (fm x fn)."""
    A = np.array(F, dtype=np.float64, order="C", copy=True)
    fm, fn = A.shape
    st = None if Stair is None else np.ascontiguousarray(Stair, dtype=np.int64)
    lib().oracle_front_qr(_p(A), C.c_int64(fm), C.c_int64(fn), None if st is None else _p(st))
    return A
