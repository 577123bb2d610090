"""Python mirror of the reference's engine interface, on top of the C ABI.

Names follow GPUQREngine/Include/GPUQREngine.hpp:17-48 and
GPUQREngine_Front.hpp:20-86 so that tests read like the reference's demo
(GPUQREngine/Demo/gpuqrengine_demo.cpp:224-242).  All numerics happen inside
libgpuqrengine_amd.so on the GPU; nothing here computes.
"""
import ctypes as C

import numpy as np

from . import _lib

QRENGINE_SUCCESS = _lib.GPUQR_SUCCESS
QRENGINE_OUTOFMEMORY = _lib.GPUQR_OUTOFMEMORY
QRENGINE_GPUERROR = _lib.GPUQR_GPUERROR
EMPTY = -1


class QREngineStats:
    """GPUQREngine_Stats.hpp:13-29."""

    def __init__(self):
        self.kernelTime = 0.0
        self.numLaunches = 0
        self.flopsActual = 0


class SparseMeta:
    """GPUQREngine_SparseMeta.hpp:19-80 (fields the engine reads)."""

    def __init__(self):
        self.fp = 0
        self.nc = 0
        self.isStaged = False
        self.pushOnly = False
        self.isSparse = False
        self.gpuS = 0           # device address
        self.Scount = 0
        self.cm = 0
        self.cn = 0
        self.csize = 0
        self.pn = 0
        self.pc = 0
        self.gpuRjmap = 0       # device address
        self.gpuRimap = 0       # device address
        self.gpuC = 0           # device address
        self.cpuRimap = None    # extension: host copy of gpuRimap (numpy int32), see gpuqr_front.cpuRimap


class Front:
    """GPUQREngine_Front.hpp:20-167.  F / cpuR are numpy arrays (host);
    gpuF is a device address (int)."""

    def __init__(self, fids, pids, fm, fn, fp=None, isColMajor=False, ldn=EMPTY):
        self.fids = fids
        self.pids = pids
        self.fidg = EMPTY
        self.pidg = EMPTY
        self.fm = fm
        self.fn = fn
        self.fp = fn if fp is None else fp
        self.rank = min(fm, fn)
        self.isColMajor = isColMajor
        self.ldn = (fm if isColMajor else fn) if ldn == EMPTY else ldn
        self.F = None
        self.gpuF = 0
        self.cpuR = None
        self.Stair = None
        self.sparseMeta = SparseMeta()

    def isDense(self):
        return not self.sparseMeta.isSparse

    def isSparse(self):
        return self.sparseMeta.isSparse

    def isStaged(self):
        return self.isSparse() and self.sparseMeta.isStaged

    def isPushOnly(self):
        return self.isSparse() and self.sparseMeta.pushOnly

    def getNumFrontValues(self):
        return self.fm * self.fn

    def getNumRValues(self):
        return self.rank * self.fn


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return a or None
    return a.ctypes.data


def _pack(fronts, keep):
    arr = (_lib.FrontC * max(len(fronts), 1))()
    for k, f in enumerate(fronts):
        c = arr[k]
        c.fids, c.pids, c.fidg, c.pidg = f.fids, f.pids, f.fidg, f.pidg
        c.fm, c.fn, c.fp, c.rank = f.fm, f.fn, f.fp, f.rank
        c.is_col_major, c.ldn = int(f.isColMajor), f.ldn
        c.F, c.gpuF, c.cpuR = _ptr(f.F), _ptr(f.gpuF), _ptr(f.cpuR)
        if f.Stair is not None:
            st = np.ascontiguousarray(f.Stair, dtype=np.int64)
            keep.append(st)
            c.Stair = st.ctypes.data
        m = f.sparseMeta
        c.is_sparse, c.is_staged, c.push_only, c.nc = int(m.isSparse), int(m.isStaged), int(m.pushOnly), m.nc
        c.gpuS, c.Scount = _ptr(m.gpuS), m.Scount
        c.cm, c.cn, c.pn = m.cm, m.cn, m.pn
        c.gpuRjmap, c.gpuRimap, c.gpuC = _ptr(m.gpuRjmap), _ptr(m.gpuRimap), _ptr(m.gpuC)
        if getattr(m, "cpuRimap", None) is not None:
            ri = np.ascontiguousarray(m.cpuRimap, dtype=np.int32)
            keep.append(ri)
            c.cpuRimap = ri.ctypes.data
    return arr


class FrontArray:
    """A Front[] as the C ABI sees it: the gpuqr_front structs filled once.  Callers that run the same
    batch many times pass this instead of a list of Front objects, as a C caller would keep its array."""

    def __init__(self, fronts):
        self.fronts = list(fronts)
        self.keep = []
        self.arr = _pack(self.fronts, self.keep)
        self.sparse = any(f.isSparse() or f.gpuF for f in self.fronts)

    def __len__(self):
        return len(self.fronts)


def GPUQREngine(gpuMemorySize, userFronts, numFronts=None, Parent=None, Childp=None, Child=None, stats=None):
    """Both overloads of GPUQREngine(...) (GPUQREngine.hpp:26-43).

    Without Parent/Childp/Child the dense overload runs (host F in, cpuR out);
    with them - or when any front is sparse or carries gpuF - the sparse
    overload runs on caller-owned device buffers.
    """
    L = _lib.lib()
    if isinstance(userFronts, FrontArray):
        fronts, arr = userFronts.fronts, userFronts.arr
        if numFronts is None:
            numFronts = len(fronts)
        sparse = Parent is not None or userFronts.sparse
    else:
        fronts = list(userFronts)
        if numFronts is None:
            numFronts = len(fronts)
        fronts = fronts[:numFronts]
        keep = []
        arr = _pack(fronts, keep)
        sparse = Parent is not None or any(f.isSparse() or f.gpuF for f in fronts)
    st = _lib.Stats()
    if sparse:
        code = L.gpuqr_factor_sparse(gpuMemorySize, arr, numFronts, None, None, None, C.byref(st))
    else:
        code = L.gpuqr_factor_dense(gpuMemorySize, arr, numFronts, C.byref(st))
        for k, f in enumerate(fronts[:numFronts]):
            f.rank = arr[k].rank
    if stats is not None:
        stats.kernelTime, stats.numLaunches, stats.flopsActual = st.kernelTime, st.numLaunches, st.flopsActual
    return code


def GPUQREngine_FindStaircase(front):
    """GPUQREngine_ExpertDense.cpp:196-217, on a row-major host front."""
    L = _lib.lib()
    F = np.ascontiguousarray(front.F, dtype=np.float64)
    p = L.gpuqr_find_staircase(F.ctypes.data, front.fm, front.fn)
    if not p:
        return None
    out = np.array([p[j] for j in range(front.fn)], dtype=np.int64)
    C.CDLL(None).free(p)
    return out


def ssgpu_maxQueueSize(gpuMemorySize):
    """ssgpu_maxQueueSize.cpp:6-27."""
    return int(_lib.lib().gpuqr_max_queue_size(gpuMemorySize))
