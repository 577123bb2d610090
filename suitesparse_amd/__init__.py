"""MI355X-native multifrontal QR front-batch engine (drop-in for GPUQREngine / SPQRGPU).

The product is suitesparse_amd/lib/libgpuqrengine_amd.so (HIP kernels + C++ host
engine behind a C ABI, see include/).  This package is the thin Python binding
used by tests and bench.py.
"""
from .engine import (EMPTY, QRENGINE_GPUERROR, QRENGINE_OUTOFMEMORY, QRENGINE_SUCCESS, Front, FrontArray,
                     GPUQREngine, GPUQREngine_FindStaircase, QREngineStats, SparseMeta,
                     ssgpu_maxQueueSize)

__all__ = ["EMPTY", "QRENGINE_GPUERROR", "QRENGINE_OUTOFMEMORY", "QRENGINE_SUCCESS", "Front", "FrontArray",
           "GPUQREngine", "GPUQREngine_FindStaircase", "QREngineStats", "SparseMeta",
           "ssgpu_maxQueueSize"]
