"""Timing probe: the sparse path under the SPQR-compatible analysis (CHOLMOD's supernodes) with the
reference's METIS order, beside the stand-in analysis with the geometric order.  Prints one line per
configuration: cold call, warm calls, kernel time, launches."""
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from oracle import ref
from suitesparse_amd import multigpu, problems, spqr

MEM = int(float(os.environ.get("BENCH_DEVICE_GIB", "200")) * 2**30)


def run(shape, mode, reps=int(os.environ.get("PROBE_REPS", "2"))):
    A = problems.stacked_ls(shape)
    t0 = time.perf_counter()
    if mode == "metis":
        Q = ref.metis_order(A)
        relax = -1
    elif mode == "metis-child":
        Q = ref.metis_order_child(A)
        relax = -1
    elif mode == "nd-cholmod":
        Q = problems.nested_dissection(shape, leaf=8)
        relax = -1
    else:
        Q = problems.nested_dissection(shape, leaf=8)
        relax = 20
    t_ord = time.perf_counter() - t0
    t0 = time.perf_counter()
    sym = spqr.analyze(A, Q, relax=relax, gpuMemorySize=MEM)
    t_an = time.perf_counter() - t0
    csc = multigpu._Csc(A)
    times = []
    for i in range(1 + reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = spqr.factorize(sym, csc, want_R=False)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    warm = min(times[1:])
    print("%-14s %-10s nf %6d stages %d TF %8.2f order %5.1fs analyze %5.1fs cold %8.1f ms warm %8.1f ms kernels %8.1f ms launches %6d -> %6.2f TF/s"
          % ("x".join(map(str, shape)), mode, sym.nf, sym.numStages, sym.flopcount / 1e12, t_ord, t_an, times[0] * 1e3, warm * 1e3,
             res.gpuKernelTime, res.gpuNumKernelLaunches, sym.flopcount / warm / 1e12), flush=True)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["1024x1024", "64x64x64", "96x96x96"]
    for s in todo:
        shape = tuple(int(x) for x in s.split("x"))
        for mode in os.environ.get("PROBE_MODES", "standin,nd-cholmod,metis").split(","):
            run(shape, mode)
        from suitesparse_amd import _lib
        _lib.lib().spqrgpu_amd_finalize()
        torch.cuda.empty_cache()
