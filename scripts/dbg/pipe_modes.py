"""Compare R of one sparse problem across launch mode / queue mode with and without row-block pipelining."""
import os, sys
os.environ["GPUQR_NO_PLAN_CACHE"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from suitesparse_amd import _lib, problems, spqr
L = _lib.lib()
for grid, leaf in [((40, 40), 4), ((9, 9, 9), 2), ((64, 64), 4)]:
    A = problems.stacked_ls(grid)
    sym = spqr.analyze(A, problems.nested_dissection(grid, leaf=leaf), relax=20, gpuMemorySize=2**34)
    res = {}
    for mode in (0, 1):
        for nopipe in (1, 0):
            if nopipe: os.environ["GPUQR_NO_PIPELINE"] = "1"
            else: os.environ.pop("GPUQR_NO_PIPELINE", None)
            L.gpuqr_set_mode(mode)
            for rep in range(2):
                r = spqr.factorize(sym, A)
                res[(mode, nopipe, rep)] = r.R
    L.gpuqr_set_mode(0)
    ref = res[(0, 1, 0)]
    for k, R in res.items():
        d = abs(R - ref).max() if R.nnz else 0.0
        print(grid, "mode %d nopipe %d rep %d: max diff vs (0,1,0) %.3e  (max |R| %.3e)" % (k + (d, abs(ref).max())), flush=True)
