"""The declared device memory is a promise the caller makes from what the device reports
(SPQR/Demo/qrdemo_gpu.cpp:34-35 declares all of it); the staging arithmetic then prices the
reference's scheduler, not this engine.  A staging that arithmetic calls feasible must run."""
import numpy as np
import pytest

from suitesparse_amd import _lib, multigpu, problems, spqr

pytestmark = pytest.mark.gpu


def _order(A, shape):
    from oracle import ref
    return ref.metis_order_child(A) if ref.available() else problems.nested_dissection(shape, leaf=8)


def test_declared_memory_is_all_the_device_has():
    """Nearly all of the device is taken by somebody else; the caller declares what hipMemGetInfo
    reports as free, as the reference's demo does.  The staging finds one stage feasible in that
    number; the driver cuts the fronts into as many groups as the real room needs and delivers the R
    of the unconstrained run, bit for bit."""
    import torch
    shape = (40, 40, 40)
    A = problems.stacked_ls(shape)
    Q = _order(A, shape)
    csc = multigpu._Csc(A)
    sym0 = spqr.analyze(A, Q, relax=-1, gpuMemorySize=200 * 2**30)
    assert sym0.numStages == 1
    res0 = spqr.factorize(sym0, csc)
    L = _lib.lib()
    L.spqrgpu_amd_finalize()
    L.gpuqr_finalize()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    leave = 3 * 2**30
    ballast = torch.empty(free - leave, dtype=torch.uint8, device="cuda")
    try:
        free2, _ = torch.cuda.mem_get_info()
        assert free2 <= leave + 2**30
        sym = spqr.analyze(A, Q, relax=-1, gpuMemorySize=int(free2))
        assert sym.status == 0 and sym.numStages >= 1
        res = spqr.factorize(sym, csc)
        assert res.numStagesRun > sym.numStages          # the driver had to cut finer than the staging
        assert res.rank == res0.rank
        assert (res.R != res0.R).nnz == 0                # bit for bit
    finally:
        del ballast
        L.spqrgpu_amd_finalize()
        L.gpuqr_finalize()
        torch.cuda.empty_cache()
