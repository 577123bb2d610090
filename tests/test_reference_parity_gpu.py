"""GPU parity tests against the REFERENCE itself: the reference's CPU SuiteSparseQR (compiled from its
own sources into oracle/_ref, a built file that travels with the tree) runs live on the box's host
cores and the HIP path must reproduce it: the symbolic arrays bit for bit (pattern of R, tree,
front sizes), R = Q'AE to 1e-12 of its largest entry after sign normalisation, and, with SPQR's
default tolerance, the same rank and the same dead columns."""
import numpy as np
import pytest
import scipy.sparse as sp

from suitesparse_amd import spqr
from tests import refcheck as rc

pytestmark = [pytest.mark.gpu, rc.needs_reference]

CASES = [
    ("grid:12x12", rc.ORD_COLAMD, 2**34), ("grid:32x32", rc.ORD_METIS, 2**34), ("grid:72x72", rc.ORD_METIS, 2**34),
    ("grid:72x72", rc.ORD_COLAMD, 0), ("grid:128x128", rc.ORD_AMD, 2**34), ("grid:256x256", rc.ORD_METIS, 2**34),
    ("grid:6x6x6", rc.ORD_COLAMD, 0), ("grid:10x10x10", rc.ORD_METIS, 2**34), ("grid:14x14x14", rc.ORD_METIS, 3 * 2**20),
    ("grid:20x17x9", rc.ORD_AMD, 2**34), ("grid:24x24x24", rc.ORD_METIS, 2**34), ("grid:32x32x32", rc.ORD_METIS, 2**34),
    ("grid:32x32x32", rc.ORD_COLAMD, 2**30),
    ("ash219:", rc.ORD_COLAMD, 2**34), ("rand:300x120x7", rc.ORD_COLAMD, 2**34), ("rand:500x400x3", rc.ORD_AMD, 0),
]


@pytest.mark.parametrize("mname,ordering,mem", CASES, ids=["%s-o%d-%d" % c for c in CASES])
def test_hip_path_reproduces_the_reference(mname, ordering, mem):
    """configs 1, 3 and 4 of the baseline at test sizes, one stage and several (mem = the declared
    device memory; 0 = every front its own stage, the reference's own staging test,
    SPQR/Tcov/qrtest.cpp:2660-2684)."""
    A, r = rc.reference_pruned(rc.matrix(mname), ordering)
    sym = spqr.analyze(A, r.Q, relax=-1, gpuMemorySize=mem)
    rc.check_symbolic(sym, r)
    res = spqr.factorize(sym, A)
    err = rc.check_numeric_pruned(res.R, res.rank, res.Rdead, r)
    assert err <= 1e-12


TOL_CASES = [("rankdef:24x6x6", rc.ORD_COLAMD), ("rankdef:24x6x6", rc.ORD_METIS), ("rankdef:64x16x16", rc.ORD_METIS),
             ("rankdef:40x0x12", rc.ORD_AMD), ("rankdef:256x64x64", rc.ORD_COLAMD), ("rankdef:256x64x64", rc.ORD_METIS),
             ("grid:20x20", rc.ORD_COLAMD)]


@pytest.mark.parametrize("mname,ordering", TOL_CASES, ids=["%s-o%d" % c for c in TOL_CASES])
def test_rank_detection_on_the_device_equals_the_reference(mname, ordering):
    """config 5: duplicated and nearly dependent columns at SPQR_DEFAULT_TOL; the reference's CPU path
    (spqr_front.cpp:424-479, blocked, fchunk 32) decides rank and dead columns, the device must decide
    the same and deliver the same squeezed R."""
    A, r = rc.reference_pruned(rc.matrix(mname), ordering, rc.DEFAULT_TOL)
    sym = spqr.analyze(A, r.Q, relax=-1, gpuMemorySize=2**34, rank_detection=True)
    rc.check_symbolic(sym, r)
    res = spqr.factorize(sym, A, tol=r.tol_used)
    rc.check_numeric_pruned(res.R, res.rank, res.Rdead, r, tol=1e-11)
    assert abs(res.norm_E_fro - r.norm_E_fro) <= 1e-3 * r.tol_used


def test_config3_at_full_size_equals_the_reference():
    """BASELINE config 3 as it is timed: the 2D 1024 x 1024 stacked least-squares matrix, METIS order,
    no tolerance.  The reference factorizes it on the host (its R has 1.4e8 entries); every entry of
    the device's R is compared."""
    A = rc.matrix("grid:1024x1024")
    r = rc.reference(A, rc.ORD_METIS)
    sym = spqr.analyze(A, r.Q, relax=-1, gpuMemorySize=200 * 2**30)
    rc.check_symbolic(sym, r)
    res = spqr.factorize(sym, A)
    err = rc.check_numeric_pruned(res.R, res.rank, res.Rdead, r)
    print("config 3 at full size: rank %d, nnz(R) %d, max | |R| - |R_ref| | / max|R_ref| = %.2e" % (res.rank, res.R.nnz, err))


def test_3d_48_equals_the_reference():
    """The config-4 generator at 48^3 (the largest 3D grid the reference's CPU path finishes in well
    under a minute on the box): METIS order, every entry of R."""
    A = rc.matrix("grid:48x48x48")
    r = rc.reference(A, rc.ORD_METIS)
    sym = spqr.analyze(A, r.Q, relax=-1, gpuMemorySize=200 * 2**30)
    rc.check_symbolic(sym, r)
    res = spqr.factorize(sym, A)
    err = rc.check_numeric_pruned(res.R, res.rank, res.Rdead, r)
    print("3D 48^3: rank %d, nnz(R) %d, max rel diff %.2e" % (res.rank, res.R.nnz, err))
