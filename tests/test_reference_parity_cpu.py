"""CPU tests against the REFERENCE itself: its CPU SuiteSparseQR compiled from its own sources
(oracle/_ref, see oracle/build_ref.sh) is run live on generated matrices, and compared with
  * the symbolic analysis of this repository in its SPQR-compatible mode (every array bit for bit),
  * the CPU oracle (oracle/qr_oracle.c), which is thereby pinned to the reference's numbers:
    R to 1e-12 of its largest entry after sign normalisation, rank and dead columns exactly.
Skipped where the library has not been built (a clone without /root/reference)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from suitesparse_amd import spqr
from tests import refcheck as rc

pytestmark = rc.needs_reference

SYM_CASES = [
    ("grid:12x12", rc.ORD_FIXED), ("grid:12x12", rc.ORD_COLAMD), ("grid:12x12", rc.ORD_AMD), ("grid:12x12", rc.ORD_METIS),
    ("grid:20x9", rc.ORD_AMD), ("grid:32x32", rc.ORD_METIS), ("grid:32x32", rc.ORD_COLAMD),
    ("grid:72x72", rc.ORD_METIS), ("grid:72x72", rc.ORD_COLAMD), ("grid:72x72", rc.ORD_AMD),
    ("grid:6x6x6", rc.ORD_COLAMD), ("grid:6x6x6", rc.ORD_METIS), ("grid:10x10x10", rc.ORD_METIS),
    ("grid:14x14x14", rc.ORD_METIS), ("grid:14x14x14", rc.ORD_COLAMD), ("grid:20x17x9", rc.ORD_AMD),
    ("grid:128x128", rc.ORD_METIS), ("grid:24x24x24", rc.ORD_METIS),
    ("ash219:", rc.ORD_COLAMD), ("ash219:", rc.ORD_FIXED),
    ("rand:300x120x7", rc.ORD_COLAMD), ("rand:500x400x3", rc.ORD_AMD), ("rand:900x700x11", rc.ORD_METIS),
]


@pytest.mark.parametrize("mname,ordering", SYM_CASES, ids=["%s-o%d" % c for c in SYM_CASES])
@pytest.mark.parametrize("rank_detection", [False, True], ids=["notol", "tol"])
def test_symbolic_arrays_equal_the_reference(mname, ordering, rank_detection):
    """Super, Rp, Rj (the pattern of R as SPQR holds it), Parent, Post, Childp, Child, Sp, Sj, Sleft,
    PLinv, Qfill, Fm, Cm and the flop count of spqr_amd_analyze in its SPQR-compatible mode against
    QRsym of the reference's spqr_1factor (SPQR/Source/spqr_1factor.cpp:624-641 -> spqr_analyze.cpp),
    for the column order the reference itself chose with each of its ordering methods."""
    A, r = rc.reference_pruned(rc.matrix(mname), ordering, rc.DEFAULT_TOL if rank_detection else rc.NO_TOL, want_R=False)
    sym = spqr.analyze(A, r.Q, relax=-1, gpuMemorySize=2**34, rank_detection=rank_detection)
    rc.check_symbolic(sym, r)


def test_a_column_order_that_is_no_postorder():
    """SPQR does not postorder a GIVEN order (spqr_analyze.cpp:241-247): a random permutation must give
    the reference's arrays too, with its supernodes cut where the tree jumps."""
    A = rc.matrix("grid:16x16")
    rng = np.random.default_rng(3)
    Q = rng.permutation(A.shape[1])
    r = rc.reference(sp.csc_matrix(A[:, Q]), rc.ORD_FIXED, want_R=False)
    sym = spqr.analyze(A, Q, relax=-1, gpuMemorySize=2**34)
    r.Qfill = Q          # the reference saw the permuted matrix with the identity
    rc.check_symbolic(sym, r)


NUM_CASES = [
    ("grid:12x12", rc.ORD_COLAMD), ("grid:12x12", rc.ORD_METIS), ("grid:20x9", rc.ORD_AMD), ("grid:32x32", rc.ORD_METIS),
    ("grid:6x6x6", rc.ORD_COLAMD), ("grid:10x10x10", rc.ORD_METIS), ("ash219:", rc.ORD_COLAMD),
    ("rand:300x120x7", rc.ORD_COLAMD), ("rand:500x400x3", rc.ORD_AMD),
]


def _oracle_r(sym, A):
    Farena, FOff = oracle.multifrontal(sym, spqr.s_values(sym, A))
    return sp.csc_matrix(oracle.r_from_fronts(sym, Farena, FOff))


@pytest.mark.parametrize("mname,ordering", NUM_CASES, ids=["%s-o%d" % c for c in NUM_CASES])
def test_oracle_r_equals_the_reference(mname, ordering):
    """The oracle's multifrontal factorization (the device's Householder convention, restated on the
    CPU) on the reference's own fronts gives the reference's R = Q'AE: |R| to 1e-12 of its largest
    entry, same rank."""
    A, r = rc.reference_pruned(rc.matrix(mname), ordering)
    sym = spqr.analyze(A, r.Q, relax=-1, gpuMemorySize=2**34)
    R = _oracle_r(sym, A)
    rank = int(sum(min(int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f])) for f in range(sym.nf)))
    rc.check_numeric_pruned(R[:rank], rank, np.zeros(sym.n, np.int8), r)


TOL_CASES = [("rankdef:24x6x6", rc.ORD_COLAMD), ("rankdef:24x6x6", rc.ORD_METIS), ("rankdef:64x16x16", rc.ORD_METIS),
             ("rankdef:40x0x12", rc.ORD_AMD), ("grid:12x12", rc.ORD_COLAMD)]


@pytest.mark.parametrize("mname,ordering", TOL_CASES, ids=["%s-o%d" % c for c in TOL_CASES])
def test_oracle_rank_rule_equals_the_reference(mname, ordering):
    """The oracle's restatement of the dead-column rule (spqr_front.cpp:424-479) against the reference
    run with SPQR_DEFAULT_TOL: rank, Rdead, the squeezed R and norm_E_fro."""
    A0 = rc.matrix(mname)
    A, r = rc.reference_pruned(A0, ordering, rc.DEFAULT_TOL)
    assert abs(spqr.default_tol(A0) - r.tol_used) <= 1e-15 * r.tol_used     # (the tolerance comes from the whole matrix)
    sym = spqr.analyze(A, r.Q, relax=-1, gpuMemorySize=2**34, rank_detection=True)
    R, Rdead, rank, _, enorm = oracle.multifrontal_tol(sym, spqr.s_values(sym, A), r.tol_used)
    rc.check_numeric_pruned(R, rank, Rdead, r, tol=1e-11)
    # the norm of what the dead columns left behind is itself at rounding level (far below tol)
    assert abs(enorm - r.norm_E_fro) <= 1e-3 * r.tol_used and enorm <= r.tol_used * np.sqrt(max(int(Rdead.sum()), 1))


HB_CASES = [(f, o) for f in ("west0067.rua", "arc130.rua", "fs_183_6.rua") for o in (rc.ORD_FIXED, rc.ORD_COLAMD, rc.ORD_AMD)]


@pytest.mark.parametrize("fname,ordering", HB_CASES, ids=["%s-o%d" % c for c in HB_CASES])
@pytest.mark.parametrize("with_tol", [False, True], ids=["notol", "tol"])
def test_bundled_unsymmetric_matrices_against_the_reference(fname, ordering, with_tol):
    """The real unsymmetric matrices the reference tree holds (UMFPACK/Demo/HB, read in place): symbolic
    arrays bit for bit, then the oracle's R, rank and dead columns against the reference's, without a
    tolerance and at SPQR_DEFAULT_TOL.  Where the reference removes column singletons first, the
    multifrontal part sees the pruned matrix (spqr_1factor.cpp:526-552), and so does this path: the
    matrix is pruned with the reference's own singleton permutations, and R22 of its [R11 R12; 0 R22] is
    what is compared."""
    A = rc.matrix("hb:" + fname)
    Y, r = rc.reference_pruned(A, ordering, rc.DEFAULT_TOL if with_tol else rc.NO_TOL)
    sym = spqr.analyze(Y, r.Q, relax=-1, gpuMemorySize=2**34, rank_detection=with_tol)
    rc.check_symbolic(sym, r)
    if with_tol:
        R, Rdead, rank, _, enorm = oracle.multifrontal_tol(sym, spqr.s_values(sym, Y), r.tol_used)
    else:
        rank = int(sum(min(int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f])) for f in range(sym.nf)))
        R, Rdead = _oracle_r(sym, Y)[:rank], np.zeros(sym.n, np.int8)
    rc.check_numeric_pruned(R, rank, Rdead, r, tol=1e-10)
