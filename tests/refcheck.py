"""Comparisons with the REFERENCE's own CPU SuiteSparseQR, run live through oracle/_ref/libspqr_ref.so
(oracle/build_ref.sh compiles it from the reference's sources where they lie; the built library
travels with the tree, the reference does not).  No reference output is stored in the repository:
every test that uses this module runs the reference on generated inputs and compares on the spot."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ref
from suitesparse_amd import problems

ORD_FIXED, ORD_COLAMD, ORD_AMD, ORD_METIS = 0, 2, 5, 6
NO_TOL, DEFAULT_TOL = -1.0, -2.0
SYM_ARRAYS = ("Super", "Rp", "Rj", "Parent", "Post", "Childp", "Child", "Sp", "Sleft", "PLinv", "Qfill")

needs_reference = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libspqr_ref.so is not built")


def ash219_standin():
    """219 x 85 0/1 matrix with two entries per row, seed 219 (the collection's ash219 is not in the
    reference tree)."""
    rng = np.random.default_rng(219)
    rows = np.repeat(np.arange(219), 2)
    cols = np.concatenate([rng.choice(85, size=2, replace=False) for _ in range(219)])
    return sp.csc_matrix((np.ones(438), (rows, cols)), shape=(219, 85))


def read_harwell_boeing(path):
    """A real assembled Harwell-Boeing file (RUA) as CSC: header of four lines, then column pointers,
    row indices and values in the fixed-width Fortran formats the header names (scipy's reader does not
    take formats like (1P3D24.15))."""
    import re

    def width(fmt):
        # (16I5) -> 5; (1P3D24.15) -> 24; (4D20.12) -> 20
        m = re.search(r"(\d+)\s*[IiDdEeFfGg]\s*(\d+)", fmt.replace("1P", "").replace("1p", "").replace(",", ""))
        return int(m.group(2))

    with open(path) as f:
        lines = f.read().split("\n")
    counts = [int(x) for x in lines[1].split()]
    ptrcrd, indcrd, valcrd = counts[1], counts[2], counts[3]
    mxtype = lines[2][:3].upper()
    nrow, ncol, nnz = (int(x) for x in lines[2][3:].split()[:3])
    assert mxtype == "RUA", mxtype
    fmts = re.findall(r"\([^)]*\)", lines[3])

    def fields(first, ncards, w, count):
        out = []
        for ln in lines[first:first + ncards]:
            out += [ln[i:i + w] for i in range(0, len(ln.rstrip()), w)]
        return [x for x in out if x.strip()][:count]

    at = 4
    ptr = np.array([int(x) for x in fields(at, ptrcrd, width(fmts[0]), ncol + 1)]) - 1
    at += ptrcrd
    ind = np.array([int(x) for x in fields(at, indcrd, width(fmts[1]), nnz)]) - 1
    at += indcrd
    val = np.array([float(x.replace("D", "E").replace("d", "e")) for x in fields(at, valcrd, width(fmts[2]), nnz)])
    assert len(ptr) == ncol + 1 and len(ind) == nnz and len(val) == nnz and ptr[-1] == nnz
    return sp.csc_matrix((val, ind, ptr), shape=(nrow, ncol))


def matrix(name):
    kind, arg = name.split(":")
    if kind == "grid":
        return problems.stacked_ls(tuple(int(x) for x in arg.split("x")))
    if kind == "rankdef":
        k, nd, nn = (int(x) for x in arg.split("x"))
        return problems.rank_deficient_ls(k, nd, nn)[0]
    if kind == "ash219":
        return ash219_standin()
    if kind == "hb":
        # a Harwell-Boeing file of the reference tree, read where it lies (never copied); the test is
        # skipped on a machine without the tree
        import os
        path = os.path.join("/root/reference/UMFPACK/Demo/HB", arg)
        if not os.path.exists(path):
            pytest.skip("the reference tree is not on this machine")
        return read_harwell_boeing(path)
    if kind == "rand":
        m, n, seed = (int(x) for x in arg.split("x"))
        A = sp.random(m, n, density=3.0 / n, random_state=seed, format="csc")
        A = A + sp.vstack([sp.identity(n), sp.csc_matrix((m - n, n))]) * 2.0
        return sp.csc_matrix(A)
    raise ValueError(name)


def reference(A, ordering, tol=NO_TOL, want_R=True):
    r = ref.spqr(A, ordering=ordering, tol=tol, symbolic=True, want_R=want_R)
    if r.n1cols:
        pytest.skip("the reference removed column singletons; the GPU path sees the pruned matrix")
    r.Q = r.Q1fill if r.Q1fill.size else np.arange(A.shape[1], dtype=np.int64)
    return r


def reference_pruned(A, ordering, tol=NO_TOL, want_R=True):
    """As reference(), for matrices from which SPQR removes column singletons first (spqr_1colamd): returns
    (Y, r) where Y is the matrix the multifrontal part factorizes, built from the reference's singleton
    permutations as spqr_1factor.cpp:526-552 builds it: Y(P1inv[i] - n1rows, k - n1cols) = A(i, Q1fill[k])
    for the rows and columns behind the singletons.  Its columns are already in the order to factorize
    them in (r.Q is the identity); r.Q1fill keeps the whole permutation."""
    r = ref.spqr(A, ordering=ordering, tol=tol, symbolic=True, want_R=want_R)
    n = A.shape[1]
    if r.n1cols == 0:
        r.Q = r.Q1fill if r.Q1fill.size else np.arange(n, dtype=np.int64)
        return sp.csc_matrix(A), r
    P1 = np.argsort(r.P1inv)                    # row P1[k] of A is row k after the singleton permutation
    Y = sp.csc_matrix(sp.csc_matrix(A)[P1[r.n1rows:], :][:, r.Q1fill[r.n1cols:]])
    Y.sort_indices()
    r.Q = np.arange(n - r.n1cols, dtype=np.int64)
    return Y, r


def check_numeric_pruned(R, rank, Rdead, r, tol=1e-12):
    """check_numeric when the reference took singletons out first: R here is the factor of Y, the
    reference's R is [R11 R12; 0 R22] with n1rows singleton rows on top; R22 is what must agree, the rank is
    n1rows more, and E lists the singleton columns first."""
    n2 = r.n - r.n1cols
    assert int(rank) + r.n1rows == r.rank == r.rank_R, (rank, r.n1rows, r.rank, r.rank_R)
    assert np.array_equal(np.asarray(Rdead[:n2]).astype(np.int8), r.Rdead[:n2])
    order = squeeze_order(r.Rdead, n2)
    q1 = r.Q1fill if r.Q1fill.size else np.arange(r.n, dtype=np.int64)
    assert np.array_equal(np.concatenate([q1[:r.n1cols], q1[r.n1cols:][order]]), r.E)
    Ra = abs(sp.csc_matrix(R))[:, order].tocsc()[:rank]
    Rg = abs(sp.csc_matrix(r.R))[r.n1rows:r.n1rows + rank, r.n1cols:]
    scale = float(abs(sp.csc_matrix(r.R)).max())
    diff = abs(Ra - Rg)
    err = float(diff.max()) / scale if diff.nnz else 0.0
    assert err <= tol, err
    return err


def check_symbolic(sym, r):
    """Every symbolic array equal to the reference's QRsym, entry for entry."""
    assert int(sym.nf) == r.nf, ("nf", sym.nf, r.nf)
    nf = r.nf
    for k in SYM_ARRAYS:
        a, b = np.asarray(getattr(sym, k)), getattr(r, k)
        if k == "Qfill" and b.size == 0:
            b = np.arange(r.sn)
        assert len(a) == len(b) and np.array_equal(a, b), k
    for k in ("Fm", "Cm"):
        assert np.array_equal(np.asarray(getattr(sym, k))[:nf], getattr(r, k)[:nf]), k
    assert float(sym.flopcount) == r.flopcount_bound
    # Sj: the reference's CPU build keeps global columns, the GPU path front-local ones
    # (spqr_analyze.cpp:858-882); map ours back through Rj and compare
    Sj = np.asarray(sym.Sj)
    Sp, Sleft, Super, Rp, Rj = (np.asarray(getattr(sym, k)) for k in ("Sp", "Sleft", "Super", "Rp", "Rj"))
    glob = np.empty_like(Sj)
    for f in range(nf):
        lo, hi = Sp[Sleft[Super[f]]], Sp[Sleft[Super[f + 1]]]
        glob[lo:hi] = Rj[Rp[f] + Sj[lo:hi]]
    lo = Sp[Sleft[Super[nf]]] if nf > 0 else 0
    glob[lo:] = Sj[lo:]
    assert np.array_equal(glob[:lo], r.Sj[:lo]), "Sj"


def squeeze_order(Rdead, n):
    """Live columns first, dead ones after: the column order of the R that [R,E] = qr(A) returns for a
    rank-deficient A (SPQR/Source/spqr_rmap.cpp)."""
    dead = np.asarray(Rdead[:n]).astype(bool)
    return np.concatenate([np.nonzero(~dead)[0], np.nonzero(dead)[0]])


def check_numeric(R, rank, Rdead, r, tol=1e-12):
    """R (scipy sparse, rank-by-n, columns in the factorization's own order) against the reference's
    [R,E] = qr(A): same rank, same dead columns, same E, |R| equal to tol of its largest entry
    (rows of R are unique up to sign).  Returns the largest relative difference."""
    n = r.n
    assert int(rank) == r.rank == r.rank_R, (rank, r.rank, r.rank_R)
    assert np.array_equal(np.asarray(Rdead[:n]).astype(np.int8), r.Rdead[:n])
    order = squeeze_order(r.Rdead, n)
    assert np.array_equal(r.Q[order], r.E)
    Ra = abs(sp.csc_matrix(R))[:, order].tocsc()[:rank]
    Rg = abs(sp.csc_matrix(r.R))[:rank]
    scale = float(Rg.max())
    diff = abs(Ra - Rg)
    err = float(diff.max()) / scale if diff.nnz else 0.0
    assert err <= tol, err
    return err
