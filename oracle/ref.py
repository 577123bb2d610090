"""ctypes loader of oracle/_ref/libspqr_ref.so: the REFERENCE's own CPU SuiteSparseQR, compiled by
oracle/build_ref.sh from the sources under /root/reference (unchanged, where they lie).

TEST INFRASTRUCTURE ONLY: imported by tests/ and bench.py's reference leg (column order and CPU
baseline), never by the product.  The library travels to the GPU box as a built file (it is git-ignored,
not gpurun-ignored); the reference tree does not, and nothing here reads it at run time."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libspqr_ref.so")
_lib = None

ORDERING_FIXED, ORDERING_NATURAL, ORDERING_COLAMD, ORDERING_AMD, ORDERING_METIS, ORDERING_DEFAULT = 0, 1, 2, 5, 6, 7
NO_TOL, DEFAULT_TOL = -1.0, -2.0

_INT_ARRAYS = ("Super", "Rp", "Rj", "Parent", "Childp", "Child", "Post", "Fm", "Cm", "Sp", "Sj", "Sleft",
               "PLinv", "Qfill", "Q1fill", "P1inv")


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libspqr_ref.so is not built (oracle/build_ref.sh needs /root/reference)")
        # The library is linked with GNU OpenMP; MKL's run-time layer must be told to thread through the
        # same runtime.  Left to its default (Intel's) in a process that already holds libgomp (torch
        # does), dlarfb returns wrong numbers without any error: seen as an R that differed from row
        # 907 on in a 1000-column problem.
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        _lib = C.CDLL(_SO)
        _lib.ref_open.restype = C.c_void_p
        _lib.ref_open.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int]
        _lib.ref_close.argtypes = [C.c_void_p]
        _lib.ref_info.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_dinfo.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_get.restype = C.c_int64
        _lib.ref_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        _lib.ref_get_rdead.restype = C.c_int64
        _lib.ref_get_rdead.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_get_E.restype = C.c_int64
        _lib.ref_get_E.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_get_R.restype = C.c_int64
        _lib.ref_get_R.argtypes = [C.c_void_p] * 4
        _lib.ref_metis_order.restype = C.c_int
        _lib.ref_metis_order.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _csc(A):
    A = A.tocsc()
    A.sort_indices()
    return (A, np.ascontiguousarray(A.indptr, np.int64), np.ascontiguousarray(A.indices, np.int64),
            np.ascontiguousarray(A.data, np.float64))


class Result:
    """What one run of the reference returned (plain numpy arrays; the handle is closed)."""


def spqr(A, ordering=ORDERING_FIXED, tol=NO_TOL, symbolic=True, want_R=True):
    """Run the reference: spqr_1factor (symbolic=True: the symbolic object's arrays, rank, Rdead, timers)
    and/or [R,E] = qr(A) (want_R=True: R as scipy CSC, E).  ordering and tol as SuiteSparseQR takes them."""
    import scipy.sparse as sp
    A, Ap, Ai, Ax = _csc(A)
    m, n = A.shape
    L = lib()
    h = L.ref_open(m, n, _p(Ap), _p(Ai), _p(Ax), int(ordering), float(tol), (1 if symbolic else 0) | (2 if want_R else 0))
    if not h:
        raise RuntimeError("the reference's SuiteSparseQR failed")
    try:
        r = Result()
        info = np.zeros(16, np.int64)
        dinfo = np.zeros(8)
        L.ref_info(h, _p(info))
        L.ref_dinfo(h, _p(dinfo))
        r.m, r.n = m, n
        (r.analyze_time, r.factorize_time, r.flopcount, r.flopcount_bound, r.tol_used, r.norm_E_fro) = map(float, dinfo[:6])
        if symbolic:
            (r.nf, r.sm, r.sn, r.anz, r.rjsize, r.rank, r.maxfrank, r.n1cols, r.n1rows) = map(int, info[:9])
            for name in _INT_ARRAYS:
                ln = int(L.ref_get(h, name.encode(), None))
                a = np.zeros(max(ln, 0), np.int64)
                if ln > 0:
                    L.ref_get(h, name.encode(), _p(a))
                setattr(r, name, a)
            rd = np.zeros(max(r.sn, 1), np.int8)
            r.Rdead = rd[:r.sn] if L.ref_get_rdead(h, _p(rd)) else np.zeros(r.sn, np.int8)
        if want_R:
            nz = int(info[9])
            r.rank_R = int(info[10])
            Rp = np.zeros(n + 1, np.int64)
            Ri = np.zeros(max(nz, 1), np.int64)
            Rx = np.zeros(max(nz, 1))
            nrow = int(L.ref_get_R(h, _p(Rp), _p(Ri), _p(Rx)))
            r.R = sp.csc_matrix((Rx[:nz], Ri[:nz], Rp), shape=(nrow, n))
            E = np.zeros(max(n, 1), np.int64)
            r.E = E[:n] if L.ref_get_E(h, _p(E)) else np.arange(n, dtype=np.int64)
        return r
    finally:
        L.ref_close(h)


def threads():
    """The number of threads MKL gives the reference's BLAS/LAPACK calls in this process."""
    L = lib()
    L.ref_threads.restype = C.c_int
    return int(L.ref_threads())


def metis_order(A):
    """Column ordering of A by the reference's own METIS binding on A'A (cholmod_l_metis of A' with
    postordering, the call SPQR makes for SPQR_ORDERING_METIS, spqr_1colamd.cpp:561-608)."""
    A, Ap, Ai, _ = _csc(A)
    m, n = A.shape
    perm = np.zeros(n, np.int64)
    if lib().ref_metis_order(m, n, _p(Ap), _p(Ai), _p(perm)) != 0:
        raise RuntimeError("cholmod_l_metis failed")
    return perm


def metis_order_child(A):
    """metis_order in a child process: METIS on a multi-million-column graph takes and frees several
    GB of heap, which is kept out of a process that drives a GPU this way."""
    import subprocess
    import sys
    import tempfile
    A, Ap, Ai, _ = _csc(A)
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "Ap.npy"), Ap)
        np.save(os.path.join(d, "Ai.npy"), Ai)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), d, str(A.shape[0]), str(A.shape[1])])
        return np.load(os.path.join(d, "perm.npy"))


if __name__ == "__main__":
    import sys
    d, m, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    Ap, Ai = np.load(os.path.join(d, "Ap.npy")), np.load(os.path.join(d, "Ai.npy"))
    perm = np.zeros(n, np.int64)
    if lib().ref_metis_order(m, n, _p(Ap), _p(Ai), _p(perm)) != 0:
        raise SystemExit("cholmod_l_metis failed")
    np.save(os.path.join(d, "perm.npy"), perm)
