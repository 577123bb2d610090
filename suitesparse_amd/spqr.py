"""Python mirror of the SPQR GPU glue, on top of the C ABI in include/spqrgpu_amd.h.

Function names follow SPQR/Include/spqrgpu.hpp:9-82.  Plumbing only.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Numeric(C.Structure):
    """spqr_amd_numeric."""
    _fields_ = [("nf", C.c_int64), ("rsize", C.c_int64), ("Rstack", C.POINTER(C.c_double)),
                ("Rblock", C.POINTER(C.c_int64)), ("Rdead", C.POINTER(C.c_char)),
                ("sumfrank", C.c_int64), ("maxfrank", C.c_int64), ("gpuKernelTime", C.c_double),
                ("gpuFlops", C.c_int64), ("gpuNumKernelLaunches", C.c_int64),
                ("numStagesRun", C.c_int64), ("assembleBytes", C.c_double),
                ("Cstack", C.POINTER(C.c_double)), ("Cblock", C.POINTER(C.c_int64)),
                ("RstackPinned", C.c_int64), ("csize", C.c_int64),
                ("RstackBorrowed", C.c_int64), ("numPasses", C.c_int64), ("norm_E_fro", C.c_double)]


_ARRAYS = ["Qfill", "PLinv", "Sleft", "Sp", "Sj", "Super", "Rp", "Rj", "Parent", "Childp", "Child",
           "Post", "Fm", "Cm", "RjmapOffsets", "RimapOffsets", "Stagingp", "StageMap", "FSize",
           "RSize", "SSize", "FOffsets", "ROffsets", "SOffsets", "ColCount"]
_SCALARS = ["m", "n", "anz", "nf", "maxfn", "rjsize", "RimapSize", "RjmapSize", "numStages",
            "flopcount", "rnz"]


def _bind():
    L = _lib.lib()
    if getattr(L, "_spqr_bound", False):
        return L
    L.spqr_amd_analyze.restype = C.c_void_p
    L.spqr_amd_analyze.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_size_t, C.POINTER(C.c_int)]
    L.spqr_amd_symbolic_free.argtypes = [C.c_void_p]
    L.spqr_amd_symbolic_array.restype = C.POINTER(C.c_int64)
    L.spqr_amd_symbolic_array.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    L.spqr_amd_symbolic_scalar.restype = C.c_double
    L.spqr_amd_symbolic_scalar.argtypes = [C.c_void_p, C.c_char_p]
    L.spqrgpu_amd_factorize.restype = C.c_int
    L.spqrgpu_amd_factorize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.POINTER(Numeric))]
    L.spqr_amd_numeric_free.argtypes = [C.POINTER(Numeric)]
    L.spqr_amd_rconvert.restype = C.c_int
    L.spqr_amd_rconvert.argtypes = [C.c_void_p, C.POINTER(Numeric), C.c_void_p, C.c_void_p, C.c_void_p]
    L.spqrgpu_amd_compute_front_staging.restype = C.c_int
    L.spqrgpu_amd_numfronts_in_stage.restype = C.c_int64
    L._spqr_bound = True
    return L


class Symbolic:
    """Owner of a spqr_amd_symbolic; arrays are exposed as numpy views by name."""

    def __init__(self, handle, status):
        self._h = handle
        self.status = status
        L = _bind()
        for name in _SCALARS:
            v = L.spqr_amd_symbolic_scalar(handle, name.encode())
            setattr(self, name, v if name == "flopcount" else int(v))
        for name in _ARRAYS:
            ln = C.c_int64(0)
            p = L.spqr_amd_symbolic_array(handle, name.encode(), C.byref(ln))
            arr = np.ctypeslib.as_array(p, shape=(ln.value,)) if ln.value > 0 else np.zeros(0, np.int64)
            setattr(self, name, arr)

    def __del__(self):
        if getattr(self, "_h", None) and _bind is not None:     # (_bind is None while the interpreter shuts down)
            _bind().spqr_amd_symbolic_free(self._h)
            self._h = None


def analyze(A, Q=None, relax=0, gpuMemorySize=0, rank_detection=False):
    """spqr_amd_analyze[_rd]: symbolic analysis + staging of A(:,Q) for the GPU path; with
    rank_detection the fronts are sized for pivot columns that die at a tolerance
    (SPQR/Source/spqr_analyze.cpp:826-842), which factorize(..., tol >= 0) needs."""
    L = _bind()
    A = sp.csc_matrix(A)
    A.sort_indices()
    m, n = A.shape
    Ap, Ai = _i64(A.indptr), _i64(A.indices)
    Qa = None if Q is None else _i64(Q)
    st = C.c_int(0)
    L.spqr_amd_analyze_rd.restype = C.c_void_p
    L.spqr_amd_analyze_rd.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_int)]
    h = L.spqr_amd_analyze_rd(m, n, _p(Ap), _p(Ai), _p(Qa), relax, gpuMemorySize, int(bool(rank_detection)), C.byref(st))
    if not h:
        raise MemoryError("spqr_amd_analyze failed, status %d" % st.value)
    sym = Symbolic(h, st.value)
    sym.gpuMemorySize = gpuMemorySize
    sym.rank_detection = bool(rank_detection)
    return sym


class NumericResult:
    _num = None

    def free(self):
        """Release the numeric object kept by factorize(..., keep_num=True)."""
        if self._num:
            _bind().spqr_amd_numeric_free(self._num)
            self._num = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def default_tol(A):
    """SPQR_DEFAULT_TOL = 20 (m+n) eps max_j ||A(:,j)||_2 (SPQR/Source/spqr_tol.cpp:14-27)."""
    L = _bind()
    A = sp.csc_matrix(A)
    Ap, Ax = _i64(A.indptr), np.ascontiguousarray(A.data, dtype=np.float64)
    L.spqr_amd_default_tol.restype = C.c_double
    L.spqr_amd_default_tol.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    return float(L.spqr_amd_default_tol(A.shape[0], A.shape[1], _p(Ap), _p(Ax)))


def factorize(sym, A, want_R=True, tol=None, keep_num=False):
    """spqrgpu_amd_factorize, or spqrgpu_amd_factorize_tol when a tolerance is given
    (+ spqr_amd_rconvert when want_R): returns R (scipy CSC, rank-by-n) and stats.  With keep_num the
    result keeps the numeric object (the packed R blocks) for rsolve(); release it with res.free()."""
    L = _bind()
    if hasattr(A, "Ap"):                # arrays already in the C ABI's types (multigpu._Csc)
        Ap, Ai, Ax = A.Ap, A.Ai, A.Ax
    else:
        A = sp.csc_matrix(A)
        A.sort_indices()
        Ap, Ai, Ax = _i64(A.indptr), _i64(A.indices), np.ascontiguousarray(A.data, dtype=np.float64)
    out = C.POINTER(Numeric)()
    if tol is None:
        code = L.spqrgpu_amd_factorize(sym._h, _p(Ap), _p(Ai), _p(Ax), C.byref(out))
    else:
        L.spqrgpu_amd_factorize_tol.restype = C.c_int
        L.spqrgpu_amd_factorize_tol.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                                C.POINTER(C.POINTER(Numeric))]
        code = L.spqrgpu_amd_factorize_tol(sym._h, _p(Ap), _p(Ai), _p(Ax), float(tol), C.byref(out))
    if code != 0:
        raise RuntimeError("spqrgpu_amd_factorize failed with code %d" % code)
    num = out.contents
    n = sym.n
    res = NumericResult()
    res.rank = int(num.sumfrank)
    res.maxfrank = int(num.maxfrank)
    res.R = None
    if want_R:
        Rp = np.zeros(n + 1, dtype=np.int64)
        L.spqr_amd_rconvert(sym._h, out, _p(Rp), None, None)
        Ri = np.zeros(max(int(Rp[n]), 1), dtype=np.int64)
        Rx = np.zeros(max(int(Rp[n]), 1), dtype=np.float64)
        L.spqr_amd_rconvert(sym._h, out, _p(Rp), _p(Ri), _p(Rx))
        res.R = sp.csc_matrix((Rx[:Rp[n]], Ri[:Rp[n]], Rp), shape=(max(res.rank, 0), n))
    res.Rdead = np.frombuffer(C.string_at(num.Rdead, n), dtype=np.int8).copy() if n > 0 else np.zeros(0, np.int8)
    res.gpuKernelTime = float(num.gpuKernelTime)
    res.gpuFlops = int(num.gpuFlops)
    res.gpuNumKernelLaunches = int(num.gpuNumKernelLaunches)
    res.numStagesRun = int(num.numStagesRun)
    res.numPasses = int(num.numPasses)
    res.norm_E_fro = float(num.norm_E_fro)
    res.assembleBytes = float(num.assembleBytes)
    if keep_num:
        res._num = out
    else:
        L.spqr_amd_numeric_free(out)
    return res


def rsolve(sym, res, Cq, nacols=None, permute=True):
    """spqr_amd_rsolve: X = E (R \\ Cq) by back substitution on the packed R blocks of a factorization
    made with keep_num (SPQR/Source/spqr_rsolve.cpp:136-338).  Cq is rank-by-k in R's row order; columns
    of the factorization from nacols on are appended right-hand sides and are passed over."""
    L = _bind()
    n = int(sym.n if nacols is None else nacols)
    Cq = np.asarray(Cq, dtype=np.float64)
    if Cq.ndim == 1:
        Cq = Cq[:, None]
    rank, k = Cq.shape
    Cf = np.zeros((max(rank, 1), max(k, 1)), order="F")
    Cf[:rank, :k] = Cq
    X = np.zeros((max(n, 1), max(k, 1)), order="F")
    Qfill = _i64(sym.Qfill) if permute else None
    L.spqr_amd_rsolve.restype = C.c_int
    L.spqr_amd_rsolve.argtypes = [C.c_void_p, C.POINTER(Numeric), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                  C.c_int64, C.c_void_p]
    code = L.spqr_amd_rsolve(sym._h, res._num, _p(Qfill) if permute else None, n, k, _p(Cf), max(rank, 1), _p(X))
    if code != 0:
        raise RuntimeError("spqr_amd_rsolve failed with code %d" % code)
    return X[:n, :k].copy()


def rtsolve(sym, res, B):
    """spqr_amd_rtsolve: X = R' \\ B on the packed R blocks of a factorization made with keep_num
    (SuiteSparseQR_solve with SPQR_RTX_EQUALS_B in the reference).  B is n-by-k in the factorization's
    column order; returns X, rank-by-k, in the row order of R."""
    L = _bind()
    L.spqr_amd_rtsolve.restype = C.c_int
    L.spqr_amd_rtsolve.argtypes = [C.c_void_p, C.POINTER(Numeric), C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    n, rank = int(sym.n), int(res.rank)
    B = np.asarray(B, dtype=np.float64).reshape(n, -1)
    k = B.shape[1]
    Bf = np.asfortranarray(B) if n > 0 else np.zeros((1, k), order="F")
    X = np.zeros((max(rank, 1), max(k, 1)), order="F")
    code = L.spqr_amd_rtsolve(sym._h, res._num, k, _p(Bf), max(n, 1), _p(X), max(rank, 1))
    if code != 0:
        raise RuntimeError("spqr_amd_rtsolve failed with code %d" % code)
    return X[:rank, :k].copy()


def rmult(sym, res, v, transposed=False):
    """spqr_amd_rmult: R v (v of size n, the factorization's column order) or R' v (v of size rank) on the
    packed R blocks of a factorization made with keep_num."""
    L = _bind()
    L.spqr_amd_rmult.restype = C.c_int
    L.spqr_amd_rmult.argtypes = [C.c_void_p, C.POINTER(Numeric), C.c_int, C.c_void_p, C.c_void_p]
    v = np.ascontiguousarray(v, dtype=np.float64)
    n, rank = int(sym.n), int(res.rank)
    assert v.size == (rank if transposed else n)
    out = np.zeros(max(n if transposed else rank, 1))
    code = L.spqr_amd_rmult(sym._h, res._num, int(bool(transposed)), _p(v), _p(out))
    if code != 0:
        raise RuntimeError("spqr_amd_rmult failed with code %d" % code)
    return out[:n if transposed else rank]


def gram_probe(sym, res, A, seed=0):
    """The reference's check of an R factor, ||R'R - (AP)'(AP)||_1 / ||A'A||_1 <= 1e-10
    (check_r_factor, SPQR/Tcov/qrtest.cpp:1218-1257,2750), through one random vector: returns
    max |R'(R y) - (AP)'((AP) y)| / max |(AP)'((AP) y)| for y uniform in (0.5, 1.5), P = Qfill.  For a
    factorization of full column rank made with keep_num."""
    A = sp.csc_matrix(A)
    n = int(sym.n)
    y = np.random.default_rng(seed).uniform(0.5, 1.5, size=n)
    z = rmult(sym, res, rmult(sym, res, y), transposed=True)
    yq = np.zeros(A.shape[1])
    yq[np.asarray(sym.Qfill)[:n]] = y           # (A P) y = A (P y)
    w = A.T @ (A @ yq)
    z2 = w[np.asarray(sym.Qfill)[:n]]
    return float(np.abs(z - z2).max() / np.abs(z2).max())


def spqrgpu_computeFrontStaging(nf, Parent, Childp, Child, Fm, Cm, Rp, Sp, Sleft, Super, Post,
                                RimapSize, RjmapSize, gpuMemorySize):
    """spqrgpu_computeFrontStaging (SPQR/Include/spqrgpu.hpp:16-57) on caller arrays.
    Returns dict(feasible, numStages, Stagingp, StageMap, FSize, RSize, SSize, FOffsets, ROffsets, SOffsets)."""
    L = _bind()
    a = [_i64(x) for x in (Parent, Childp, Child, Fm, Cm, Rp, Sp, Sleft, Super, Post)]
    numStages = C.c_int64(0)
    Stagingp = np.zeros(nf + 2, np.int64)
    StageMap = np.zeros(nf + 1, np.int64)
    FSize, RSize, SSize = (np.zeros(nf + 1, np.uint64) for _ in range(3))
    FOff, ROff, SOff = (np.zeros(nf + 1, np.int64) for _ in range(3))
    L.spqrgpu_amd_compute_front_staging.argtypes = [C.c_int64] + [C.c_void_p] * 10 + [
        C.c_int64, C.c_int64, C.c_size_t, C.POINTER(C.c_int64)] + [C.c_void_p] * 8
    ok = L.spqrgpu_amd_compute_front_staging(
        nf, *[_p(x) for x in a], RimapSize, RjmapSize, gpuMemorySize, C.byref(numStages),
        _p(Stagingp), _p(StageMap), _p(FSize), _p(RSize), _p(SSize), _p(FOff), _p(ROff), _p(SOff))
    return dict(feasible=bool(ok), numStages=int(numStages.value), Stagingp=Stagingp, StageMap=StageMap,
                FSize=FSize, RSize=RSize, SSize=SSize, FOffsets=FOff, ROffsets=ROff, SOffsets=SOff)


def spqrgpu_buildAssemblyMaps(sym, Sx):
    """spqrgpu_buildAssemblyMaps (SPQR/Include/spqrgpu.hpp:59-82) for a Symbolic; returns
    (Stair, Rjmap, Rimap, cpuS) with cpuS a structured array (findex, value)."""
    L = _bind()
    nf, n = sym.nf, sym.n
    Fmap = np.zeros(max(n, 1), np.int64)
    Stair = np.zeros(max(int(sym.Rp[nf]), 1), np.int64)
    Rjmap = np.zeros(max(sym.RjmapSize, 1), np.int32)
    Rimap = np.zeros(max(sym.RimapSize, 1), np.int32)
    cpuS = np.zeros(max(sym.anz, 1), dtype=np.dtype([("findex", np.int64), ("value", np.float64)]))
    Sx = np.ascontiguousarray(Sx, dtype=np.float64)
    L.spqrgpu_amd_build_assembly_maps.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 19
    L.spqrgpu_amd_build_assembly_maps.restype = None
    L.spqrgpu_amd_build_assembly_maps(
        nf, n, _p(Fmap), _p(sym.Post), _p(sym.Super), _p(sym.Rp), _p(sym.Rj), _p(sym.Sleft), _p(sym.Sp),
        _p(sym.Sj), _p(Sx), _p(sym.Fm), _p(sym.Cm), _p(sym.Childp), _p(sym.Child), _p(Stair), _p(Rjmap),
        _p(sym.RjmapOffsets), _p(Rimap), _p(sym.RimapOffsets), _p(cpuS))
    return Stair, Rjmap, Rimap, cpuS


def s_values(sym, A):
    """Sx: the values of S = A(P,Q) in row order (spqr_stranspose2)."""
    A = sp.csc_matrix(A)
    A.sort_indices()
    Sx = np.zeros(sym.anz)
    w = sym.Sp[:-1].copy()
    for k in range(sym.n):
        col = sym.Qfill[k]
        for p in range(A.indptr[col], A.indptr[col + 1]):
            r = sym.PLinv[A.indices[p]]
            Sx[w[r]] = A.data[p]
            w[r] += 1
    return Sx


def append_rhs(A, B, Q=None):
    """[A B] with B's columns stored structurally dense (explicit zeros kept) and an ordering that
    puts them last.  A zero of B that was dropped from the pattern could let the column etree's
    postorder number a column of B before columns of A (a right-hand side supported on one block of
    a block-diagonal A, say); a dense column is an ancestor of every column, so it stays last, which
    is where the reference keeps B (SPQR/Source/SuiteSparseQR.cpp:256-260 never orders it)."""
    A = sp.csc_matrix(A)
    m, n = A.shape
    B = np.asarray(B, dtype=np.float64).reshape(m, -1)
    k = B.shape[1]
    Bs = sp.csc_matrix((B.ravel(order="F"), np.tile(np.arange(m), k), np.arange(k + 1) * m), shape=(m, k))
    AB = sp.csc_matrix(sp.hstack([A, Bs], format="csc"))
    assert AB.nnz == A.nnz + m * k                        # the explicit zeros are still there
    Qa = np.arange(n) if Q is None else np.asarray(Q)
    return AB, np.concatenate([Qa, n + np.arange(k)])


def qr_solve(A, B, Q=None, relax=0, gpuMemorySize=2**34):
    """min ||A x - B||_2 the way SuiteSparseQR gets Q'B on the GPU path (SuiteSparseQR.cpp:256-260):
    B's columns are appended to A and ride through the same factorization, so that the first n rows
    of their part of R are C = Q'B; the back substitution R x = C is done on the host, as
    spqr_rsolve is in the reference.  A is m-by-n of full column rank, B m-by-k dense.
    Returns (x in A's column order, R, Symbolic of [A B])."""
    from scipy.sparse.linalg import spsolve_triangular
    A = sp.csc_matrix(A)
    m, n = A.shape
    B = np.asarray(B, dtype=np.float64).reshape(m, -1)
    k = B.shape[1]
    AB, Qab = append_rhs(A, B, Q)
    sym = analyze(AB, Qab, relax=relax, gpuMemorySize=gpuMemorySize)
    # R(0:n,0:n) is the factor of A(:,Q) only if the analysis kept B's columns behind all of A's
    if not np.array_equal(np.sort(sym.Qfill[:n]), np.arange(n)):
        raise RuntimeError("the appended right-hand sides were ordered among A's columns")
    res = factorize(sym, AB, keep_num=True)
    try:
        R = res.R.tocsc()
        if res.rank < n:
            raise np.linalg.LinAlgError("A(:,Q) is structurally rank deficient on the GPU path (rank %d < %d)" % (res.rank, n))
        # rows 0..rank-1 of the appended columns are Q'B (and what lies below row n the residual's part);
        # the back substitution runs on the packed blocks and passes over the appended columns
        x = rsolve(sym, res, R[:, n:].toarray(), nacols=n)
    finally:
        res.free()
    return x, R, sym


def symbolic_from_arrays(sym, gpuMemorySize=None):
    """spqr_amd_symbolic_from_arrays on the arrays of an existing Symbolic (or any object with the same
    attribute names, e.g. arrays dumped from an SPQR run): the object an SPQR build makes from QRsym."""
    L = _bind()
    nf = int(sym.nf)
    a = {k: _i64(getattr(sym, k)) for k in ("Sp", "Sj", "Sleft", "Super", "Rp", "Rj", "Parent", "Childp", "Child",
                                             "Post", "Fm", "Cm", "RimapOffsets", "RjmapOffsets", "Stagingp", "StageMap")}
    L.spqr_amd_symbolic_from_arrays.restype = C.c_void_p
    L.spqr_amd_symbolic_from_arrays.argtypes = ([C.c_int64] * 4 + [C.c_void_p] * 13 + [C.c_int64, C.c_void_p, C.c_int64,
                                                C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)])
    st = C.c_int(0)
    mem = int(gpuMemorySize if gpuMemorySize is not None else getattr(sym, "gpuMemorySize", 2**34))
    h = L.spqr_amd_symbolic_from_arrays(
        int(sym.m), int(sym.n), int(sym.anz), nf, _p(a["Sp"]), _p(a["Sj"]), _p(a["Sleft"]), _p(a["Super"]), _p(a["Rp"]),
        _p(a["Rj"]), _p(a["Parent"]), _p(a["Childp"]), _p(a["Child"]), _p(a["Post"]), _p(a["Fm"]), _p(a["Cm"]),
        _p(a["RimapOffsets"]), int(sym.RimapSize), _p(a["RjmapOffsets"]), int(sym.RjmapSize),
        int(sym.numStages), _p(a["Stagingp"]), _p(a["StageMap"]), mem, C.byref(st))
    if not h:
        raise ValueError("spqr_amd_symbolic_from_arrays failed, status %d" % st.value)
    return Symbolic(h, st.value)


def spqrgpu_kernel(sym, Sx, stack_room=None):
    """spqrgpu_amd_kernel: the body of the reference's spqrgpu_kernel (SPQR/SPQRGPU/spqrgpu_kernel.cpp:63-782)
    on plain arrays.  Returns dict(code, stack (the R blocks as laid on the caller's stack), Rblock, Rdead,
    sumfrank, maxfrank, used, kernel_ms, flops, launches)."""
    L = _bind()
    nf, n = int(sym.nf), int(sym.n)
    Sx = None if Sx is None else np.ascontiguousarray(Sx, dtype=np.float64)
    room = int(sym.rnz if stack_room is None else stack_room)
    stack = np.zeros(max(room, 1))
    Rblock = np.full(max(nf, 1), -1, np.int64)
    Rdead = np.zeros(max(n, 1), np.int8)
    sumfrank, maxfrank, used, flops, launches = (C.c_int64(0) for _ in range(5))
    kms = C.c_double(0)
    L.spqrgpu_amd_kernel.restype = C.c_int
    L.spqrgpu_amd_kernel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p] + [
        C.POINTER(C.c_int64)] * 3 + [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    code = L.spqrgpu_amd_kernel(sym._h, _p(Sx), _p(stack), room, _p(Rblock), _p(Rdead), C.byref(sumfrank),
                                C.byref(maxfrank), C.byref(used), C.byref(kms), C.byref(flops), C.byref(launches))
    return dict(code=code, stack=stack, Rblock=Rblock, Rdead=Rdead, sumfrank=sumfrank.value, maxfrank=maxfrank.value,
                used=used.value, kernel_ms=kms.value, flops=flops.value, launches=launches.value)


class KeptFactorization:
    """spqrgpu_amd_factorize_keep: R plus H kept on the device, for Q'B and solves with right-hand
    sides that arrive later (SuiteSparseQR_factorize + SuiteSparseQR_qmult / _solve in the reference's
    expert interface, SPQR/Source/SuiteSparseQR_expert.cpp:324-343, which its GPU path cannot serve)."""

    def __init__(self, sym, A, nrhs=1):
        L = _bind()
        self.sym, self.nrhs = sym, int(nrhs)
        A = sp.csc_matrix(A)
        A.sort_indices()
        Ap, Ai, Ax = _i64(A.indptr), _i64(A.indices), np.ascontiguousarray(A.data, dtype=np.float64)
        out = C.POINTER(Numeric)()
        self._qt = C.c_void_p(0)
        L.spqrgpu_amd_factorize_keep.restype = C.c_int
        L.spqrgpu_amd_factorize_keep.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                                 C.POINTER(C.POINTER(Numeric)), C.POINTER(C.c_void_p)]
        code = L.spqrgpu_amd_factorize_keep(sym._h, _p(Ap), _p(Ai), _p(Ax), self.nrhs, C.byref(out), C.byref(self._qt))
        if code != 0:
            raise RuntimeError("spqrgpu_amd_factorize_keep failed with code %d" % code)
        n = sym.n
        Rp = np.zeros(n + 1, dtype=np.int64)
        L.spqr_amd_rconvert(sym._h, out, _p(Rp), None, None)
        Ri = np.zeros(max(int(Rp[n]), 1), dtype=np.int64)
        Rx = np.zeros(max(int(Rp[n]), 1), dtype=np.float64)
        L.spqr_amd_rconvert(sym._h, out, _p(Rp), _p(Ri), _p(Rx))
        self.rank = int(out.contents.sumfrank)
        self.R = sp.csc_matrix((Rx[:Rp[n]], Ri[:Rp[n]], Rp), shape=(self.rank, n))
        self._num = out                     # the packed R blocks: spqrgpu_amd_solve substitutes back on them
        L.spqr_amd_qt_rank.restype = C.c_int64
        L.spqr_amd_qt_rank.argtypes = [C.c_void_p]
        assert L.spqr_amd_qt_rank(self._qt) == self.rank

    def qtmult(self, B):
        """The first rank rows of Q'B (rows in the order of R), B m-by-k with k <= nrhs."""
        L = _bind()
        m = self.sym.m
        B = np.asarray(B, dtype=np.float64).reshape(m, -1)
        k = B.shape[1]
        if k > self.nrhs:
            raise ValueError("more right-hand sides than the factorization was kept for")
        Bf = np.zeros((m, self.nrhs), order="F")
        Bf[:, :k] = B
        Cf = np.zeros((max(self.rank, 1), self.nrhs), order="F")
        L.spqrgpu_amd_qtmult.restype = C.c_int
        L.spqrgpu_amd_qtmult.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        code = L.spqrgpu_amd_qtmult(self._qt, _p(Bf), m, _p(Cf), max(self.rank, 1))
        if code != 0:
            raise RuntimeError("spqrgpu_amd_qtmult failed with code %d" % code)
        return Cf[:self.rank, :k].copy()

    def solve(self, B):
        """min ||A x - B||_2: x = E (R \\ (Q'B)) through spqrgpu_amd_solve: Q'B on the device, back
        substitution on the packed R blocks on the host as spqr_rsolve does it in the reference
        (SPQR/Source/spqr_rsolve.cpp).  A rank-deficient A gets the basic solution."""
        L = _bind()
        m, n = self.sym.m, self.sym.n
        B = np.asarray(B, dtype=np.float64).reshape(m, -1)
        k = B.shape[1]
        if k > self.nrhs:
            raise ValueError("more right-hand sides than the factorization was kept for")
        Bf = np.zeros((m, self.nrhs), order="F")
        Bf[:, :k] = B
        X = np.zeros((max(n, 1), self.nrhs), order="F")
        L.spqrgpu_amd_solve.restype = C.c_int
        L.spqrgpu_amd_solve.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Numeric), C.c_void_p, C.c_int64, C.c_void_p]
        code = L.spqrgpu_amd_solve(self._qt, self.sym._h, self._num, _p(Bf), m, _p(X))
        if code != 0:
            raise RuntimeError("spqrgpu_amd_solve failed with code %d" % code)
        return X[:n, :k].copy()

    def close(self):
        if getattr(self, "_qt", None) and self._qt.value:
            L = _bind()
            L.spqr_amd_qt_free.argtypes = [C.c_void_p]
            L.spqr_amd_qt_free(self._qt)
            self._qt = C.c_void_p(0)
        if getattr(self, "_num", None):
            _bind().spqr_amd_numeric_free(self._num)
            self._num = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
