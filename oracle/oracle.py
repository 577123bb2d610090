"""ctypes loader of the CPU oracle (oracle/qr_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


class SEntry(C.Structure):
    _fields_ = [("findex", C.c_int64), ("value", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_max_queue_size.argtypes = [C.c_size_t]
        _lib.oracle_max_queue_size.restype = C.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def max_queue_size(mem):
    return int(lib().oracle_max_queue_size(mem))


def find_staircase(F):
    F = np.ascontiguousarray(F, dtype=np.float64)
    fm, fn = F.shape
    st = np.zeros(fn, dtype=np.int64)
    lib().oracle_find_staircase(_p(F), C.c_int64(fm), C.c_int64(fn), _p(st))
    return st


def front_qr(F, Stair=None):
    """Householder QR of a row-major front; returns the reduced front (fm x fn)."""
    A = np.array(F, dtype=np.float64, order="C", copy=True)
    fm, fn = A.shape
    st = None if Stair is None else np.ascontiguousarray(Stair, dtype=np.int64)
    lib().oracle_front_qr(_p(A), C.c_int64(fm), C.c_int64(fn), None if st is None else _p(st))
    return A


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def compute_front_staging(nf, Parent, Childp, Child, Fm, Cm, Rp, Sp, Sleft, Super, Post,
                          RimapSize, RjmapSize, gpuMemorySize):
    """oracle_compute_front_staging; same result dict as suitesparse_amd.spqr.spqrgpu_computeFrontStaging."""
    a = [_i64(x) for x in (Parent, Childp, Child, Fm, Cm, Rp, Sp, Sleft, Super, Post)]
    numStages = C.c_int64(0)
    Stagingp = np.zeros(nf + 2, np.int64)
    StageMap = np.zeros(nf + 1, np.int64)
    FSize, RSize, SSize = (np.zeros(nf + 1, np.uint64) for _ in range(3))
    FOff, ROff, SOff = (np.zeros(nf + 1, np.int64) for _ in range(3))
    f = lib().oracle_compute_front_staging
    f.restype = C.c_int
    f.argtypes = [C.c_int64] + [C.c_void_p] * 10 + [C.c_int64, C.c_int64, C.c_size_t,
                                                    C.POINTER(C.c_int64)] + [C.c_void_p] * 8
    ok = f(nf, *[_p(x) for x in a], RimapSize, RjmapSize, gpuMemorySize, C.byref(numStages),
           _p(Stagingp), _p(StageMap), _p(FSize), _p(RSize), _p(SSize), _p(FOff), _p(ROff), _p(SOff))
    return dict(feasible=bool(ok), numStages=int(numStages.value), Stagingp=Stagingp, StageMap=StageMap,
                FSize=FSize, RSize=RSize, SSize=SSize, FOffsets=FOff, ROffsets=ROff, SOffsets=SOff)


def build_assembly_maps(sym, Sx):
    """oracle_build_assembly_maps on the arrays of a symbolic object (any object with the
    SPQR field names); returns (Stair, Rjmap, Rimap, cpuS)."""
    nf, n = sym.nf, sym.n
    Fmap = np.zeros(max(n, 1), np.int64)
    Stair = np.zeros(max(int(sym.Rp[nf]), 1), np.int64)
    Rjmap = np.zeros(max(sym.RjmapSize, 1), np.int32)
    Rimap = np.zeros(max(sym.RimapSize, 1), np.int32)
    cpuS = np.zeros(max(sym.anz, 1), dtype=np.dtype([("findex", np.int64), ("value", np.float64)]))
    Sx = np.ascontiguousarray(Sx, dtype=np.float64)
    f = lib().oracle_build_assembly_maps
    f.restype = None
    f.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 19
    f(nf, n, _p(Fmap), _p(_i64(sym.Post)), _p(_i64(sym.Super)), _p(_i64(sym.Rp)), _p(_i64(sym.Rj)),
      _p(_i64(sym.Sleft)), _p(_i64(sym.Sp)), _p(_i64(sym.Sj)), _p(Sx), _p(_i64(sym.Fm)), _p(_i64(sym.Cm)),
      _p(_i64(sym.Childp)), _p(_i64(sym.Child)), _p(Stair), _p(Rjmap), _p(_i64(sym.RjmapOffsets)),
      _p(Rimap), _p(_i64(sym.RimapOffsets)), _p(cpuS))
    return Stair, Rjmap, Rimap, cpuS


def multifrontal(sym, Sx):
    """CPU multifrontal QR over the whole tree in the engine's layout (one stage).
    Returns (Farena, FOffsets): front f, reduced, is Farena[FOffsets[f]:][:fm*fn] row-major."""
    nf = sym.nf
    Stair, Rjmap, Rimap, cpuS = build_assembly_maps(sym, Sx)
    fn = np.diff(_i64(sym.Rp))[:nf]
    sizes = _i64(sym.Fm)[:nf] * fn
    FOff = np.zeros(nf + 1, np.int64)
    FOff[1:] = np.cumsum(sizes)
    # S entries are packed front after front in postorder
    Scount = np.array([sym.Sp[sym.Sleft[sym.Super[f + 1]]] - sym.Sp[sym.Sleft[sym.Super[f]]] for f in range(nf)],
                      dtype=np.int64)
    SOff = np.zeros(nf + 1, np.int64)
    post = _i64(sym.Post)[:nf]
    SOff[post] = np.concatenate([[0], np.cumsum(Scount[post])[:-1]]) if nf > 0 else 0
    Farena = np.zeros(max(int(FOff[nf]), 1))
    f = lib().oracle_multifrontal
    f.restype = None
    f.argtypes = [C.c_int64] + [C.c_void_p] * 17
    f(nf, _p(post), _p(_i64(sym.Parent)), _p(_i64(sym.Super)), _p(_i64(sym.Rp)), _p(_i64(sym.Fm)),
      _p(_i64(sym.Cm)), _p(_i64(sym.Sleft)), _p(_i64(sym.Sp)), _p(Stair), _p(Rjmap),
      _p(_i64(sym.RjmapOffsets)), _p(Rimap), _p(_i64(sym.RimapOffsets)), _p(cpuS), _p(FOff), _p(SOff),
      _p(Farena))
    return Farena, FOff


def r_from_fronts(sym, Farena, FOff):
    """Assemble the dense n-column R (rows = sum of front ranks) from reduced fronts."""
    nf, n = sym.nf, sym.n
    rows = int(sum(min(int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f])) for f in range(nf)))
    R = np.zeros((rows, n))
    row1 = 0
    for f in range(nf):
        fm = int(sym.Fm[f])
        fp = int(sym.Super[f + 1] - sym.Super[f])
        cols = _i64(sym.Rj)[sym.Rp[f]:sym.Rp[f + 1]]
        fnn = cols.size
        F = Farena[FOff[f]:FOff[f] + fm * fnn].reshape(fm, fnn)
        rk = min(fm, fp)
        for i in range(rk):
            R[row1 + i, cols[i:]] = F[i, i:]
        row1 += rk
    return R


def front_qr_batch(mats, nthreads=1):
    """Factorize a list of fronts in place on the CPU (OpenMP over fronts); returns seconds."""
    import time
    fm = _i64([a.shape[0] for a in mats])
    fn = _i64([a.shape[1] for a in mats])
    F = np.concatenate([np.ascontiguousarray(a, dtype=np.float64).ravel() for a in mats])
    f = lib().oracle_front_qr_batch
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    t0 = time.perf_counter()
    f(_p(F), _p(fm), _p(fn), len(mats), nthreads)
    return time.perf_counter() - t0


def multifrontal_lapack(sym, Sx, keep_R=True):
    """CPU multifrontal QR with the fronts factorized by LAPACK's blocked Householder QR (dgeqrf:
    dlarfg / dlarft / dlarfb, the routines the reference's CPU path is built from,
    SPQR/Source/spqr_front.cpp:141-154,402-408, spqr_larftb.cpp:48-83) on the threaded BLAS the host
    has.  Fronts in postorder, one at a time, as spqr_kernel does without TBB
    (SPQR/Source/spqr_kernel.cpp:134-321); assembly through the same maps as the device path.  Unlike
    spqr_front it does not skip the zeros under the staircase.  Returns (R blocks: front -> rank x fn
    upper-trapezoidal array, or None when keep_R is False; seconds spent in dgeqrf; seconds in all)."""
    import time
    from scipy.linalg import lapack
    nf = sym.nf
    t_all = time.perf_counter()
    Stair, Rjmap, Rimap, cpuS = build_assembly_maps(sym, Sx)
    post = _i64(sym.Post)[:nf]
    Scount = np.array([sym.Sp[sym.Sleft[sym.Super[f + 1]]] - sym.Sp[sym.Sleft[sym.Super[f]]] for f in range(nf)],
                      dtype=np.int64)
    SOff = np.zeros(nf + 1, np.int64)
    SOff[post] = np.concatenate([[0], np.cumsum(Scount[post])[:-1]]) if nf > 0 else 0
    Cblk = {}
    Rb = {} if keep_R else None
    t_qr = 0.0
    for f in post:
        f = int(f)
        fm, fn, fp = int(sym.Fm[f]), int(sym.Rp[f + 1] - sym.Rp[f]), int(sym.Super[f + 1] - sym.Super[f])
        F = np.zeros((fm, fn), order="F")          # column-major, as LAPACK wants it
        e = cpuS[SOff[f]:SOff[f] + Scount[f]]
        if fn > 0:
            F[e["findex"] // fn, e["findex"] % fn] = e["value"]
        for c in sym.Child[sym.Childp[f]:sym.Childp[f + 1]]:
            c = int(c)
            Cc = Cblk.pop(c, None)
            if Cc is None or Cc.size == 0:
                continue
            cm, cn = Cc.shape
            ri = Rimap[sym.RimapOffsets[c]:sym.RimapOffsets[c] + cm]
            rj = Rjmap[sym.RjmapOffsets[c]:sym.RjmapOffsets[c] + cn]
            # a parent row takes exactly one child row, whose entries left of the diagonal are zero
            F[np.ix_(ri, rj)] = Cc
        rk = min(fm, fp)
        if fm > 0 and fn > 0:
            t0 = time.perf_counter()
            lw, info = lapack.dgeqrf_lwork(fm, fn)      # the blocked code needs its full workspace
            qr, tau, work, info = lapack.dgeqrf(F, lwork=max(int(lw), 1), overwrite_a=True)
            t_qr += time.perf_counter() - t0
            assert info == 0
        else:
            qr = F
        if keep_R:
            Rb[f] = np.triu(qr[:rk])
        cm, cn = int(sym.Cm[f]), fn - fp
        if int(sym.Parent[f]) < nf and cm * cn > 0:
            Cblk[f] = np.triu(qr[rk:rk + cm, fp:])
    return Rb, t_qr, time.perf_counter() - t_all


def multifrontal_tol(sym, Sx, tol):
    """CPU multifrontal QR with SPQR's rank detection (oracle_multifrontal_tol restates
    SPQR/Source/spqr_front.cpp:356-479 and spqr_kernel.cpp:134-321).  sym must come from an analysis with
    rank detection allowed.  Returns (R scipy CSC rank-by-n squeezed, explicit zeros kept, Rdead, rank, per-front ranks, norm_E_fro)."""
    nf, n = sym.nf, sym.n
    Stair, Rjmap, Rimap, cpuS = build_assembly_maps(sym, Sx)
    fn = np.diff(_i64(sym.Rp))[:nf]
    FOff = np.zeros(nf + 1, np.int64)
    FOff[1:] = np.cumsum(_i64(sym.Fm)[:nf] * fn)
    Scount = np.array([sym.Sp[sym.Sleft[sym.Super[f + 1]]] - sym.Sp[sym.Sleft[sym.Super[f]]] for f in range(nf)],
                      dtype=np.int64)
    SOff = np.zeros(nf + 1, np.int64)
    post = _i64(sym.Post)[:nf]
    SOff[post] = np.concatenate([[0], np.cumsum(Scount[post])[:-1]]) if nf > 0 else 0
    Farena = np.zeros(max(int(FOff[nf]), 1))
    Rdead = np.zeros(max(n, 1), np.int8)
    Rank = np.zeros(max(nf, 1), np.int64)
    enorm = C.c_double(0)
    f = lib().oracle_multifrontal_tol
    f.restype = None
    f.argtypes = [C.c_int64] + [C.c_void_p] * 16 + [C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    f(nf, _p(post), _p(_i64(sym.Parent)), _p(_i64(sym.Super)), _p(_i64(sym.Rp)), _p(_i64(sym.Fm)),
      _p(_i64(sym.Cm)), _p(_i64(sym.Sleft)), _p(_i64(sym.Sp)), _p(Rjmap), _p(_i64(sym.RjmapOffsets)),
      _p(Rimap), _p(_i64(sym.RimapOffsets)), _p(cpuS), _p(FOff), _p(SOff), _p(Farena),
      float(tol), _p(Rdead), _p(Rank), C.byref(enorm))
    import scipy.sparse as sp
    rank = int(Rank[:nf].sum())
    ri, ci, vals = [], [], []
    row1 = 0
    for fr in range(nf):
        fm = int(sym.Fm[fr])
        col1, fp = int(sym.Super[fr]), int(sym.Super[fr + 1] - sym.Super[fr])
        cols = _i64(sym.Rj)[sym.Rp[fr]:sym.Rp[fr + 1]]
        fnn = cols.size
        F = Farena[FOff[fr]:FOff[fr] + fm * fnn].reshape(fm, fnn)
        rm = 0
        for k in range(fnn):
            if k < fp and not Rdead[col1 + k] and rm < fm:
                rm += 1
            if rm > 0:
                ri.append(np.arange(row1, row1 + rm))
                ci.append(np.full(rm, cols[k]))
                vals.append(F[:rm, k].copy())
        row1 += int(Rank[fr])
    if ri:
        R = sp.csc_matrix((np.concatenate(vals), (np.concatenate(ri), np.concatenate(ci))), shape=(rank, n))
    else:
        R = sp.csc_matrix((rank, n))
    return R, Rdead[:n].copy(), rank, Rank[:nf].copy(), float(enorm.value)
