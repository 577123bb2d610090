"""CPU tests: oracle vs LAPACK, host-side integer logic vs the oracle (bit exact), C ABI surface."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import suitesparse_amd as ss
from oracle import oracle
from suitesparse_amd import _lib, problems, spqr
from tests.util import demo_rand_front, gram_resid, r_rel_diff, rownorm_sign, stair_front

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- oracle pinning

@pytest.mark.parametrize("count,m,n", [(2, 8, 8), (2, 12, 8), (2, 64, 32), (1, 100, 200)])
def test_oracle_demo_known_answer(count, m, n):
    """The dense demo's assertion ||triu(R)'triu(R) - A'A|| / ||A'A|| <= 1e-12
    (GPUQREngine/Demo/gpuqrengine_demo.cpp:94-100) on its four geometries (:305-308)."""
    for i in range(count):
        A = demo_rand_front(m, n, seed_state=1 + i)
        R = oracle.front_qr(A)
        assert gram_resid(R[:min(m, n)], A) <= 1e-12
        assert np.all(np.tril(R, -1) == 0.0)


@pytest.mark.parametrize("m,n", [(5, 5), (40, 17), (17, 40), (300, 64), (128, 128)])
def test_oracle_matches_lapack_up_to_row_signs(m, n):
    rng = np.random.default_rng(m * 131 + n)
    A = rng.uniform(-1, 1, size=(m, n))
    R = oracle.front_qr(A)[:min(m, n)]
    Rl = np.linalg.qr(A, mode="r")
    assert r_rel_diff(rownorm_sign(R), rownorm_sign(Rl)) <= 1e-13
    # device convention: non-negative diagonal, except where nothing lies below it
    # (sigma = 0 keeps the entry as it is, factorize_vt.cu:277-283)
    kk = min(m, n) - (1 if m <= n else 0)
    assert np.all(np.diag(R)[:kk] >= 0)


def test_oracle_staircase_front():
    rng = np.random.default_rng(2)
    A = stair_front(200, 80, rng)
    st = oracle.find_staircase(A)
    R1 = oracle.front_qr(A, st)
    R0 = oracle.front_qr(A)
    assert np.allclose(R1[:80], R0[:80], rtol=0, atol=1e-13 * np.abs(R0).max())


# ---------------------------------------------------------------- host functions vs oracle

@pytest.mark.parametrize("mem", [0, 1, 2, 50000, 10**6, 2**20 + 104 * 100000 + 5, 10**8, 10**10, 2**40])
def test_max_queue_size(mem):
    assert ss.ssgpu_maxQueueSize(mem) == oracle.max_queue_size(mem)


def test_find_staircase_matches_oracle():
    rng = np.random.default_rng(13)
    for shape in [(90, 40), (10, 30), (1, 1), (64, 64)]:
        A = stair_front(*shape, rng)
        f = ss.Front(0, ss.EMPTY, *shape)
        f.F = A
        assert np.array_equal(ss.GPUQREngine_FindStaircase(f), oracle.find_staircase(A))


def _cases():
    yield "2d-12", problems.stacked_ls((12, 12)), problems.nested_dissection((12, 12), leaf=3)
    yield "2d-20x9", problems.stacked_ls((20, 9)), problems.nested_dissection((20, 9), leaf=4)
    yield "3d-6", problems.stacked_ls((6, 6, 6)), problems.nested_dissection((6, 6, 6), leaf=2)
    rng = np.random.default_rng(4)
    R = sp.random(60, 35, density=0.08, random_state=4, format="csc") + sp.vstack(
        [sp.identity(35), sp.csc_matrix((25, 35))])
    yield "random", sp.csc_matrix(R), rng.permutation(35)


@pytest.mark.parametrize("name,A,Q", list(_cases()), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("relax", [0, 20])
def test_symbolic_consistency(name, A, Q, relax):
    sym = spqr.analyze(A, Q, relax=relax, gpuMemorySize=2**33)
    n, nf = sym.n, sym.nf
    assert sorted(sym.Qfill.tolist()) == list(range(n))
    # column counts against a brute-force symbolic Cholesky of (AQ)'(AQ)
    AQ = A[:, sym.Qfill].toarray()
    pat = (AQ.T @ AQ) != 0
    for k in range(n):
        nz = np.nonzero(pat[k + 1:, k])[0] + k + 1
        pat[np.ix_(nz, nz)] = True
    cc = 1 + np.array([pat[k + 1:, k].sum() for k in range(n)])
    assert np.array_equal(cc, sym.ColCount)
    # every front pattern contains the pattern of each of its pivot columns' R row
    for f in range(nf):
        cols = set(sym.Rj[sym.Rp[f]:sym.Rp[f + 1]].tolist())
        for j in range(sym.Super[f], sym.Super[f + 1]):
            assert set(np.nonzero(pat[j:, j])[0] + j) | {j} <= cols
        if relax == 0:
            assert len(cols) == cc[sym.Super[f]]
    # tree conventions of spqr_analyze
    assert sym.Parent[nf] == -1 and sym.Post[nf] == nf and sym.Childp[nf + 1] == nf
    assert all(sym.Parent[f] > f for f in range(nf))
    assert sym.Sleft[n + 1] == sym.m and sym.Sp[sym.m] == sym.anz


@pytest.mark.parametrize("name,A,Q", list(_cases()), ids=[c[0] for c in _cases()])
def test_assembly_maps_bit_exact(name, A, Q):
    sym = spqr.analyze(A, Q, gpuMemorySize=2**33)
    Sx = spqr.s_values(sym, A)
    got = spqr.spqrgpu_buildAssemblyMaps(sym, Sx)
    ref = oracle.build_assembly_maps(sym, Sx)
    for g, r in zip(got, ref):
        assert g.tobytes() == r.tobytes()
    # Stair ends one past the last row of each column and the last one is fm
    Stair = got[0]
    for f in range(sym.nf):
        st = Stair[sym.Rp[f]:sym.Rp[f + 1]]
        assert np.all(np.diff(st) >= 0) and st[-1] == sym.Fm[f]


def test_assembly_maps_of_many_fronts_on_several_threads_bit_exact():
    """From 4096 fronts on the maps are built by several host threads, front by front: same arrays, bit
    for bit, as the oracle's single loop (and as a second run: no order dependence)."""
    shape = (300, 300)
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=3), relax=0, gpuMemorySize=2**36)
    assert sym.nf >= 4096
    Sx = spqr.s_values(sym, A)
    got = spqr.spqrgpu_buildAssemblyMaps(sym, Sx)
    ref = oracle.build_assembly_maps(sym, Sx)
    again = spqr.spqrgpu_buildAssemblyMaps(sym, Sx)
    for g, r, a in zip(got, ref, again):
        assert g.tobytes() == r.tobytes() and a.tobytes() == r.tobytes()


@pytest.mark.parametrize("name,A,Q", list(_cases()), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("mem", [0, 1, 2**21, 2**22 + 12345, 11_000_000, 2**33])
def test_staging_bit_exact(name, A, Q, mem):
    sym = spqr.analyze(A, Q, gpuMemorySize=2**33)
    args = (sym.nf, sym.Parent, sym.Childp, sym.Child, sym.Fm, sym.Cm, sym.Rp, sym.Sp, sym.Sleft, sym.Super,
            sym.Post, sym.RimapSize, sym.RjmapSize, mem)
    got = spqr.spqrgpu_computeFrontStaging(*args)
    ref = oracle.compute_front_staging(*args)
    assert got["feasible"] == ref["feasible"] and got["numStages"] == ref["numStages"]
    if got["feasible"]:
        ns, nf = got["numStages"], sym.nf
        for key in ("Stagingp",):
            assert np.array_equal(got[key][:ns + 1], ref[key][:ns + 1])
        for key in ("FSize", "RSize", "SSize"):
            assert np.array_equal(got[key][:ns], ref[key][:ns])
        for key in ("StageMap", "FOffsets", "ROffsets", "SOffsets"):
            assert np.array_equal(got[key][:nf], ref[key][:nf])


def test_staging_quirks_recorded_from_the_reference():
    """Outcomes observed with the reference's own function (tiny memory -> empty stage 0 and one
    front per stage; gpuMemorySize = 1 -> infeasible; huge memory -> one stage)."""
    A = problems.stacked_ls((12, 12))
    Q = problems.nested_dissection((12, 12), leaf=3)
    sym = spqr.analyze(A, Q, gpuMemorySize=2**33)
    args = (sym.nf, sym.Parent, sym.Childp, sym.Child, sym.Fm, sym.Cm, sym.Rp, sym.Sp, sym.Sleft, sym.Super,
            sym.Post, sym.RimapSize, sym.RjmapSize)
    one = spqr.spqrgpu_computeFrontStaging(*args, 10**10)
    assert one["feasible"] and one["numStages"] == 1
    assert one["SSize"][0] == sym.anz
    zero = spqr.spqrgpu_computeFrontStaging(*args, 0)
    assert zero["feasible"] and list(zero["Stagingp"][:4]) == [0, 0, 1, 2]
    bad = spqr.spqrgpu_computeFrontStaging(*args, 1)
    assert not bad["feasible"] and bad["numStages"] == 0


def test_oracle_multifrontal_matches_dense_qr():
    """R from the CPU multifrontal oracle equals LAPACK's R of A(:,Qfill) up to row signs."""
    for name, A, Q in _cases():
        sym = spqr.analyze(A, Q, gpuMemorySize=2**33)
        Farena, FOff = oracle.multifrontal(sym, spqr.s_values(sym, A))
        R = oracle.r_from_fronts(sym, Farena, FOff)
        Rl = np.linalg.qr(A[:, sym.Qfill].toarray(), mode="r")
        assert R.shape[0] == sym.n
        assert r_rel_diff(rownorm_sign(R), rownorm_sign(Rl)) <= 1e-12, name


def test_flopcount_formula():
    """flopcount = sum over fronts/columns of 3h + 4h(fn-j-1) (spqr_analyze.cpp:888-902); for a
    dense m-by-n front that is about 2n^2(m - n/3)."""
    A = sp.csc_matrix(np.ones((50, 20)))
    sym = spqr.analyze(A, None)
    assert sym.nf == 1 and sym.Fm[0] == 50
    m, n = 50, 20
    want = sum(3 * (m - j) + 4 * (m - j) * (n - j - 1) for j in range(n))
    assert sym.flopcount == want


def test_task_flop_counts_are_the_references_formulas():
    """gpuqr_flops_factorize / gpuqr_flops_apply against the oracle's restatement of flopsFactorizeVT,
    flopsFactorize and flopsApply (GPUQREngine/Source/TaskDescriptor/TaskDescriptor_flops.cpp:18-53) on
    the reference's tile shapes, and gpuqr_front_flops as their sum over one front's tasks."""
    L = _lib.lib()
    O = oracle.lib()
    for f in (L.gpuqr_flops_factorize, L.gpuqr_flops_apply, L.gpuqr_front_flops, O.oracle_flops_factorize_vt,
              O.oracle_flops_factorize, O.oracle_flops_apply):
        f.restype = C.c_int64
    L.gpuqr_flops_factorize.argtypes = [C.c_int64, C.c_int64, C.c_int]
    L.gpuqr_flops_apply.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    O.oracle_flops_factorize_vt.argtypes = [C.c_int]
    O.oracle_flops_factorize.argtypes = [C.c_int, C.c_int]
    O.oracle_flops_apply.argtypes = [C.c_int, C.c_int]
    for tiles in (1, 2, 3):
        assert L.gpuqr_flops_factorize(32 * tiles, 32, 1) == O.oracle_flops_factorize_vt(tiles)
        for n in (32, 64, 2048, 16384):
            assert L.gpuqr_flops_apply(32 * tiles, 32, n) == O.oracle_flops_apply(tiles, n)
    for m, n in [(96, 32), (64, 32), (32, 32), (17, 9), (9, 17), (72, 64), (256, 32), (1, 1)]:
        assert L.gpuqr_flops_factorize(m, n, 0) == O.oracle_flops_factorize(m, n)
    # the paper's numbers for the reference's tasks (TaskDescriptor_flops.cpp evaluated by hand)
    assert O.oracle_flops_factorize_vt(3) == 2 * 95 + 32 * (6 + 4 * 96 * 32 + 160 + 192) + (-128 - 192 - 5) * 528 + 3 * 11440 - 2 * 63
    assert O.oracle_flops_apply(3, 64) == 32 * 64 * (4 * 96 - 29)
    # one 256 x 32 front: a single panel task without T
    L.gpuqr_front_flops.argtypes = [C.c_int64, C.c_int64, C.c_void_p]
    assert L.gpuqr_front_flops(256, 32, None) == L.gpuqr_flops_factorize(256, 32, 0)
    # 256 x 64: panel 0 with T (256 rows), its update of 32 columns, panel 1 on rows 32.. without T
    assert L.gpuqr_front_flops(256, 64, None) == (L.gpuqr_flops_factorize(256, 32, 1) + L.gpuqr_flops_apply(256, 32, 32)
                                                  + L.gpuqr_flops_factorize(224, 32, 0))


# ---------------------------------------------------------------- C ABI surface

def test_library_exports_every_declared_symbol():
    names = []
    for hdr in ("gpuqrengine_amd.h", "spqrgpu_amd.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b((?:gpuqr|spqrgpu_amd|spqr_amd)_[a-z_]+)\s*\(", txt)
    names = sorted(set(names))
    assert len(names) >= 15
    L = C.CDLL(_lib.LIB_PATH)
    for nme in names:
        assert hasattr(L, nme), nme


def test_error_codes_without_compute():
    L = _lib.lib()
    assert L.gpuqr_version() >= 1000
    assert L.gpuqr_factor_dense(0, None, 0, None) == _lib.GPUQR_SUCCESS
    assert L.gpuqr_factor_sparse(0, None, 3, None, None, None, None) == _lib.GPUQR_GPUERROR
    assert L.gpuqr_front_flops(4096, 2048, None) > 2 * 2048 ** 2 * (4096 - 2048 / 3)


def test_cpp_shim_compiles_and_links():
    """A C++ program written against the reference's header names (Front, SparseMeta, the two
    GPUQREngine overloads) builds against include/GPUQREngine.hpp and links to the library."""
    import subprocess
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-s", "-C", d])
    out = subprocess.check_output([os.path.join(d, "demo_dense"), "linkcheck"]).decode()
    assert "compiled and linked" in out


def test_symbolic_from_arrays_round_trip_and_validation():
    """The constructor an SPQR build uses (arrays of QRsym / QRsym->QRgpu in, symbolic object out):
    every array comes back as given; inconsistent input is refused; nothing here needs a GPU."""
    A = problems.stacked_ls((10, 10))
    sym = spqr.analyze(A, problems.nested_dissection((10, 10), leaf=3), gpuMemorySize=2**33)
    sym2 = spqr.symbolic_from_arrays(sym)
    assert (sym2.m, sym2.n, sym2.anz, sym2.nf, sym2.maxfn, sym2.rnz) == (sym.m, sym.n, sym.anz, sym.nf, sym.maxfn, sym.rnz)
    nf = sym.nf
    for name, ln in (("Sp", sym.m + 1), ("Sj", sym.anz), ("Sleft", sym.n + 2), ("Super", nf + 1), ("Rp", nf + 1),
                     ("Rj", int(sym.Rp[nf])), ("Parent", nf + 1), ("Childp", nf + 2), ("Child", nf + 1), ("Post", nf),
                     ("Fm", nf), ("Cm", nf), ("RimapOffsets", nf), ("RjmapOffsets", nf), ("StageMap", nf),
                     ("Stagingp", sym.numStages + 1)):
        assert np.array_equal(getattr(sym2, name)[:ln], getattr(sym, name)[:ln]), name
    assert (sym2.RimapSize, sym2.RjmapSize, sym2.numStages) == (sym.RimapSize, sym.RjmapSize, sym.numStages)

    class Bad:
        pass
    bad = Bad()
    bad.__dict__.update({k: getattr(sym, k) for k in ("m", "n", "anz", "nf", "RimapSize", "RjmapSize", "numStages",
                                                      "Sp", "Sj", "Sleft", "Super", "Rp", "Rj", "Parent", "Childp", "Child",
                                                      "Post", "Fm", "Cm", "RimapOffsets", "RjmapOffsets", "Stagingp",
                                                      "StageMap")})
    bad.Parent = sym.Parent.copy()
    bad.Parent[0] = 0                                   # a front cannot be its own parent
    with pytest.raises(ValueError):
        spqr.symbolic_from_arrays(bad, 2**33)
    bad.Parent = sym.Parent
    bad.Cm = sym.Cm.copy()
    bad.Cm[0] = 10**6                                   # a contribution block taller than its front
    with pytest.raises(ValueError):
        spqr.symbolic_from_arrays(bad, 2**33)
    # (argument validation happens before any device call)
    assert spqr.spqrgpu_kernel(sym2, None)["code"] == 3


def test_appended_rhs_with_structured_zeros_stays_last():
    """Two right-hand sides, each supported on one block of a block-diagonal A: with their zeros
    dropped, the postorder of the column etree puts a column of B among A's; stored dense they stay
    last (the reference never orders B's columns, SPQR/Source/SuiteSparseQR.cpp:256-260)."""
    blk = problems.stacked_ls((4, 4))
    A = sp.block_diag([blk, blk], format="csc")
    m, n = A.shape
    B = np.zeros((m, 2))
    B[:m // 2, 0] = 1.0
    B[m // 2:, 1] = 2.0
    AB, Qab = spqr.append_rhs(A, B)
    sym = spqr.analyze(AB, Qab, gpuMemorySize=2**32)
    assert np.array_equal(np.sort(sym.Qfill[:n]), np.arange(n))
    assert set(sym.Qfill[n:]) == {n, n + 1}
    # the failure this guards against is real: with the zeros dropped a column of B moves forward
    ABs = sp.hstack([A, sp.csc_matrix(B)], format="csc")
    syms = spqr.analyze(ABs, Qab, gpuMemorySize=2**32)
    assert not np.array_equal(np.sort(syms.Qfill[:n]), np.arange(n))


WEST0067 = "/root/reference/UMFPACK/Demo/HB/west0067.rua"


@pytest.mark.skipif(not os.path.exists(WEST0067), reason="the reference tree is not on this machine")
def test_qrdemo_west0067_rank_and_residual():
    """BASELINE config 1 on the one bundled matrix the reference tree still holds, read in place
    (never copied): analysis, the oracle's multifrontal QR, least squares through an appended b = ones
    as SPQR/Demo/qrdemo.cpp:87-105 sets it up, and qrdemo's own check (qrdemo.cpp:22-70):
    rank 67 and ||Ax-b||_2 / (||A||_1 ||x||_2) at rounding level (a build of the reference printed
    6.8e-17 for its default ordering)."""
    import scipy.io
    A = sp.csc_matrix(scipy.io.hb_read(WEST0067))
    m, n = A.shape
    assert (m, n, A.nnz) == (67, 67, 294)
    b = np.ones((m, 1))
    AB, Qab = spqr.append_rhs(A, b)
    sym = spqr.analyze(AB, Qab, gpuMemorySize=2**32)
    rank = int(sum(min(sym.Fm[f], sym.Super[f + 1] - sym.Super[f]) for f in range(sym.nf)))
    assert rank == 67
    Farena, FOff = oracle.multifrontal(sym, spqr.s_values(sym, AB))
    R = oracle.r_from_fronts(sym, Farena, FOff)
    assert np.abs(np.diag(R[:n, :n])).min() > 1e-3          # numerically full rank as well
    x = np.zeros(n)
    x[sym.Qfill[:n]] = np.linalg.solve(R[:n, :n], R[:n, n])
    r = A @ x - b[:, 0]
    rel = np.linalg.norm(r) / (abs(A).sum(axis=0).max() * np.linalg.norm(x))
    assert rel < 5e-16
    # R is the factor of A(:,Qfill): unique up to row signs, so LAPACK must agree
    Rl = np.linalg.qr(A[:, sym.Qfill[:n]].toarray(), mode="r")
    assert np.abs(np.abs(R[:n, :n]) - np.abs(Rl)).max() <= 1e-12 * np.abs(Rl).max()


def test_oracle_rank_detection_against_lapack():
    """The oracle's restatement of SPQR's dead-column rule (spqr_front.cpp:424-479) on config 5 at a
    small size: exactly the duplicated and the nearly dependent columns die at SPQR_DEFAULT_TOL, the
    live columns' part of the squeezed R is LAPACK's R of A(:,live) (unique up to row signs), the dead
    columns hold Q_live' a_k cut at their would-be pivot row, and R'R = (AQ)'(AQ) to the tolerance."""
    A, dup, near = problems.rank_deficient_ls(24, 12, 12)
    Q = problems.nested_dissection((24, 24), leaf=4)
    sym = spqr.analyze(A, Q, relax=20, gpuMemorySize=2**34, rank_detection=True)
    n = sym.n
    tol = spqr.default_tol(A)
    assert 0 < tol < 1e-9
    R, Rdead, rank, Rank, enorm = oracle.multifrontal_tol(sym, spqr.s_values(sym, A), tol)
    assert rank == n - 24 and int(Rdead.sum()) == 24 and enorm < 24 * tol
    # a dead column is one of each dependent pair (whichever the ordering reaches second)
    pairs = dup + near
    inv = np.argsort(sym.Qfill)
    dead_orig = set(sym.Qfill[np.nonzero(Rdead)[0]].tolist())
    assert all((a in dead_orig) != (b in dead_orig) for a, b in pairs)
    R = R.toarray()
    AQ = A[:, sym.Qfill].toarray()
    live = np.nonzero(Rdead == 0)[0]
    Rl = np.linalg.qr(AQ[:, live], mode="r")
    assert np.abs(np.abs(R[:, live]) - np.abs(Rl)).max() <= 1e-13 * np.abs(Rl).max()
    G = AQ.T @ AQ
    assert np.linalg.norm(R.T @ R - G, 1) <= 1e-12 * np.linalg.norm(G, 1)
    # squeezed: column j of R has nothing below the number of live pivots up to and including it
    nlive = np.cumsum(Rdead == 0)
    for j in range(n):
        assert not R[nlive[j]:, j].any()
    # without a tolerance nothing dies
    assert oracle.multifrontal_tol(sym, spqr.s_values(sym, A), -1.0)[2] == n
    _ = inv


def test_rsolve_on_packed_blocks_without_a_gpu():
    """spqr_amd_rsolve is host code: the packed R blocks of a CPU factorization (the oracle's fronts,
    packed as SPQR's stack holds them) wrapped in a spqr_amd_numeric, C = R^-T (A Q)' b, and the back
    substitution front by front must give the least-squares solution in A's column order; with Qfill
    left out it gives it in R's column order."""
    import ctypes as C
    from scipy.sparse.linalg import spsolve_triangular
    from mgpu_worker import oracle_factor_subset
    from suitesparse_amd import multigpu
    A = problems.stacked_ls((9, 9))
    m, n = A.shape
    sym = spqr.analyze(A, problems.nested_dissection((9, 9), leaf=3), relax=4, gpuMemorySize=2**33)
    nf = sym.nf
    Rb, _, _ = oracle_factor_subset(sym, A, np.ones(nf, bool), {})
    R = multigpu.blocks_to_csc(sym, Rb)
    num = spqr.Numeric()
    Rblock = np.zeros(nf, np.int64)
    off = 0
    for f in range(nf):
        Rblock[f] = off
        off += Rb[f].size
    Rstack = np.concatenate([Rb[f] for f in range(nf)])
    Rdead = np.zeros(n, np.int8)
    num.nf, num.rsize = nf, Rstack.size
    num.Rstack = Rstack.ctypes.data_as(C.POINTER(C.c_double))
    num.Rblock = Rblock.ctypes.data_as(C.POINTER(C.c_int64))
    num.Rdead = Rdead.ctypes.data_as(type(num.Rdead))
    num.sumfrank = n
    num.maxfrank = int(max(min(int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f])) for f in range(nf)))

    class Res:
        _num = C.pointer(num)

    rng = np.random.default_rng(2)
    b = rng.standard_normal((m, 3))
    AQ = A[:, sym.Qfill]
    Cq = spsolve_triangular(R.T.tocsr(), AQ.T @ b, lower=True)
    x = spqr.rsolve(sym, Res, Cq)
    xl = np.linalg.lstsq(A.toarray(), b, rcond=None)[0]
    assert np.abs(x - xl).max() <= 1e-11 * np.abs(xl).max()
    y = spqr.rsolve(sym, Res, Cq, permute=False)
    assert np.array_equal(y, x[sym.Qfill])
    # R v and R' v on the same blocks, and the vector form of check_r_factor built from them
    Res.rank = n
    v = rng.standard_normal(n)
    assert np.abs(spqr.rmult(sym, Res, v) - R @ v).max() <= 1e-13 * np.abs(R @ v).max()
    assert np.abs(spqr.rmult(sym, Res, v, transposed=True) - R.T @ v).max() <= 1e-13 * np.abs(R.T @ v).max()
    assert spqr.gram_probe(sym, Res, A) <= 1e-13
    # R' \ B, and with it the seminormal equations R'R x = A'b
    g = rng.standard_normal((n, 2))
    wt = spqr.rtsolve(sym, Res, g)
    assert np.abs(R.T @ wt - g).max() <= 1e-12 * np.abs(g).max()
    xs = np.zeros((n, 3))
    xs[sym.Qfill] = spqr.rsolve(sym, Res, spqr.rtsolve(sym, Res, (A.T @ b)[sym.Qfill]), permute=False)
    assert np.abs(xs - xl).max() <= 1e-10 * np.abs(xl).max()
    # a numeric object that holds only a part of R is refused
    Rblock[nf // 2] = -1
    with pytest.raises(RuntimeError):
        spqr.rsolve(sym, Res, Cq)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_code_under_address_and_ub_sanitizers():
    """tests/cpp/host_asan.cpp: every host-only entry point of the library (analysis in both modes,
    staging, assembly maps, planner counts, symbolic object from arrays and its validation, R conversion,
    products and triangular solves on packed blocks) built with -fsanitize=address,undefined on the host
    side and run without a GPU.  Any report of either sanitizer ends the program with a non-zero status."""
    import subprocess
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-s", "-C", d, "host_asan"], check=True, timeout=900)
    r = subprocess.run([os.path.join(d, "host_asan")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "all checks passed" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_starts_its_own_ranks_and_fails_loudly_without_a_gpu():
    """python bench.py --gpus 2 with no launcher around it starts two ranks itself (each says so before it
    touches a device); on a machine without a GPU both then fail, and so does the command: there is no
    CPU path to fall back to.  (With a GPU: tests/test_bench_gpu.py.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU behaviour")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "sparse3d:8",
                        "--steps", "1", "--warmup", "1", "--extras", "none", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert "rank 0 of 2 started" in r.stderr and "rank 1 of 2 started" in r.stderr, r.stderr[-2000:]
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
