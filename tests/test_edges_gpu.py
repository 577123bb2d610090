"""Edge cases of the engine entry points on the device: degenerate fronts, invalid descriptors, a
failing host allocator (the reference's test suite injects malloc failures the same way,
SPQR/Tcov/qrtest.cpp), and rows padded by the caller."""
import ctypes as C

import numpy as np
import pytest

import suitesparse_amd as ss
from suitesparse_amd import _lib
from oracle import oracle
from tests.util import r_rel_diff, rownorm_sign

pytestmark = pytest.mark.gpu


def _device_fronts(mats, pitch=None):
    import torch
    dev = torch.device("cuda:0")
    fronts, bufs = [], []
    for k, A in enumerate(mats):
        m, n = A.shape
        ld = n if pitch is None else pitch[k]
        P = np.zeros((max(m, 1), max(ld, 1)))
        P[:m, :n] = A
        t = torch.from_numpy(P).to(dev)
        f = ss.Front(k, ss.EMPTY, m, n)
        f.ldn = ld
        f.gpuF = t.data_ptr()
        fronts.append(f)
        bufs.append(t)
    return fronts, bufs


def test_degenerate_fronts_in_a_batch():
    rng = np.random.default_rng(5)
    shapes = [(0, 5), (5, 0), (1, 1), (37, 3), (2, 300), (0, 0), (64, 64)]
    mats = [rng.standard_normal(s) for s in shapes]
    fronts, bufs = _device_fronts(mats)
    assert ss.GPUQREngine(0, fronts, len(fronts)) == ss.QRENGINE_SUCCESS
    import torch
    torch.cuda.synchronize()
    for A, t in zip(mats, bufs):
        m, n = A.shape
        if m == 0 or n == 0:
            continue
        R = np.triu(t.cpu().numpy()[:min(m, n), :n])
        Rref = np.triu(oracle.front_qr(A)[:min(m, n)])
        assert r_rel_diff(R, Rref) <= 1e-12


def test_caller_padded_rows_give_the_same_r():
    rng = np.random.default_rng(6)
    shapes = [(300, 77), (129, 129), (40, 203), (513, 31)]
    mats = [rng.standard_normal(s) for s in shapes]
    f0, b0 = _device_fronts(mats)
    f1, b1 = _device_fronts(mats, pitch=[n + 1 + (n & 1) * 0 + 2 * k for k, (_, n) in enumerate(shapes)])
    assert ss.GPUQREngine(0, f0, len(f0)) == ss.QRENGINE_SUCCESS
    assert ss.GPUQREngine(0, f1, len(f1)) == ss.QRENGINE_SUCCESS
    import torch
    torch.cuda.synchronize()
    for (m, n), t0, t1 in zip(shapes, b0, b1):
        k = min(m, n)
        R0 = np.triu(t0.cpu().numpy()[:k, :n])
        R1 = np.triu(t1.cpu().numpy()[:k, :n])
        assert np.array_equal(R0, R1)
        assert np.all(t1.cpu().numpy()[:, n:] == 0)          # the padding is never written


def test_invalid_descriptors_are_refused():
    rng = np.random.default_rng(7)
    fronts, bufs = _device_fronts([rng.standard_normal((8, 8)), rng.standard_normal((8, 8))])
    fronts[1].fids = 5                                   # fids must equal the array index
    assert ss.GPUQREngine(0, fronts, 2) == ss.QRENGINE_GPUERROR
    fronts[1].fids = 1
    fronts[1].pids = 1                                   # a front cannot be its own parent
    fronts[1].sparseMeta.isSparse = True
    assert ss.GPUQREngine(0, fronts, 2, Parent=[0], Childp=[0], Child=[0]) == ss.QRENGINE_GPUERROR
    fronts[1].pids = ss.EMPTY
    fronts[1].sparseMeta.isSparse = False
    fronts[0].gpuF = 0                                   # no device buffer and no host buffer
    L = _lib.lib()
    arr = (_lib.FrontC * 1)()
    arr[0].fids, arr[0].pids, arr[0].fm, arr[0].fn, arr[0].fp, arr[0].rank = 0, -1, 4, 4, 4, 4
    assert L.gpuqr_factor_sparse(0, arr, 1, None, None, None, None) == _lib.GPUQR_GPUERROR


def test_failing_host_allocator_reports_out_of_memory():
    L = _lib.lib()
    MALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t)
    CALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_size_t)
    FREE = C.CFUNCTYPE(None, C.c_void_p)
    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = C.c_void_p, [C.c_size_t]
    libc.calloc.restype, libc.calloc.argtypes = C.c_void_p, [C.c_size_t, C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    budget = {"n": 0}

    def my_malloc(sz):
        budget["n"] -= 1
        return None if budget["n"] < 0 else libc.malloc(sz)

    def my_calloc(a, b):
        budget["n"] -= 1
        return None if budget["n"] < 0 else libc.calloc(a, b)

    cbs = (MALLOC(my_malloc), CALLOC(my_calloc), FREE(lambda p: libc.free(p)))
    L.gpuqr_set_allocator.argtypes = [MALLOC, CALLOC, FREE]
    rng = np.random.default_rng(8)
    A = rng.standard_normal((200, 80))
    codes = set()
    try:
        for allowed in range(0, 12):
            fronts, bufs = _device_fronts([A])
            budget["n"] = allowed
            L.gpuqr_set_allocator(*cbs)
            codes.add(ss.GPUQREngine(0, fronts, 1))
            L.gpuqr_set_allocator(MALLOC(0), CALLOC(0), FREE(0))
    finally:
        L.gpuqr_set_allocator(MALLOC(0), CALLOC(0), FREE(0))
    assert ss.QRENGINE_OUTOFMEMORY in codes          # early allocations refused
    assert codes <= {ss.QRENGINE_OUTOFMEMORY, ss.QRENGINE_SUCCESS}
    # and the engine still works afterwards
    fronts, bufs = _device_fronts([A])
    assert ss.GPUQREngine(0, fronts, 1) == ss.QRENGINE_SUCCESS


def test_failing_host_allocator_reaches_the_spqr_glue():
    """The reference routes every host allocation through SuiteSparse_config's table so that its
    failing-allocator tests bite (SPQR/Tcov/qrtest.cpp:58-140).  Here: the numeric driver under an
    allocator that refuses after N calls returns out-of-memory (1) for small N, succeeds for large N,
    leaves no half-built cache behind, and the result afterwards is the usual one."""
    from suitesparse_amd import problems, spqr
    L = _lib.lib()
    MALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t)
    CALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_size_t)
    FREE = C.CFUNCTYPE(None, C.c_void_p)
    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = C.c_void_p, [C.c_size_t]
    libc.calloc.restype, libc.calloc.argtypes = C.c_void_p, [C.c_size_t, C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    budget = {"n": 0, "calls": 0}

    def my_malloc(sz):
        budget["n"] -= 1
        budget["calls"] += 1
        return None if budget["n"] < 0 else libc.malloc(sz)

    def my_calloc(a, b):
        budget["n"] -= 1
        budget["calls"] += 1
        return None if budget["n"] < 0 else libc.calloc(a, b)

    cbs = (MALLOC(my_malloc), CALLOC(my_calloc), FREE(lambda p: libc.free(p)))
    L.gpuqr_set_allocator.argtypes = [MALLOC, CALLOC, FREE]
    A = problems.stacked_ls((12, 12))
    Q = problems.nested_dissection((12, 12), leaf=3)
    good = spqr.factorize(spqr.analyze(A, Q, gpuMemorySize=2**33), A)
    codes = []
    try:
        for allowed in list(range(0, 40, 3)) + [10**6]:
            sym = spqr.analyze(A, Q, gpuMemorySize=2**33)        # a fresh object: the pattern cache is rebuilt
            budget["n"], budget["calls"] = allowed, 0
            L.gpuqr_set_allocator(*cbs)
            try:
                res = spqr.factorize(sym, A)
                codes.append(0)
                assert (res.R != good.R).nnz == 0
            except RuntimeError as e:
                codes.append(int(str(e).split()[-1]))
            finally:
                L.gpuqr_set_allocator(MALLOC(0), CALLOC(0), FREE(0))
            if allowed == 10**6:
                assert budget["calls"] > 20                  # the glue's allocations do go through the hooks
    finally:
        L.gpuqr_set_allocator(MALLOC(0), CALLOC(0), FREE(0))
    assert codes[0] == 1 and codes[-1] == 0
    assert set(codes) <= {0, 1}
    res = spqr.factorize(spqr.analyze(A, Q, gpuMemorySize=2**33), A)
    assert (res.R != good.R).nnz == 0


def test_failing_host_allocator_reaches_the_tolerance_path_and_the_back_substitution():
    """The same for the passes of spqrgpu_amd_factorize_tol (plans, reviews, the host-packed R of
    fronts with dead columns) and for spqr_amd_rsolve: out of memory (1) or the usual answer, never
    anything else, and nothing left half-built."""
    import scipy.sparse as sp
    from suitesparse_amd import problems, spqr
    L = _lib.lib()
    MALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t)
    CALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_size_t)
    FREE = C.CFUNCTYPE(None, C.c_void_p)
    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = C.c_void_p, [C.c_size_t]
    libc.calloc.restype, libc.calloc.argtypes = C.c_void_p, [C.c_size_t, C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    budget = {"n": 0}

    def my_malloc(sz):
        budget["n"] -= 1
        return None if budget["n"] < 0 else libc.malloc(sz)

    def my_calloc(a, b):
        budget["n"] -= 1
        return None if budget["n"] < 0 else libc.calloc(a, b)

    cbs = (MALLOC(my_malloc), CALLOC(my_calloc), FREE(lambda p: libc.free(p)))
    L.gpuqr_set_allocator.argtypes = [MALLOC, CALLOC, FREE]
    A, _, _ = problems.rank_deficient_ls(12, 5, 4)
    A = sp.csc_matrix(A)
    Q = problems.nested_dissection((12, 12), leaf=3)
    tol = spqr.default_tol(A)
    good = spqr.factorize(spqr.analyze(A, Q, gpuMemorySize=2**33, rank_detection=True), A, tol=tol, keep_num=True)
    assert good.numPasses >= 2 and good.rank < A.shape[1]
    sym0 = spqr.analyze(A, Q, gpuMemorySize=2**33, rank_detection=True)
    Cq = np.random.default_rng(1).standard_normal((good.rank, 2))
    x_good = spqr.rsolve(sym0, good, Cq)
    codes, rcodes = [], []
    try:
        for allowed in list(range(0, 120, 7)) + [10**6]:
            sym = spqr.analyze(A, Q, gpuMemorySize=2**33, rank_detection=True)
            budget["n"] = allowed
            L.gpuqr_set_allocator(*cbs)
            try:
                res = spqr.factorize(sym, A, tol=tol)
                codes.append(0)
                assert res.rank == good.rank and np.array_equal(res.Rdead, good.Rdead) and (res.R != good.R).nnz == 0
            except RuntimeError as e:
                codes.append(int(str(e).split()[-1]))
            finally:
                L.gpuqr_set_allocator(MALLOC(0), CALLOC(0), FREE(0))
        for allowed in (0, 1, 2, 3, 10**6):
            budget["n"] = allowed
            L.gpuqr_set_allocator(*cbs)
            try:
                x = spqr.rsolve(sym0, good, Cq)
                rcodes.append(0)
                assert np.array_equal(x, x_good)
            except RuntimeError as e:
                rcodes.append(int(str(e).split()[-1]))
            finally:
                L.gpuqr_set_allocator(MALLOC(0), CALLOC(0), FREE(0))
    finally:
        L.gpuqr_set_allocator(MALLOC(0), CALLOC(0), FREE(0))
        good.free()
    assert codes[0] == 1 and codes[-1] == 0 and set(codes) <= {0, 1}, codes
    assert rcodes[0] == 1 and rcodes[-1] == 0 and set(rcodes) <= {0, 1}, rcodes


def test_repeated_device_batches_reuse_the_schedule_and_follow_the_values():
    """The engine recognises a repeated problem description (same shapes, buffers, staircases) and
    replays its schedule; the factors must follow the new values, and a different batch in between
    must not be confused with it."""
    import torch
    rng = np.random.default_rng(8)
    shapes = [(300, 77), (129, 129), (40, 203), (513, 31), (700, 260)]
    mats = [rng.standard_normal(s) for s in shapes]
    fronts, bufs = _device_fronts(mats)
    other, _obufs = _device_fronts([rng.standard_normal((90, 40))])
    for trial in range(3):
        new = [rng.standard_normal(s) * (trial + 1) for s in shapes]
        for t, A in zip(bufs, new):
            t.copy_(torch.from_numpy(A))
        assert ss.GPUQREngine(0, fronts, len(fronts)) == ss.QRENGINE_SUCCESS
        torch.cuda.synchronize()
        for A, t in zip(new, bufs):
            m, n = A.shape
            R = np.triu(t.cpu().numpy()[:min(m, n), :n])
            Rref = np.triu(oracle.front_qr(A)[:min(m, n)])
            assert r_rel_diff(R, Rref) <= 1e-12
        if trial == 1:
            assert ss.GPUQREngine(0, other, 1) == ss.QRENGINE_SUCCESS


@pytest.mark.parametrize("switch", ["GPUQR_LOOKAHEAD_TILES=2", "GPUQR_LOOKAHEAD_TILES=8", "GPUQR_NO_PIPELINE=1",
                                    "GPUQR_SPLIT_TIERS=1", "GPUQR_RIDE_BELOW=0", "GPUQR_SMALL_ON=0",
                                    "GPUQR_ASM_JOIN_BELOW=0", "GPUQR_EVENT_RING=16", "GPUQR_PIPE_ROWS=128",
                                    "GPUQR_R_PIECES=128", "GPUQR_R_PIECES=0", "GPUQR_RCOPY_GROUP=8"])
def test_schedule_switches_in_a_fresh_process(tmp_path, switch):
    """The schedule switches change which tasks share a launch and which stream runs them, never the
    arithmetic of a task: a process started with one of them set must produce bit-identical R for a
    dense batch and a sparse problem (the look-ahead splits an update by columns, and columns are
    updated independently of each other).  The switches are read once per process, hence the child."""
    import os
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import knob_worker
    ref = knob_worker.run()
    out = str(tmp_path / "r.npz")
    name, value = switch.split("=")
    with open(str(tmp_path / "out.txt"), "w") as fo:
        rc = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "knob_worker.py"), out],
                            stdout=fo, stderr=subprocess.STDOUT, env=dict(os.environ, **{name: value}), timeout=150).returncode
    assert rc == 0, open(str(tmp_path / "out.txt")).read()[-3000:]
    z = np.load(out)
    for k in ref:
        if k.endswith("launches"):
            continue
        assert z[k].shape == ref[k].shape and np.array_equal(z[k], ref[k]), k
    if name == "GPUQR_LOOKAHEAD_TILES":
        assert int(z["dense_launches"]) > int(ref["dense_launches"])     # the switch did take effect
    if name == "GPUQR_NO_PIPELINE":
        assert int(z["sp_launches"]) > int(ref["sp_launches"])


def test_serial_debug_mode_gives_the_same_bits():
    """gpuqr_set_serial: every launch waited for and checked (the reference's -DFORCE_SYNCHRONIZE,
    Scheduler_LaunchKernel.cpp:55-58).  Same tasks in the same order per stream: R bit for bit."""
    from suitesparse_amd import _lib, problems, spqr
    A = problems.stacked_ls((14, 14, 14))
    Q = problems.nested_dissection((14, 14, 14), leaf=4)
    sym = spqr.analyze(A, Q, relax=-1, gpuMemorySize=2**33)
    R0 = spqr.factorize(sym, A).R
    L = _lib.lib()
    L.gpuqr_set_serial(1)
    try:
        R1 = spqr.factorize(sym, A).R
    finally:
        L.gpuqr_set_serial(0)
    assert (R0 != R1).nnz == 0


@pytest.mark.parametrize("name", ["5x0", "0x4", "0x0", "allzero5x3", "1x1", "onecol", "onerow", "emptyrows"])
def test_degenerate_sparse_matrices_through_the_numeric_driver(name):
    """No columns, no rows, no entries, one entry, one column, one row, rows without entries: the analysis
    and the numeric driver must come back with the structural rank, R of that many rows, and the dead
    columns flagged (the reference's CPU path treats most of these as column singletons before its
    multifrontal part and reports the same ranks: 0, 0, 0, 0, 1, 1, 1)."""
    import scipy.sparse as sp
    from suitesparse_amd import spqr
    cases = {
        "5x0": (sp.csc_matrix((5, 0)), 0), "0x4": (sp.csc_matrix((0, 4)), 0), "0x0": (sp.csc_matrix((0, 0)), 0),
        "allzero5x3": (sp.csc_matrix((5, 3)), 0), "1x1": (sp.csc_matrix(np.array([[2.5]])), 1),
        "onecol": (sp.csc_matrix(np.array([[1.], [2.], [0.], [3.]])), 1),
        "onerow": (sp.csc_matrix(np.array([[1., 0., 2., 3.]])), 1),
        "emptyrows": (sp.csc_matrix(np.array([[0., 0., 0.], [1., 2., 0.], [0., 0., 0.], [0., 3., 4.], [0., 0., 0.]])), 2),
    }
    A, rank = cases[name]
    m, n = A.shape
    sym = spqr.analyze(A, None, relax=-1, gpuMemorySize=2**30)
    res = spqr.factorize(sym, A)
    assert res.rank == rank
    assert res.R.shape == (rank, n)
    assert int(np.asarray(res.Rdead).sum()) == n - rank
    # the same through the tolerance path (nothing here is numerically doubtful: same rank, same dead columns)
    sym_rd = spqr.analyze(A, None, relax=-1, gpuMemorySize=2**30, rank_detection=True)
    res_rd = spqr.factorize(sym_rd, A, tol=max(spqr.default_tol(A), 0.0))
    assert res_rd.rank == rank and np.array_equal(np.asarray(res_rd.Rdead)[:n], np.asarray(res.Rdead)[:n])
    if rank > 0:
        # R'R = (A Qfill)'(A Qfill) on the live part: R is the factor of the columns in the order of Qfill
        AQ = A[:, sym.Qfill[:n]].toarray()
        R = res.R.toarray()
        live = np.nonzero(np.asarray(res.Rdead)[:n] == 0)[0]
        G = AQ.T @ AQ
        assert np.abs((R.T @ R)[np.ix_(live, range(n))] - G[np.ix_(live, range(n))]).max() <= 1e-13 * max(1.0, np.abs(G).max())
