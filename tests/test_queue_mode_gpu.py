"""The persistent work-queue kernel runs the same tasks as the per-round launches: its results must
agree with theirs to rounding (1e-12 of the largest entry of R; its panels sum their columns over
8 row groups instead of 16), for dense batches and for the sparse path, and with LAPACK."""
import numpy as np
import pytest

import suitesparse_amd as ss
from suitesparse_amd import _lib, problems, spqr

pytestmark = pytest.mark.gpu


@pytest.fixture
def queue_mode():
    L = _lib.lib()

    def set_mode(m):
        L.gpuqr_set_mode(m)
    yield set_mode
    L.gpuqr_set_mode(0)


def _dense(mats):
    fronts = []
    for k, A in enumerate(mats):
        m, n = A.shape
        f = ss.Front(k, ss.EMPTY, m, n)
        f.F = np.ascontiguousarray(A)
        f.cpuR = np.zeros((min(m, n), n), dtype=np.float64)
        fronts.append(f)
    stats = ss.QREngineStats()
    assert ss.GPUQREngine(0, fronts, len(fronts), stats=stats) == ss.QRENGINE_SUCCESS
    return [np.array(f.cpuR) for f in fronts], stats


# The queue kernel factorizes panels with 8 row groups per column (256-thread workgroups), the
# stand-alone panel kernels with 16: the sums over a column associate differently, so the two
# modes agree to rounding, not bit for bit.  Tolerance: 1e-12 of the largest entry of R.
def _same_to_rounding(a, b, tol=1e-12):
    a = np.asarray(a) ; b = np.asarray(b)
    return a.shape == b.shape and (a.size == 0 or np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max()))


def _same_sparse_to_rounding(A, B, tol=1e-12):
    return A.shape == B.shape and (A.nnz == 0 or abs(A - B).max() <= tol * max(1.0, abs(A).max()))


def test_dense_batch_equal_to_rounding(queue_mode):
    rng = np.random.default_rng(11)
    shapes = [(1, 1), (40, 70), (96, 32), (300, 100), (257, 257), (1024, 384), (700, 900), (33, 31)]
    mats = [rng.standard_normal((m, n)) for m, n in shapes]
    queue_mode(0)
    R0, s0 = _dense(mats)
    queue_mode(1)
    R1, s1 = _dense(mats)
    assert s1.numLaunches == 1 and s0.numLaunches > 1
    assert s0.flopsActual == s1.flopsActual
    for a, b in zip(R0, R1):
        assert _same_to_rounding(a, b)
    # and against LAPACK: R is unique up to the signs of its rows
    def rows_positive(R):
        sgn = np.sign(np.diag(R)) ; sgn[sgn == 0] = 1
        return R * sgn[:, None]
    for A, b in zip(mats, R1):
        Rl = rows_positive(np.linalg.qr(A, mode="r"))
        assert np.abs(rows_positive(np.triu(b)) - Rl).max() <= 1e-11 * max(1.0, np.abs(Rl).max())


@pytest.mark.parametrize("grid,leaf", [((40, 40), 4), ((9, 9, 9), 2)])
def test_sparse_equal_to_rounding(queue_mode, grid, leaf):
    A = problems.stacked_ls(grid)
    sym = spqr.analyze(A, problems.nested_dissection(grid, leaf=leaf), relax=20, gpuMemorySize=2**34)
    queue_mode(0)
    r0 = spqr.factorize(sym, A)
    queue_mode(1)
    r1 = spqr.factorize(sym, A)
    assert r0.rank == r1.rank == sym.n
    assert _same_sparse_to_rounding(r0.R, r1.R)


def test_staged_sparse_in_queue_mode(queue_mode):
    grid = (40, 40)
    A = problems.stacked_ls(grid)
    Q = problems.nested_dissection(grid, leaf=4)
    one = spqr.factorize(spqr.analyze(A, Q, relax=0, gpuMemorySize=2**34), A)
    queue_mode(1)
    sym = spqr.analyze(A, Q, relax=0, gpuMemorySize=10_700_000)
    assert sym.numStages > 1
    res = spqr.factorize(sym, A)
    assert res.numStagesRun == sym.numStages
    assert _same_sparse_to_rounding(one.R, res.R)


def test_auto_mode_picks_queue_for_a_chain_and_launches_for_a_batch(queue_mode):
    rng = np.random.default_rng(3)
    chain = [rng.standard_normal((700, 500))]                     # one front: a long chain of small launches
    batch = [rng.standard_normal((96, 32)) for _ in range(3000)]  # one full launch
    queue_mode(0)
    Rc0, sc0 = _dense(chain)
    Rb0, sb0 = _dense(batch)
    queue_mode(2)
    Rc2, sc2 = _dense(chain)
    Rb2, sb2 = _dense(batch)
    assert sc2.numLaunches == 1 and sc0.numLaunches > 1
    assert sb2.numLaunches == sb0.numLaunches
    assert all(_same_to_rounding(a, b) for a, b in zip(Rc0, Rc2))      # the queue ran the chain
    assert all(np.array_equal(a, b) for a, b in zip(Rb0, Rb2))         # launches both times
