"""GPU parity of the dense GPUQREngine overload against the CPU oracle and LAPACK."""
import numpy as np
import pytest

import suitesparse_amd as ss
from oracle import oracle
from tests.util import demo_rand_front, gram_resid, r_rel_diff, rownorm_sign, stair_front

pytestmark = pytest.mark.gpu


def run_dense(mats, stairs=None, col_major=False):
    fronts = []
    for k, A in enumerate(mats):
        m, n = A.shape
        f = ss.Front(k, ss.EMPTY, m, n, isColMajor=col_major)
        if col_major:
            f.F = np.asfortranarray(A)
            f.ldn = m
            f.cpuR = np.zeros((m, n), dtype=np.float64, order="F")     # R(i,j) at cpuR[j*ldn + i]
        else:
            f.F = np.ascontiguousarray(A)
            f.cpuR = np.zeros((min(m, n), n), dtype=np.float64)
        if stairs is not None:
            f.Stair = stairs[k]
        fronts.append(f)
    stats = ss.QREngineStats()
    code = ss.GPUQREngine(0, fronts, len(fronts), stats=stats)
    assert code == ss.QRENGINE_SUCCESS
    return [np.array(f.cpuR[:min(f.fm, f.fn), :]) for f in fronts], stats


def check(A, R, tol_r=1e-11, tol_g=1e-12):
    Rref = oracle.front_qr(A)
    k = min(A.shape)
    assert gram_resid(R, A) <= tol_g
    if np.linalg.matrix_rank(A) == min(A.shape) and A.shape[0] >= A.shape[1]:
        assert r_rel_diff(rownorm_sign(R), rownorm_sign(Rref[:k])) <= tol_r
    # inside every 32-column panel the diagonal block is left upper triangular;
    # further left the rows hold Householder vectors of earlier panels, which
    # the callers never read (spqrgpu_kernel.cpp:419-436 copies j >= i only)
    for c0 in range(0, k, 32):
        blk = R[c0:c0 + 32, c0:c0 + 32]
        assert np.all(np.tril(blk, -1) == 0.0)


# the four experiments of gpuqrengine_demo.cpp:305-308
@pytest.mark.parametrize("count,m,n", [(2, 8, 8), (2, 12, 8), (2, 64, 32), (1, 100, 200)])
def test_demo_geometries(count, m, n):
    mats = [demo_rand_front(m, n, seed_state=1 + i) for i in range(count)]
    outs, stats = run_dense(mats)
    for A, R in zip(mats, outs):
        check(A, R)
    assert stats.numLaunches >= 1 and stats.flopsActual > 0


@pytest.mark.parametrize("m,n", [(1, 1), (1, 7), (7, 1), (31, 33), (33, 31), (96, 32), (97, 33),
                                 (256, 32), (257, 64), (300, 100), (512, 512), (1000, 130),
                                 (2049, 96), (2304, 65), (96, 2048), (700, 1500)])
def test_shapes(m, n):
    rng = np.random.default_rng(m * 10007 + n)
    A = rng.uniform(-1, 1, size=(m, n))
    (R,), _ = run_dense([A])
    check(A, R)


def test_batch_mixed():
    rng = np.random.default_rng(7)
    shapes = [(96, 32), (192, 64), (384, 128), (768, 256), (40, 90), (5, 5), (1536, 64), (333, 77)]
    mats = [rng.uniform(-1, 1, size=s) for s in shapes]
    outs, _ = run_dense(mats)
    for A, R in zip(mats, outs):
        check(A, R)


def test_staircase_fronts():
    rng = np.random.default_rng(11)
    mats = [stair_front(400, 150, rng), stair_front(1200, 300, rng, aggression=8), stair_front(64, 64, rng, 2)]
    # engine finds the staircase itself (Stair = NULL) ...
    outs, _ = run_dense(mats)
    for A, R in zip(mats, outs):
        check(A, R)
    # ... or takes the caller's
    stairs = [oracle.find_staircase(A) for A in mats]
    outs2, _ = run_dense(mats, stairs=stairs)
    for R1, R2 in zip(outs, outs2):
        assert np.array_equal(R1, R2)


def test_col_major_layout():
    rng = np.random.default_rng(3)
    mats = [rng.uniform(-1, 1, size=(130, 70)), rng.uniform(-1, 1, size=(50, 120))]
    outs_rm, _ = run_dense(mats)
    outs_cm, _ = run_dense(mats, col_major=True)
    for a, b in zip(outs_rm, outs_cm):
        assert np.array_equal(a, b)


def test_zero_and_rank_deficient_columns():
    rng = np.random.default_rng(5)
    A = rng.uniform(-1, 1, size=(200, 64))
    A[:, 10] = 0.0
    A[:, 20] = A[:, 19]
    (R,), _ = run_dense([A])
    assert gram_resid(R, A) <= 1e-12
    assert abs(R[20, 20]) <= 1e-12 * np.abs(R).max()


def test_cpp_demo_program():
    """The reference's dense demo restated in C++ against the shim header (built by build())."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "demo_dense")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all dense demo experiments pass" in out.stdout
