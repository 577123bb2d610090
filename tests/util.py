"""Shared helpers of the parity tests."""
import numpy as np


def rownorm_sign(R):
    """Flip rows so that the diagonal is non-negative (row-sign normalisation of R)."""
    R = np.array(R, copy=True)
    k = min(R.shape)
    for i in range(k):
        if R[i, i] < 0:
            R[i, :] = -R[i, :]
    return R


def r_rel_diff(R, Rref):
    """max |abs(R) - abs(Rref)| / max |Rref| over the upper trapezoid."""
    k = min(R.shape[0], Rref.shape[0])
    A = np.triu(np.abs(R[:k]))
    B = np.triu(np.abs(Rref[:k]))
    return float(np.max(np.abs(A - B)) / max(np.max(np.abs(B)), 1e-300))


def gram_resid(R, A):
    """|| triu(R)' triu(R) - A'A ||_1 / || A'A ||_1, the reference's dense-demo and
    check_r_factor metric (GPUQREngine/Demo/gpuqrengine_demo.cpp:94-100,
    SPQR/Tcov/qrtest.cpp:1218-1257)."""
    R = np.triu(R)
    G = A.T @ A
    return float(np.linalg.norm(R.T @ R - G, 1) / max(np.linalg.norm(G, 1), 1e-300))


def demo_rand_front(m, n, seed_state=None):
    """Entries rand()/1e8 as in gpuqrengine_demo.cpp:21-28; glibc rand() is replaced
    by a seeded numpy generator of the same range [0, 2^31) (the C library's
    generator is not reproducible across platforms)."""
    rng = np.random.default_rng(1 if seed_state is None else seed_state)
    return rng.integers(0, 2**31, size=(m, n)).astype(np.float64) / 1e8


def stair_front(m, n, rng, aggression=4):
    """Staircase-shaped front like randfill_stair (gpuqrengine_demo.cpp:32-44)."""
    F = rng.uniform(-1, 1, size=(m, n))
    last = 0
    for j in range(n):
        last = min(m - 1, last + int(rng.integers(0, aggression + 1)))
        F[last + 1:, j] = 0.0
    return F
