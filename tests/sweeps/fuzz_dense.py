"""Randomised sweep of the dense front-batch entry point on the GPU: batches of random fronts (1..1600
rows and columns, tall and wide, some with a staircase of leading zeros, row- or column-major) against
LAPACK (R'R = A'A, R up to row signs for full-rank tall fronts), the same batch twice (bit-identical),
and through the persistent work-queue kernel (equal to rounding).  Exit code 1 on failure.

    python tests/sweeps/fuzz_dense.py [cases] [first_seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from suitesparse_amd import _lib                                 # noqa: E402
from tests.test_dense_gpu import run_dense                       # noqa: E402
from tests.util import gram_resid, r_rel_diff, rownorm_sign, stair_front    # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    nfronts = int(rng.integers(1, 9))
    mats = []
    for _ in range(nfronts):
        m = int(rng.choice([rng.integers(1, 40), rng.integers(1, 400), rng.integers(1, 1600)]))
        n = int(rng.choice([rng.integers(1, 40), rng.integers(1, 400), rng.integers(1, 1600)]))
        if rng.random() < 0.3 and m >= 4 and n >= 2:
            mats.append(stair_front(m, n, rng, aggression=int(rng.integers(2, 9))))
        else:
            mats.append(rng.uniform(-1, 1, size=(m, n)))
    col_major = bool(rng.random() < 0.3)
    tag = "seed %d: %s%s" % (seed, [a.shape for a in mats], " col-major" if col_major else "")
    fails = []
    L = _lib.lib()
    outs, _ = run_dense(mats, col_major=col_major)
    again, _ = run_dense(mats, col_major=col_major)
    for A, R, R2 in zip(mats, outs, again):
        k = min(A.shape)
        g = gram_resid(R, A)
        if not g <= 1e-12:
            fails.append("%s: front %s: R'R differs from A'A by %.2e" % (tag, A.shape, g))
        if A.shape[0] >= A.shape[1] and np.linalg.cond(A) < 1e8:
            Rl = np.linalg.qr(A, mode="r")
            d = r_rel_diff(rownorm_sign(np.triu(R[:k])), rownorm_sign(Rl))
            if not d <= 1e-10:
                fails.append("%s: front %s: R differs from LAPACK's by %.2e" % (tag, A.shape, d))
        if not np.array_equal(R, R2):
            fails.append("%s: front %s: two runs differ" % (tag, A.shape))
    L.gpuqr_set_mode(1)
    try:
        queue, _ = run_dense(mats, col_major=col_major)
    finally:
        L.gpuqr_set_mode(0)
    for A, R, Rq in zip(mats, outs, queue):
        k = min(A.shape)
        if not np.abs(np.triu(R[:k]) - np.triu(Rq[:k])).max() <= 1e-11 * max(1.0, np.abs(R).max()):
            fails.append("%s: front %s: work-queue kernel differs by %.2e" % (tag, A.shape, np.abs(np.triu(R[:k]) - np.triu(Rq[:k])).max()))
    return fails


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0, bad = time.time(), []
    for s in range(first, first + cases):
        f = one_case(s)
        for line in f:
            print("FAIL", line, flush=True)
        bad += f
        if (s - first) % 25 == 24:
            print("... %d cases, %d failures, %.0f s" % (s - first + 1, len(bad), time.time() - t0), flush=True)
    print("%d cases from seed %d: %d failures in %.0f s" % (cases, first, len(bad), time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
