"""Randomised parity sweep on the GPU: random sparse matrices, orderings, amalgamation and declared
device memory (so one stage, many stages, one front per stage) through the numeric driver, R compared
with LAPACK's QR of A(:,Qfill) after row-sign normalisation; every few cases the same matrix with
dependent columns through the tolerance path, compared with the oracle's restatement of the CPU rule
(rank, Rdead, R).  Prints one line per failure and a summary; exit code 1 if anything failed.

    python tests/sweeps/fuzz_parity.py [cases] [first_seed]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import oracle            # noqa: E402  (checker only)
from suitesparse_amd import spqr     # noqa: E402


COUNT = {"infeasible": 0, "gram": 0, "lapack": 0, "tol": 0, "tol_dead": 0, "tol_borderline": 0, "stages>1": 0, "max_passes": 0}
NMAX = int(os.environ.get("FUZZ_NMAX", "260"))
TOL_EVERY = int(os.environ.get("FUZZ_TOL_EVERY", "3"))


def borderline(B, sym, dead_dev, dead_cpu, tol):
    """The first column on which the two sets of decisions differ, seen through numpy: its distance from
    the span of the columns both sides keep alive before it.  The decisions up to there agree, so that
    distance is the number both sides compare with tol; if rounding alone (eps times the condition of
    those live columns) can move it across tol, either answer is the rule's."""
    j = int(np.nonzero(dead_dev != dead_cpu)[0][0])
    BQ = B[:, sym.Qfill].toarray()
    live = [i for i in range(j) if not dead_cpu[i]]
    if not live:
        return False
    sv = np.linalg.svd(BQ[:, live], compute_uv=False)
    Qm, _ = np.linalg.qr(BQ[:, live], mode="reduced")
    r = BQ[:, j] - Qm @ (Qm.T @ BQ[:, j])
    r = np.linalg.norm(r - Qm @ (Qm.T @ r))
    noise = 50 * 2.2e-16 * sv[0] * (sv[0] / max(sv[-1], 1e-300))
    return abs(r - tol) <= noise


def grid_tol_case(seed):
    """A 2D grid least-squares problem of random size with duplicated and nearly duplicated columns
    (BASELINE's rank-deficient configuration at random sizes), nested dissection, through the tolerance
    path against the CPU rule's restatement."""
    from suitesparse_amd import problems
    rng = np.random.default_rng(seed)
    k = int(rng.integers(6, 90))
    n = k * k
    ndup, nnear = int(rng.integers(0, max(1, n // 20))), int(rng.integers(0, max(1, n // 20)))
    B, _, _ = problems.rank_deficient_ls(k, ndup, nnear, seed=seed)
    Q = problems.nested_dissection((k, k), leaf=int(rng.integers(2, 9)))
    relax = int(rng.choice([0, 4, 20]))
    mem = int(rng.choice([0, 40_000_000, 2**34]))
    tag = "seed %d: grid %dx%d with %d + %d dependent columns, relax %d mem %d" % (seed, k, k, ndup, nnear, relax, mem)
    symr = spqr.analyze(B, Q, relax=relax, gpuMemorySize=mem, rank_detection=True)
    if symr.numStages == 0:
        symr = spqr.analyze(B, Q, relax=relax, gpuMemorySize=2**34, rank_detection=True)
    tol = spqr.default_tol(B)
    rr = spqr.factorize(symr, B, tol=tol)
    Ro, Rdead_o, rank_o, _, normE = oracle.multifrontal_tol(symr, spqr.s_values(symr, B), tol)
    COUNT["tol"] += 1
    COUNT["tol_dead"] += int(rr.numPasses > 1)
    COUNT["max_passes"] = max(COUNT["max_passes"], rr.numPasses)
    COUNT["stages>1"] += int(rr.numStagesRun > 1)
    if rr.rank != rank_o or not np.array_equal(rr.Rdead, Rdead_o):
        return ["%s: rank %d vs %d, dead only on the device %s, only by the CPU rule %s" % (
            tag, rr.rank, rank_o, np.nonzero(rr.Rdead > Rdead_o)[0][:8], np.nonzero(rr.Rdead < Rdead_o)[0][:8])]
    Rs, Rs_o = rr.R.tocsc(), Ro.tocsc()
    Rs.sort_indices()
    Rs_o.sort_indices()
    if not (np.array_equal(Rs.indptr, Rs_o.indptr) and np.array_equal(Rs.indices, Rs_o.indices)):
        return ["%s: pattern of the squeezed R differs" % tag]
    d = np.abs(np.abs(Rs.data) - np.abs(Rs_o.data)).max() / max(np.abs(Rs_o.data).max(), 1e-300) if Rs.nnz else 0.0
    if not d <= 1e-10:
        return ["%s: R differs by %.2e" % (tag, d)]
    if abs(rr.norm_E_fro - normE) > 0.05 * tol:
        return ["%s: norm_E_fro %.3e vs %.3e" % (tag, rr.norm_E_fro, normE)]
    return []


def one_case(seed):
    if os.environ.get("FUZZ_GRID") is not None:
        return grid_tol_case(seed)
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, NMAX))
    m = int(rng.integers(max(1, n // 2), 3 * n + 8))
    dens = float(rng.choice([0.01, 0.03, 0.08, 0.2, 0.6]))
    A = sp.random(m, n, density=dens, random_state=int(seed), format="lil")
    k = min(m, n)
    if rng.random() < 0.7:
        A[np.arange(k), np.arange(k)] = 1.0 + rng.random(k)        # mostly full rank
    A = sp.csc_matrix(A)
    Q = rng.permutation(n) if rng.random() < 0.5 else None
    relax = int(rng.choice([0, 4, 20]))
    mem = int(rng.choice([0, 3_000_000, 6_000_000, 2**34]))
    tag = "seed %d: %dx%d dens %.2f relax %d mem %d" % (seed, m, n, dens, relax, mem)
    fails = []
    try:
        sym = spqr.analyze(A, Q, relax=relax, gpuMemorySize=mem)
    except RuntimeError as e:
        if mem == 3_000_000 or mem == 6_000_000:
            COUNT["infeasible"] += 1
            return []                                   # a front does not fit the declared memory: infeasible
        return ["%s: analyze: %s" % (tag, e)]
    res = spqr.factorize(sym, A)
    COUNT["gram"] += 1
    COUNT["stages>1"] += int(res.numStagesRun > 1)
    R = res.R.toarray()
    AQ = A[:, sym.Qfill].toarray()
    # structural rank rows; compare R'R (unique) and, when A has full column rank, R itself
    G = AQ.T @ AQ
    E = np.abs(R.T @ R - G).max()
    if not E <= 1e-11 * max(np.abs(G).max(), 1e-300):
        fails.append("%s: R'R differs from (AQ)'(AQ) by %.2e" % (tag, E))
    if m >= n and res.rank == n and np.linalg.matrix_rank(AQ) == n and np.linalg.cond(AQ) < 1e8:
        COUNT["lapack"] += 1
        Rl = np.linalg.qr(AQ, mode="r")
        sg = np.where(np.sign(np.diag(Rl)) * np.sign(np.diag(R[:n])) < 0, -1.0, 1.0)
        d = np.abs(R[:n] - sg[:, None] * Rl).max() / np.abs(Rl).max()
        if not d <= 1e-10:
            fails.append("%s: R differs from LAPACK's by %.2e" % (tag, d))
    # the tolerance path on a variant with dependent columns
    if seed % TOL_EVERY == 0 and n >= 4:
        B = sp.lil_matrix(A)
        for _ in range(int(rng.integers(1, 5))):
            a, b = rng.integers(0, n, 2)
            if a != b:
                B[:, a] = B[:, b] * float(rng.choice([1.0, -2.0, 0.0]))
        B = sp.csc_matrix(B)
        try:
            symr = spqr.analyze(B, Q, relax=relax, gpuMemorySize=mem, rank_detection=True)
        except RuntimeError:
            symr = spqr.analyze(B, Q, relax=relax, gpuMemorySize=2**34, rank_detection=True)
        tol = spqr.default_tol(B)
        if tol > 0:
            try:
                rr = spqr.factorize(symr, B, tol=tol)
                Ro, Rdead_o, rank_o, _, normE = oracle.multifrontal_tol(symr, spqr.s_values(symr, B), tol)
                COUNT["tol"] += 1
                COUNT["tol_dead"] += int(rr.numPasses > 1)
                COUNT["max_passes"] = max(COUNT["max_passes"], rr.numPasses)
                if (rr.rank != rank_o or not np.array_equal(rr.Rdead, Rdead_o)) and borderline(B, symr, rr.Rdead, Rdead_o, tol):
                    COUNT["tol_borderline"] += 1
                elif rr.rank != rank_o or not np.array_equal(rr.Rdead, Rdead_o):
                    fails.append("%s: tol path rank %d vs %d, dead only on the device %s, only by the CPU rule %s" % (
                        tag, rr.rank, rank_o, np.nonzero(rr.Rdead > Rdead_o)[0][:8], np.nonzero(rr.Rdead < Rdead_o)[0][:8]))
                else:
                    Rr, Roo = rr.R.toarray(), Ro.toarray()
                    Rs, Rs_o = rr.R.tocsc(), Ro.tocsc()
                    Rs.sort_indices()
                    Rs_o.sort_indices()
                    if not (np.array_equal(Rs.indptr, Rs_o.indptr) and np.array_equal(Rs.indices, Rs_o.indices)):
                        fails.append("%s: tol path: pattern of the squeezed R differs" % tag)
                    # R(:,i)'R(:,j) = (BQ)(:,i)'(BQ)(:,j) for every live i and every j >= i, dead ones included
                    # (what a dead column dropped lies below row i): holds whatever the rounding did
                    BQ = B[:, symr.Qfill].toarray()
                    lv = np.nonzero(Rdead_o == 0)[0]
                    E = np.triu(Rr.T @ Rr - BQ.T @ BQ)[lv]
                    if E.size and not np.abs(E).max() <= 1e-11 * max(np.abs(BQ).max() ** 2 * BQ.shape[0], 1e-300):
                        fails.append("%s: tol path: R'R differs from (BQ)'(BQ) on live rows by %.2e" % (tag, np.abs(E).max()))
                    # entry by entry only when no live pivot is so small that its row of R is decided by rounding
                    dg = np.array([abs(Roo[i, j]) for i, j in enumerate(lv[:Roo.shape[0]])])
                    d = np.abs(np.abs(Rr) - np.abs(Roo)).max() / max(np.abs(Roo).max(), 1e-300)
                    if dg.size and dg.min() > 1e-6 * np.abs(Roo).max() and not d <= 1e-9:
                        fails.append("%s: tol path R differs by %.2e" % (tag, d))
                    # (mostly rounding noise of dependent columns on both sides: allow for that much)
                    noise = 50 * 2.2e-16 * np.sqrt(float(abs(B).power(2).sum()) * max(int(Rdead_o.sum()), 1))
                    if dg.size and dg.min() > 0:
                        noise *= np.abs(Roo).max() / dg.min()          # ... times the condition of the live columns
                    if dg.size and dg.min() > 1e-6 * np.abs(Roo).max() and abs(rr.norm_E_fro - normE) > 0.05 * tol + noise:
                        fails.append("%s: norm_E_fro %.3e vs %.3e" % (tag, rr.norm_E_fro, normE))
            except RuntimeError as e:
                fails.append("%s: tol path: %s" % (tag, e))
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
    print("%d cases from seed %d: %d failures in %.0f s; checked: %s" % (cases, first, len(bad), time.time() - t0, COUNT))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
