"""Randomised sweep of the other ways through the numeric driver, on the GPU: for random sparse
full-column-rank problems (random ordering, amalgamation), the R of the one-stage factorization must
come back (a) bit for bit from staged runs (declared memory 0 / small), (b) to rounding from the tree cut
into pieces for 2-4 "ranks" that one process factorizes call by call (device or host blocks), (c) from
a factorization that keeps H, whose Q'B and least-squares solutions are checked against LAPACK,
(d) from the appended-right-hand-side recipe, and (e) bit for bit through the binding an SPQR build
uses (symbolic object from plain arrays, S's values in, R on the caller's stack).  One case in five is
wide (pivots run out of rows; (c) and (d) are skipped there).  Prints failures and a summary; exit code 1 on failure.

    python tests/sweeps/fuzz_paths.py [cases] [first_seed]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from suitesparse_amd import multigpu, spqr     # noqa: E402

COUNT = {"revalued": 0, "staged": 0, "pieces": 0, "binding": 0, "kept": 0, "appended": 0}
NMAX = int(os.environ.get("FUZZ_NMAX", "220"))


GRID = os.environ.get("FUZZ_GRID") is not None      # grid least-squares problems of random shape instead


def grid_case(rng):
    from suitesparse_amd import problems
    if rng.random() < 0.5:
        shape = (int(rng.integers(3, 140)), int(rng.integers(3, 140)))
    else:
        shape = tuple(int(x) for x in rng.integers(2, 22, 3))
    A = problems.stacked_ls(shape)
    Q = problems.nested_dissection(shape, leaf=int(rng.integers(2, 9)))
    return sp.csc_matrix(A), Q, "grid %s" % "x".join(map(str, shape))


def one_case(seed):
    import torch
    rng = np.random.default_rng(seed)
    if GRID:
        A, Q, what = grid_case(rng)
        relax = int(rng.choice([0, 4, 20]))
        return run_paths(seed, rng, A, Q, relax, False, what)
    n = int(rng.integers(2, NMAX))
    wide = bool(rng.random() < 0.2)             # more columns than rows: pivots run out of rows
    m = int(rng.integers(max(1, n // 3), n)) if wide else int(rng.integers(n, 3 * n + 8))
    dens = float(rng.choice([0.01, 0.03, 0.08, 0.2]))
    A = sp.random(m, n, density=dens, random_state=int(seed), format="lil")
    kd = min(m, n)
    A[np.arange(kd), np.arange(kd)] = 1.0 + rng.random(kd)
    A = sp.csc_matrix(A)
    Q = rng.permutation(n) if rng.random() < 0.6 else None
    relax = int(rng.choice([0, 4, 20]))
    return run_paths(seed, rng, A, Q, relax, wide, "dens %.2f" % dens)


def run_paths(seed, rng, A, Q, relax, wide, what):
    import torch
    m, n = A.shape
    tag = "seed %d: %dx%d %s relax %d" % (seed, m, n, what, relax)
    fails = []
    sym = spqr.analyze(A, Q, relax=relax, gpuMemorySize=2**34)
    base = spqr.factorize(sym, A)
    scale = np.abs(base.R).max()
    AQs = A[:, sym.Qfill]
    G = (AQs.T @ AQs)
    E = abs(base.R.T @ base.R - G).max()
    if not E <= 1e-11 * abs(G).max():
        fails.append("%s: R'R differs from (AQ)'(AQ) by %.2e" % (tag, E))
    # (f) the same pattern with other values and back, through whatever the driver and the engine kept from
    # the first call (maps, S permutation, schedule): 4 A gives 4 R (to rounding: the reciprocal and root
    # seeds of the panel kernel are not exactly scale-invariant), and A again gives the first R bit for bit
    r4 = spqr.factorize(sym, A * 4.0)
    r5 = spqr.factorize(sym, A)
    COUNT["revalued"] += 1
    if not abs(r4.R - base.R * 4.0).max() <= 4e-12 * scale:
        fails.append("%s: second factorization on the same symbolic object (values x 4) is not 4 R: %.2e" % (
            tag, abs(r4.R - base.R * 4.0).max() / scale))
    if (r5.R != base.R).nnz != 0:
        fails.append("%s: third factorization (the first values again) differs from the first" % tag)
    # (a) staged runs
    for mem in ((0, 4_000_000) if not GRID else (0, 30_000_000)):
        try:
            s2 = spqr.analyze(A, Q, relax=relax, gpuMemorySize=mem)
        except RuntimeError:
            continue
        if s2.numStages == 0:
            continue                            # a front does not fit the declared memory: infeasible, as in the reference
        r2 = spqr.factorize(s2, A)
        COUNT["staged"] += 1
        if (r2.R != base.R).nnz != 0:
            fails.append("%s: staged run (memory %d, %d stages) is not bit-identical: %.2e" % (
                tag, mem, r2.numStagesRun, np.abs(r2.R - base.R).max() / scale))
    # (b) the tree in pieces
    parts = int(rng.integers(2, 5))
    dev_blocks = bool(rng.random() < 0.5)
    mem = int(rng.choice([0, 5_000_000 if not GRID else 40_000_000, 2**34]))
    try:
        s3 = spqr.analyze(A, Q, relax=relax, gpuMemorySize=mem)
    except RuntimeError:
        s3 = sym
    if s3.numStages == 0:
        s3, mem = sym, 2**34
    owner, level = multigpu.map_tree(s3, parts)
    held, Rall = {}, {}
    for L in range(int(level.max()) + 1 if s3.nf else 0):
        for r in range(parts):
            mine = (owner == r) & (level == L)
            if not mine.any():
                continue
            limbo = {c: held[c] for f in np.nonzero(mine)[0] for c in multigpu._children(s3, int(f)) if not mine[c]}
            if dev_blocks:
                Rb, Cb, _ = multigpu.gpu_factor_subset_dev(s3, A, mine, limbo)
                Cb = {f: t.clone() for f, t in Cb.items()}
            else:
                Rb, Cb, _ = multigpu.gpu_factor_subset(s3, A, mine, limbo)
            held.update(Cb)
            Rall.update(Rb)
    torch.cuda.synchronize()
    Rp = multigpu.blocks_to_csc(s3, Rall)
    COUNT["pieces"] += 1
    if Rp.shape != base.R.shape or not np.abs(Rp - base.R).max() <= 1e-12 * scale:
        fails.append("%s: %d pieces (%s blocks, memory %d) differ from the whole tree" % (
            tag, parts, "device" if dev_blocks else "host", mem))
    # (e) the binding an SPQR build uses: the symbolic object from plain arrays, S's values in, R on the caller's stack
    sym_b = spqr.symbolic_from_arrays(s3)
    out = spqr.spqrgpu_kernel(sym_b, spqr.s_values(s3, A))
    COUNT["binding"] += 1
    if out["code"] != 0:
        fails.append("%s: spqrgpu_amd_kernel returned %d" % (tag, out["code"]))
    else:
        Rb = {f: out["stack"][out["Rblock"][f]:out["Rblock"][f] + multigpu._r_size(s3, f)] for f in range(s3.nf)}
        Rk = multigpu.blocks_to_csc(s3, Rb)
        if Rk.shape != base.R.shape or (Rk != base.R).nnz != 0:
            fails.append("%s: R through the SPQR binding (memory %d) is not bit-identical" % (tag, mem))
    if wide or n > 2500:
        return fails
    # (c) H kept on the device
    k = int(rng.integers(1, 6))
    B = rng.standard_normal((m, k))
    AQ = A[:, sym.Qfill].toarray()
    if np.linalg.cond(AQ) < 1e8:
        kept = spqr.KeptFactorization(sym, A, nrhs=k)
        Qd, Rd = np.linalg.qr(AQ, mode="reduced")
        # row signs: the device's R has a non-negative diagonal except where a column had nothing below
        # its pivot (then the reflector is the identity and the pivot keeps its sign)
        sgn = np.where(np.diag(Rd) * kept.R.diagonal()[:n] < 0, -1.0, 1.0)
        Cq = kept.qtmult(B)
        ref = (Qd * sgn).T @ B
        COUNT["kept"] += 1
        if not np.abs(Cq - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1.0):
            fails.append("%s: Q'B from the kept factorization differs by %.2e" % (tag, np.abs(Cq - ref).max()))
        x = kept.solve(B)
        xl = np.linalg.lstsq(A.toarray(), B, rcond=None)[0]
        if not np.abs(x - xl).max() <= 1e-7 * max(np.abs(xl).max(), 1.0):
            fails.append("%s: solve from the kept factorization differs by %.2e" % (tag, np.abs(x - xl).max()))
        kept.close()
        # (d) appended right-hand sides
        x2, _, _ = spqr.qr_solve(A, B, Q, relax=relax)
        COUNT["appended"] += 1
        if not np.abs(x2 - xl).max() <= 1e-7 * max(np.abs(xl).max(), 1.0):
            fails.append("%s: solve with appended right-hand sides differs by %.2e" % (tag, np.abs(x2 - xl).max()))
    return fails


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0, bad = time.time(), []
    for s in range(first, first + cases):
        try:
            f = one_case(s)
        except RuntimeError as e:
            f = ["seed %d: %s" % (s, e)]
        for line in f:
            print("FAIL", line, flush=True)
        bad += f
        if (s - first) % 25 == 24:
            print("... %d cases, %d failures, %.0f s" % (s - first + 1, len(bad), time.time() - t0), flush=True)
    print("%d cases from seed %d: %d failures in %.0f s; checked: %s" % (cases, first, len(bad), time.time() - t0, COUNT))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
