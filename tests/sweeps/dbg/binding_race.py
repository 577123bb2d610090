"""A staged run of a tiny wide problem through the SPQR binding, six times in a fresh process: every call
must leave the same R on the caller's stack.  (The first call of a process is the slow one: a helper
thread that copies a stage's R out of a device block was still reading it when the stage after an
R-less stage packed into the same block.)  Exit code 1 if the calls disagree.

    [GPUQR_PAGEABLE_R=1] python tests/sweeps/dbg/binding_race.py seed declared_memory
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np, scipy.sparse as sp
from suitesparse_amd import multigpu, spqr
seed, MEM = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
n = int(rng.integers(2, 220)); wide = bool(rng.random() < 0.2)
m = int(rng.integers(max(1, n // 3), n)) if wide else int(rng.integers(n, 3 * n + 8))
dens = float(rng.choice([0.01, 0.03, 0.08, 0.2]))
A = sp.random(m, n, density=dens, random_state=int(seed), format="lil")
kd = min(m, n); A[np.arange(kd), np.arange(kd)] = 1.0 + rng.random(kd); A = sp.csc_matrix(A)
Q = rng.permutation(n) if rng.random() < 0.6 else None
relax = int(rng.choice([0, 4, 20]))
sym = spqr.analyze(A, Q, relax=relax, gpuMemorySize=MEM)
symb = spqr.symbolic_from_arrays(sym)
Sx = spqr.s_values(sym, A)
stacks = []
for trial in range(6):
    out = spqr.spqrgpu_kernel(symb, Sx)
    stacks.append(out["stack"].copy())
    c1 = out["stack"].copy()
    time.sleep(0.05)
    c2 = out["stack"].copy()
    print("trial", trial, "code", out["code"], "at return", c1[:6], "later", c2[:6], "changed" if not np.array_equal(c1, c2) else "")
sys.exit(0 if all(np.array_equal(s, stacks[-1]) for s in stacks) else 1)
