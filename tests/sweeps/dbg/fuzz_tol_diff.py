"""Where the tolerance path's R differs from the CPU rule's restatement, for one seed of tests/sweeps/fuzz_parity.py."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np, scipy.sparse as sp
from oracle import oracle
from suitesparse_amd import spqr
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
n = int(rng.integers(1, int(os.environ.get("FUZZ_NMAX", "260")))); m = int(rng.integers(max(1, n // 2), 3 * n + 8))
dens = float(rng.choice([0.01, 0.03, 0.08, 0.2, 0.6]))
A = sp.random(m, n, density=dens, random_state=int(seed), format="lil")
k = min(m, n)
if rng.random() < 0.7:
    A[np.arange(k), np.arange(k)] = 1.0 + rng.random(k)
A = sp.csc_matrix(A)
Q = rng.permutation(n) if rng.random() < 0.5 else None
relax = int(rng.choice([0, 4, 20])); mem = int(rng.choice([0, 3_000_000, 6_000_000, 2**34]))
B = sp.lil_matrix(A)
for _ in range(int(rng.integers(1, 5))):
    a, b = rng.integers(0, n, 2)
    if a != b:
        B[:, a] = B[:, b] * float(rng.choice([1.0, -2.0, 0.0]))
B = sp.csc_matrix(B)
sym = spqr.analyze(B, Q, relax=relax, gpuMemorySize=2**34, rank_detection=True)
tol = spqr.default_tol(B)
rr = spqr.factorize(sym, B, tol=tol)
Ro, Rdead_o, rank_o, franks, normE = oracle.multifrontal_tol(sym, spqr.s_values(sym, B), tol)
print("passes", rr.numPasses, "rank", rr.rank, rank_o, "dead equal", np.array_equal(rr.Rdead, Rdead_o), "ndead", int(Rdead_o.sum()))
Rd, Rc = rr.R.toarray(), Ro.toarray()
D = np.abs(np.abs(Rd) - np.abs(Rc))
print("max diff", D.max(), "scale", np.abs(Rc).max())
rows, cols = np.nonzero(D > 1e-9 * np.abs(Rc).max())
print("differing rows", sorted(set(rows.tolist()))[:40])
print("differing cols", sorted(set(cols.tolist()))[:40])
# which front holds the first differing row
row0 = 0
for f in range(sym.nf):
    fp = sym.Super[f + 1] - sym.Super[f]
    live = int((rr.Rdead[sym.Super[f]:sym.Super[f + 1]] == 0).sum())
    rk = min(live, int(sym.Fm[f]))
    if len(rows) and row0 <= rows.min() < row0 + rk:
        print("first differing row is in front", f, "rows", row0, row0 + rk, "pivots", sym.Super[f], sym.Super[f + 1], "Fm", sym.Fm[f],
              "fn", sym.Rp[f + 1] - sym.Rp[f], "dead in it", np.nonzero(rr.Rdead[sym.Super[f]:sym.Super[f + 1]])[0], "children", sym.Child[sym.Childp[f]:sym.Childp[f + 1]])
    row0 += rk
G = B[:, sym.Qfill].toarray()
live = np.nonzero(Rdead_o == 0)[0]
for name, R in (("device", Rd), ("cpu rule", Rc)):
    E = R[:, live].T @ R[:, live] - G[:, live].T @ G[:, live]
    print(name, "R'R - (AQ)'(AQ) on the live columns:", np.abs(E).max())
# Gram identity of the differing columns against the live columns before them: must hold for dead ones too
for c in sorted(set(cols.tolist()))[:6]:
    lb = live[live < c]
    for name, R in (("device", Rd), ("cpu rule", Rc)):
        e = R[:, lb].T @ R[:, c] - G[:, lb].T @ G[:, c]
        print("col", c, "dead" if Rdead_o[c] else "live", name, "Gram error against live columns before it: %.3e" % (np.abs(e).max() if len(lb) else 0.0),
              "worst at live col", int(lb[np.argmax(np.abs(e))]) if len(lb) else -1)
    print("   entries device", Rd[rows.min() - 1:, c][:6], "cpu", Rc[rows.min() - 1:, c][:6])
