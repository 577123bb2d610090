"""Re-factorization on one symbolic object: same values twice, then values x 4 twice (grid mode seed of fuzz_paths)."""
import os, sys
os.environ["FUZZ_GRID"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np
import fuzz_paths
from suitesparse_amd import spqr
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
A, Q, what = fuzz_paths.grid_case(rng)
relax = int(rng.choice([0, 4, 20]))
MEM = int(sys.argv[2]) if len(sys.argv) > 2 else 2**34
sym = spqr.analyze(A, Q, relax=relax, gpuMemorySize=MEM)
print(what, A.shape, "relax", relax, "nf", sym.nf)
runs = [spqr.factorize(sym, A).R, spqr.factorize(sym, A).R, spqr.factorize(sym, A * 4.0).R, spqr.factorize(sym, A * 4.0).R, spqr.factorize(sym, A).R]
sc = abs(runs[0]).max()
print("same values twice: differing entries", (runs[0] != runs[1]).nnz, "max", abs(runs[0] - runs[1]).max() / sc)
print("x4 twice:          differing entries", (runs[2] != runs[3]).nnz)
print("x4 vs 4 R:         differing entries", (runs[2] != runs[0] * 4.0).nnz, "max", abs(runs[2] - runs[0] * 4.0).max() / sc)
print("back to the first values: differing entries", (runs[4] != runs[0]).nnz)
D = (runs[2] - runs[0] * 4.0).tocoo()
if D.nnz:
    big = np.argsort(-abs(D.data))[:5]
    print("largest:", [(int(D.row[i]), int(D.col[i]), float(D.data[i])) for i in big])
D = (runs[2] - runs[0] * 4.0).tocsr()
rows = np.nonzero(np.diff(D.indptr))[0]
print("first differing rows", rows[:10])
row0 = 0
for f in range(sym.nf):
    rk = min(int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f]))
    if row0 <= rows[0] < row0 + rk:
        fn = int(sym.Rp[f + 1] - sym.Rp[f])
        print("in front", f, "rows of R", row0, row0 + rk, "fm", int(sym.Fm[f]), "fn", fn, "fp", int(sym.Super[f + 1] - sym.Super[f]),
              "children", sym.Child[sym.Childp[f]:sym.Childp[f + 1]], "local row", rows[0] - row0)
        r = D[rows[0]].tocoo()
        print("   columns that differ in that row:", r.col[:12], "values", r.data[:6], "of", runs[0][rows[0]].data[:6] if hasattr(runs[0][rows[0]], "data") else "")
    row0 += rk
