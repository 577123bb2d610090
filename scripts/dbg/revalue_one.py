"""Re-factorization on one symbolic object: same values twice, then values x 4 twice (grid mode seed of fuzz_paths)."""
import os, sys
os.environ["FUZZ_GRID"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import fuzz_paths
from suitesparse_amd import spqr
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
A, Q, what = fuzz_paths.grid_case(rng)
relax = int(rng.choice([0, 4, 20]))
sym = spqr.analyze(A, Q, relax=relax, gpuMemorySize=2**34)
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
