"""The R of the direct driver and of the SPQR binding side by side for one seed of tests/sweeps/fuzz_paths.py."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np, scipy.sparse as sp
from suitesparse_amd import multigpu, spqr
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
n = int(rng.integers(2, 220)); wide = bool(rng.random() < 0.2)
m = int(rng.integers(max(1, n // 3), n)) if wide else int(rng.integers(n, 3 * n + 8))
dens = float(rng.choice([0.01, 0.03, 0.08, 0.2]))
A = sp.random(m, n, density=dens, random_state=int(seed), format="lil")
kd = min(m, n); A[np.arange(kd), np.arange(kd)] = 1.0 + rng.random(kd); A = sp.csc_matrix(A)
Q = rng.permutation(n) if rng.random() < 0.6 else None
relax = int(rng.choice([0, 4, 20]))
MEM = int(sys.argv[2]) if len(sys.argv) > 2 else 2**34
sym = spqr.analyze(A, Q, relax=relax, gpuMemorySize=MEM)
base = spqr.factorize(sym, A)
out = spqr.spqrgpu_kernel(spqr.symbolic_from_arrays(sym), spqr.s_values(sym, A))
Rb = {f: out["stack"][out["Rblock"][f]:out["Rblock"][f] + multigpu._r_size(sym, f)] for f in range(sym.nf)}
Rk = multigpu.blocks_to_csc(sym, Rb)
np.set_printoptions(precision=4, linewidth=220)
print("nf", sym.nf, "Super", sym.Super[:sym.nf + 1], "Fm", sym.Fm[:sym.nf], "fn", np.diff(sym.Rp)[:sym.nf], "Post", sym.Post[:sym.nf])
print("rank", base.rank, out["sumfrank"], "Rdead", np.nonzero(base.Rdead)[0], np.nonzero(out["Rdead"][:sym.n])[0])
print("shapes", base.R.shape, Rk.shape, "nnz", base.R.nnz, Rk.nnz, "Rblock", out["Rblock"][:sym.nf], "used", out["used"], "rnz", sym.rnz)
if n <= 24:
    print(base.R.toarray()); print(Rk.toarray())
else:
    D = abs(base.R - Rk) if base.R.shape == Rk.shape else None
    if D is not None:
        r, c = D.nonzero(); print("differ at rows", sorted(set(r.tolist()))[:20], "cols", sorted(set(c.tolist()))[:20])
print("stack", out["stack"][:12], "numStages", sym.numStages, "Stagingp", sym.Stagingp[:sym.numStages + 2], "StageMap", sym.StageMap[:sym.nf])
