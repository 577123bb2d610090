"""One seed of the grid mode of tests/sweeps/fuzz_paths.py: which staged run fails, and how."""
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
print(what, A.shape, "relax", relax)
for mem in (2**34, 30_000_000, 0):
    try:
        s = spqr.analyze(A, Q, relax=relax, gpuMemorySize=mem)
        print("memory", mem, "stages", s.numStages, "nf", s.nf, "max front", int(s.Fm[:s.nf].max()), "x", int(np.diff(s.Rp)[:s.nf].max()), flush=True)
        r = spqr.factorize(s, A)
        print("   ok, stages run", r.numStagesRun, flush=True)
    except RuntimeError as e:
        print("   FAILED:", e, flush=True)
