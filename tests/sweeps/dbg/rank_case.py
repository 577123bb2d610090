"""Prints what the device's rank detection and the CPU rule's restatement give on one constructed case."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, scipy.sparse as sp
from suitesparse_amd import spqr
from oracle import oracle
from test_sparse_gpu import _rank_case
name = sys.argv[1] if len(sys.argv) > 1 else "row-left-by-a-dead-column"
A, Q, _ = _rank_case(name)
A = sp.csc_matrix(A)
sym = spqr.analyze(A, Q, relax=20, gpuMemorySize=2**36, rank_detection=True)
tol = spqr.default_tol(A)
res = spqr.factorize(sym, A, tol=tol)
Ro, Rdead_o, rank_o, _, _ = oracle.multifrontal_tol(sym, spqr.s_values(sym, A), tol)
print("tol", tol, "passes", res.numPasses, "rank", res.rank, "cpu rule", rank_o)
print("Rdead device", np.nonzero(res.Rdead)[0], "cpu rule", np.nonzero(Rdead_o)[0])
np.set_printoptions(precision=3, linewidth=200)
if A.shape[1] <= 16:
    print(res.R.toarray())
    print(Ro.toarray())
