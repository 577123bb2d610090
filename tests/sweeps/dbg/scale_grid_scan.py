"""Smallest grid problems for which R(4A) != 4 R(A) bit for bit through the sparse path."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np
from suitesparse_amd import problems, spqr
found = 0
for shape in [(k, k) for k in range(3, 40)] + [(k, k, k) for k in range(2, 12)]:
    for leaf in (2, 4, 8):
        for relax in (0, 20):
            A = problems.stacked_ls(shape)
            sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=leaf), relax=relax, gpuMemorySize=2**34)
            R1 = spqr.factorize(sym, A).R
            R4 = spqr.factorize(sym, A * 4.0).R
            nd = (R4 != R1 * 4.0).nnz
            if nd:
                D = (R4 - R1 * 4.0).tocsr()
                row = int(np.nonzero(np.diff(D.indptr))[0][0])
                row0 = 0
                for f in range(sym.nf):
                    rk = min(int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f]))
                    if row0 <= row < row0 + rk:
                        print(shape, "leaf", leaf, "relax", relax, "nf", sym.nf, ":", nd, "entries; first in front", f, "fm", int(sym.Fm[f]),
                              "fn", int(sym.Rp[f + 1] - sym.Rp[f]), "fp", int(sym.Super[f + 1] - sym.Super[f]), "local row", row - row0,
                              "children", len(sym.Child[sym.Childp[f]:sym.Childp[f + 1]]), flush=True)
                    row0 += rk
                found += 1
        if found >= 6:
            break
    if found >= 6:
        break
print("done, found", found)
