"""Which contribution blocks below the first non-scale-invariant front are themselves not exactly scaled?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np
from suitesparse_amd import multigpu, problems, spqr
shape, leaf, relax, front = (15, 15), 8, 0, 78
A = problems.stacked_ls(shape)
sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=leaf), relax=relax, gpuMemorySize=2**34)
nf = sym.nf
def subtree(f):
    out, st = [], [f]
    while st:
        g = st.pop(); out.append(g); st.extend(int(c) for c in sym.Child[sym.Childp[g]:sym.Childp[g + 1]])
    return out
def walk(f, depth):
    for c in sym.Child[sym.Childp[f]:sym.Childp[f + 1]]:
        c = int(c)
        mine = np.zeros(nf, bool); mine[subtree(c)] = True
        R1, C1, _ = multigpu.gpu_factor_subset(sym, A, mine, {})
        R4, C4, _ = multigpu.gpu_factor_subset(sym, A * 4.0, mine, {})
        bad = int(np.count_nonzero(C4[c] != 4.0 * C1[c]))
        fm, fn, fp = int(sym.Fm[c]), int(sym.Rp[c + 1] - sym.Rp[c]), int(sym.Super[c + 1] - sym.Super[c])
        print("  " * depth + "front %d (fm %d fn %d fp %d, %d children): C block %s, %d of %d entries not exactly scaled" % (
            c, fm, fn, fp, sym.Childp[c + 1] - sym.Childp[c], (int(sym.Cm[c]), fn - fp), bad, C1[c].size))
        if bad and depth < 6:
            walk(c, depth + 1)
        if bad and depth == 0:
            cm, cn = int(sym.Cm[c]), fn - fp
            D = (C4[c] - 4.0 * C1[c]).reshape(cm, cn)
            r, cc = np.nonzero(D)
            print("      first differing C row", r.min(), "cols", cc[r == r.min()][:8], "values", C1[c].reshape(cm, cn)[r.min(), cc[r == r.min()][:4]], "diff", D[r.min(), cc[r == r.min()][:4]])
walk(front, 0)
