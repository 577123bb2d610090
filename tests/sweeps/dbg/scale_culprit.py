"""Follow the non-scale-invariance down the tree to the first front whose inputs are exactly scaled and
whose contribution block is not; assemble that front on the host and push it through the dense entry
point in both scalings."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np
from suitesparse_amd import multigpu, problems, spqr
from tests.test_dense_gpu import run_dense
shape, leaf, relax, front = (15, 15), 8, 0, 78
A = problems.stacked_ls(shape)
sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=leaf), relax=relax, gpuMemorySize=2**34)
nf = sym.nf
def subtree(f):
    out, st = [], [f]
    while st:
        g = st.pop(); out.append(g); st.extend(int(c) for c in sym.Child[sym.Childp[g]:sym.Childp[g + 1]])
    return out
def cblock(c, scale):
    mine = np.zeros(nf, bool); mine[subtree(c)] = True
    _, C, _ = multigpu.gpu_factor_subset(sym, A * scale, mine, {})
    return C[c]
def exact(c):
    return np.array_equal(cblock(c, 4.0), 4.0 * cblock(c, 1.0))
f = front
while True:
    kids = [int(c) for c in sym.Child[sym.Childp[f]:sym.Childp[f + 1]]]
    badkids = [c for c in kids if int(sym.Cm[c]) * (int(sym.Rp[c + 1] - sym.Rp[c]) - int(sym.Super[c + 1] - sym.Super[c])) > 0 and not exact(c)]
    if not badkids:
        break
    f = badkids[0]
fm, fn, fp = int(sym.Fm[f]), int(sym.Rp[f + 1] - sym.Rp[f]), int(sym.Super[f + 1] - sym.Super[f])
kids = [int(c) for c in sym.Child[sym.Childp[f]:sym.Childp[f + 1]]]
print("culprit front", f, "fm", fm, "fn", fn, "fp", fp, "children", kids, [(int(sym.Fm[c]), int(sym.Rp[c+1]-sym.Rp[c]), int(sym.Super[c+1]-sym.Super[c])) for c in kids])
# its input, assembled on the host from S and the children's blocks (scale 1), as the assembly kernels do
Stair, Rjmap, Rimap, cpuS = spqr.spqrgpu_buildAssemblyMaps(sym, spqr.s_values(sym, A))
Scount = np.array([sym.Sp[sym.Sleft[sym.Super[g + 1]]] - sym.Sp[sym.Sleft[sym.Super[g]]] for g in range(nf)])
post = sym.Post[:nf]
SOff = np.zeros(nf + 1, np.int64); SOff[post] = np.concatenate([[0], np.cumsum(Scount[post])[:-1]])
F = np.zeros(fm * fn)
e = cpuS[SOff[f]:SOff[f] + Scount[f]]
F[e["findex"]] = e["value"]
F = F.reshape(fm, fn)
for c in kids:
    cm, cn = int(sym.Cm[c]), int(sym.Rp[c + 1] - sym.Rp[c]) - int(sym.Super[c + 1] - sym.Super[c])
    if cm * cn == 0:
        continue
    Cb = cblock(c, 1.0).reshape(cm, cn)
    ri = Rimap[sym.RimapOffsets[c]:sym.RimapOffsets[c] + cm]; rj = Rjmap[sym.RjmapOffsets[c]:sym.RjmapOffsets[c] + cn]
    for i in range(cm):
        F[ri[i], rj[i:]] = Cb[i, i:]
np.set_printoptions(precision=3, linewidth=250, suppress=False)
print("staircase", Stair[sym.Rp[f]:sym.Rp[f + 1]])
print("nonzeros per column", (F != 0).sum(axis=0), "tiny entries (<1e-10):", int(((np.abs(F) < 1e-10) & (F != 0)).sum()), "smallest nonzero %.3e" % np.abs(F[F != 0]).min())
(R1,), _ = run_dense([F]); (R4,), _ = run_dense([F * 4.0])
k = min(fm, fn)
d = np.triu(R4[:k]) - 4.0 * np.triu(R1[:k])
print("dense entry point on the assembled front: entries that differ", int(np.count_nonzero(d)))
st = Stair[sym.Rp[f]:sym.Rp[f + 1]]
from tests.test_dense_gpu import run_dense as rd
(S1,), _ = rd([F], stairs=[st]); (S4,), _ = rd([F * 4.0], stairs=[st])
d2 = np.triu(S4[:k]) - 4.0 * np.triu(S1[:k])
print("the same with the front's staircase given: entries that differ", int(np.count_nonzero(d2)))
if np.count_nonzero(d2) or np.count_nonzero(d):
    dd = d2 if np.count_nonzero(d2) else d
    r, c = np.nonzero(dd); print("first differing row", r.min(), "cols", c[r == r.min()][:10])
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "culprit_front.npy"), F)
