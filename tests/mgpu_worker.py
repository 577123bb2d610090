"""Worker of the multi-process tests: factorizes one problem over the ranks of a gloo (CPU) or
nccl (GPU) group.  On CPU the per-rank numeric work is done by the oracle (test infrastructure);
the partitioning, the exchange and the assembly of R are the product's code either way."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_factor_subset(sym, A, mine, limbo):
    """CPU stand-in for spqrgpu_amd_factorize_subset built from the oracle's primitives."""
    from oracle import oracle
    from suitesparse_amd import spqr
    nf = sym.nf
    Sx = spqr.s_values(sym, A)
    Stair, Rjmap, Rimap, cpuS = oracle.build_assembly_maps(sym, Sx)
    Scount = np.array([sym.Sp[sym.Sleft[sym.Super[f + 1]]] - sym.Sp[sym.Sleft[sym.Super[f]]] for f in range(nf)])
    SOff = np.concatenate([[0], np.cumsum(Scount)])          # Post is the identity here
    fronts, Rb, Cb = {}, {}, {}
    for f in range(nf):
        if not mine[f]:
            continue
        fm, fn, fp = int(sym.Fm[f]), int(sym.Rp[f + 1] - sym.Rp[f]), int(sym.Super[f + 1] - sym.Super[f])
        F = np.zeros(fm * fn)
        e = cpuS[SOff[f]:SOff[f + 1]]
        F[e["findex"]] = e["value"]
        F = F.reshape(fm, fn)
        for c in sym.Child[sym.Childp[f]:sym.Childp[f + 1]]:
            c = int(c)
            cm = int(sym.Cm[c])
            cn = int(sym.Rp[c + 1] - sym.Rp[c]) - int(sym.Super[c + 1] - sym.Super[c])
            if cm * cn == 0:
                continue
            Cblk = fronts.pop(c) if mine[c] else np.asarray(limbo[c]).reshape(cm, cn)
            ri = Rimap[sym.RimapOffsets[c]:sym.RimapOffsets[c] + cm]
            rj = Rjmap[sym.RjmapOffsets[c]:sym.RjmapOffsets[c] + cn]
            for i in range(cm):
                F[ri[i], rj[i:]] = Cblk[i, i:]
        F = oracle.front_qr(F, Stair[sym.Rp[f]:sym.Rp[f + 1]])
        rk = min(fm, fp)
        Rb[f] = np.concatenate([F[:min(j + 1, rk), j] for j in range(fn)]) if fn > 0 else np.zeros(0)
        cm, cn = int(sym.Cm[f]), fn - fp
        Cblk = np.triu(F[rk:rk + cm, fp:]) if cm * cn > 0 else np.zeros((cm, cn))
        p = int(sym.Parent[f])
        if p < nf and mine[p]:
            fronts[f] = Cblk
        elif p < nf:
            Cb[f] = Cblk.ravel().copy()
    return Rb, Cb, dict(kernel_ms=0.0, flops=0, launches=0)


def main():
    import faulthandler
    faulthandler.enable()               # a rank that dies of a fatal signal says where (Python frames) on stderr
    import torch
    import torch.distributed as dist
    from suitesparse_amd import multigpu, problems, spqr

    mode = sys.argv[1]                  # gloo (oracle numerics), gloo-gpu (GPU numerics, gloo exchange), nccl
    backend = "nccl" if mode == "nccl" else "gloo"
    out_path = sys.argv[2]
    shape = tuple(int(x) for x in sys.argv[3].split("x"))
    rank = int(os.environ["RANK"])
    t_begin = __import__("time").time()

    def trace(msg):
        # where this rank is, for the parent test's report when a run hangs
        path = os.environ.get("MGPU_TRACE")
        if path:
            with open("%s.%d" % (path, rank), "a") as f:
                f.write("%7.2f %s\n" % (__import__("time").time() - t_begin, msg))

    trace("start, mode %s" % mode)
    if os.environ.get("MGPU_SEGV_TRACE"):
        # debug aid: print the native call stack if this rank dies of a segmentation fault
        import ctypes
        ctypes.CDLL(os.environ["MGPU_SEGV_TRACE"])
    if mode == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    elif mode == "gloo-gpu":
        torch.cuda.set_device(0)        # all ranks share the one GPU of the test box
    trace("torch imported, device set")
    dist.init_process_group(backend)
    trace("process group up")
    A = problems.stacked_ls(shape)
    Q = problems.nested_dissection(shape, leaf=3)
    sym = spqr.analyze(A, Q, relax=0, gpuMemorySize=2**34)
    fs = oracle_factor_subset if mode == "gloo" else None
    trace("analysed")
    R, info = multigpu.factorize_distributed(sym, A, factor_subset=fs, trace=trace)
    trace("factorized")
    if rank == 0:
        np.savez(out_path, data=R.data, indices=R.indices, indptr=R.indptr, shape=np.array(R.shape),
                 n_cut=info["n_cut"], levels=info["levels"], bound=info["speedup_bound"], Qfill=sym.Qfill)
    trace("saved")
    dist.barrier()
    trace("barrier passed")
    dist.destroy_process_group()
    trace("done")


if __name__ == "__main__":
    main()
