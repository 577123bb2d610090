"""Sparse QR across the GPUs of one node: independent subtrees of the assembly tree go to
different ranks, the fronts above the cut are factorized on rank 0, and the contribution
blocks of the subtree roots travel to rank 0 by point-to-point transfers (RCCL over xGMI with
the nccl backend, gloo on CPUs).  The single-GPU reference has no counterpart; this is its
stage / limbo hand-off (SPQR/SPQRGPU/spqrgpu_kernel.cpp:556-611,687-724) with "a later stage"
replaced by "another GPU".  One process per GPU, launched by torch.distributed.run.
"""
import ctypes as C

import numpy as np

from . import spqr


def front_weights(sym):
    """Flop estimate per front (dense Householder count on its fm-by-fn shape, fp pivots)."""
    nf = sym.nf
    fm = sym.Fm[:nf].astype(np.float64)
    fn = np.diff(sym.Rp)[:nf].astype(np.float64)
    fp = np.diff(sym.Super)[:nf].astype(np.float64)
    k = np.minimum(fm, fp)
    return 4.0 * k * fm * fn - 2.0 * k * k * (fm + fn) + (4.0 / 3.0) * k ** 3 + 1.0


def partition_subtrees(sym, nparts, oversplit=4):
    """Cut the assembly tree into subtrees and deal them to `nparts` ranks.

    Start from the roots; repeatedly replace the heaviest subtree by its children (its root joins
    the "top" set kept on rank 0) while there are fewer subtrees than ranks or the heaviest one
    outweighs a rank's fair share of what is left below the cut (never beyond `oversplit * nparts`
    subtrees); then assign subtrees to ranks, heaviest first, each to the least loaded rank (the
    top set is charged to rank 0 first).  Returns (owner, top): owner[f] = rank, top[f] = bool.
    """
    nf = sym.nf
    w = front_weights(sym)
    Childp, Child, Parent = sym.Childp, sym.Child, sym.Parent
    sub = w.copy()
    for f in range(nf):                         # fronts are in postorder: children before parents
        p = Parent[f]
        if p < nf:
            sub[p] += sub[f]
    top = np.zeros(nf, dtype=bool)
    owner = np.zeros(nf, dtype=np.int64)
    if nparts <= 1 or nf == 0:
        return owner, top
    roots = [int(c) for c in Child[Childp[nf]:Childp[nf + 1]]]
    subtrees = list(roots)
    while len(subtrees) < oversplit * nparts:
        subtrees.sort(key=lambda f: sub[f])
        below = sum(sub[f] for f in subtrees)
        if len(subtrees) >= nparts and sub[subtrees[-1]] <= 1.1 * below / nparts:
            break
        pick = None
        for f in reversed(subtrees):
            if Childp[f + 1] > Childp[f]:
                pick = f
                break
        if pick is None:
            break
        subtrees.remove(pick)
        top[pick] = True
        subtrees += [int(c) for c in Child[Childp[pick]:Childp[pick + 1]]]
    load = np.zeros(nparts)
    load[0] = w[top].sum()
    root_owner = {}
    for f in sorted(subtrees, key=lambda f: -sub[f]):
        r = int(np.argmin(load))
        root_owner[f] = r
        load[r] += sub[f]
    # propagate ownership down the subtrees (parents have larger indices than children)
    own = np.full(nf, -1, dtype=np.int64)
    for f in subtrees:
        own[f] = root_owner[f]
    for f in range(nf - 1, -1, -1):
        if top[f]:
            own[f] = 0
        elif own[f] < 0:
            own[f] = own[Parent[f]]
    return own, top


def gpu_factor_subset(sym, A, mine, limbo):
    """Default backend: spqrgpu_amd_factorize_subset on the current HIP device.
    Returns (Rblocks, Cblocks): dicts front -> packed R block / compact C block."""
    import scipy.sparse as sp
    L = spqr._bind()
    A = sp.csc_matrix(A)
    A.sort_indices()
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
    Ai = np.ascontiguousarray(A.indices, dtype=np.int64)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    nf = sym.nf
    mine_c = np.ascontiguousarray(mine, dtype=np.int8)
    ptrs = (C.c_void_p * max(nf, 1))()
    keep = []
    for c, blk in limbo.items():
        b = np.ascontiguousarray(blk, dtype=np.float64)
        keep.append(b)
        ptrs[c] = b.ctypes.data
    out = C.POINTER(spqr.Numeric)()
    L.spqrgpu_amd_factorize_subset.restype = C.c_int
    L.spqrgpu_amd_factorize_subset.argtypes = [C.c_void_p] * 6 + [C.POINTER(C.POINTER(spqr.Numeric))]
    code = L.spqrgpu_amd_factorize_subset(sym._h, Ap.ctypes.data, Ai.ctypes.data, Ax.ctypes.data,
                                          mine_c.ctypes.data, C.cast(ptrs, C.c_void_p), C.byref(out))
    if code != 0:
        raise RuntimeError("spqrgpu_amd_factorize_subset failed with code %d" % code)
    num = out.contents
    Rblocks, Cblocks = {}, {}
    fm = sym.Fm
    fnn = np.diff(sym.Rp)
    fpp = np.diff(sym.Super)
    for f in np.nonzero(mine_c)[0]:
        rk = min(int(fm[f]), int(fpp[f]))
        size = rk * (rk + 1) // 2 + rk * (int(fnn[f]) - rk)
        off = num.Rblock[f]
        Rblocks[int(f)] = np.ctypeslib.as_array(num.Rstack, shape=(max(num.rsize, 1),))[off:off + size].copy()
        coff = num.Cblock[f]
        if coff >= 0:
            csz = int(sym.Cm[f]) * (int(fnn[f]) - int(fpp[f]))
            Cblocks[int(f)] = np.ctypeslib.as_array(num.Cstack, shape=(max(coff + csz, 1),))[coff:coff + csz].copy()
    stats = dict(kernel_ms=float(num.gpuKernelTime), flops=int(num.gpuFlops), launches=int(num.gpuNumKernelLaunches))
    L.spqr_amd_numeric_free(out)
    return Rblocks, Cblocks, stats


def _c_size(sym, f):
    return int(sym.Cm[f]) * int((sym.Rp[f + 1] - sym.Rp[f]) - (sym.Super[f + 1] - sym.Super[f]))


def _r_size(sym, f):
    rk = min(int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f]))
    return rk * (rk + 1) // 2 + rk * (int(sym.Rp[f + 1] - sym.Rp[f]) - rk)


def factorize_distributed(sym, A, factor_subset=None, gather_R=True, oversplit=4):
    """Factorize A = QR over the ranks of the default torch.distributed group.

    Every rank calls this with the same `sym` and `A`.  Phase 1: each rank factorizes its
    subtrees.  Exchange: the subtree roots' contribution blocks go to rank 0, one message per
    rank.  Phase 2: rank 0 factorizes the top fronts.  With gather_R, rank 0 also collects every
    R block and returns R (scipy CSC); other ranks return None for it.
    Returns (R or None, info dict).
    """
    import torch
    import torch.distributed as dist

    if factor_subset is None:
        factor_subset = gpu_factor_subset
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    else:
        rank, world, dev = 0, 1, torch.device("cpu")
    nf = sym.nf
    owner, top = partition_subtrees(sym, world, oversplit)
    Parent = sym.Parent
    # fronts whose contribution block crosses the cut: parent is a top front, front itself is not
    cut = [f for f in range(nf) if not top[f] and Parent[f] < nf and top[Parent[f]]]

    mine1 = (owner == rank) & ~top
    R1, C1, st1 = factor_subset(sym, A, mine1, {})
    info = dict(rank=rank, world=world, phase1=st1, n_top=int(top.sum()), n_cut=len(cut),
                cut_bytes=8 * sum(_c_size(sym, f) for f in cut))

    # ---- contribution blocks of the cut to rank 0: one flat message per sending rank -------------
    limbo = {}
    if rank == 0:
        for f in cut:
            if owner[f] == 0:
                limbo[f] = C1[f]
        for src in range(1, world):
            fs = [f for f in cut if owner[f] == src]
            total = sum(_c_size(sym, f) for f in fs)
            if total == 0:
                continue
            buf = torch.empty(total, dtype=torch.float64, device=dev)
            dist.recv(buf, src=src)
            flat = buf.cpu().numpy()
            off = 0
            for f in fs:
                sz = _c_size(sym, f)
                limbo[f] = flat[off:off + sz]
                off += sz
    else:
        fs = [f for f in cut if owner[f] == rank]
        total = sum(_c_size(sym, f) for f in fs)
        if total > 0:
            flat = np.concatenate([np.asarray(C1[f]).ravel() for f in fs]) if fs else np.zeros(0)
            dist.send(torch.from_numpy(flat).to(dev), dst=0)

    R2 = {}
    if rank == 0 and top.any():
        R2, C2, st2 = factor_subset(sym, A, top, limbo)
        info["phase2"] = st2

    if not gather_R:
        return None, info

    # ---- R blocks to rank 0 ---------------------------------------------------------------------
    Rall = dict(R1)
    Rall.update(R2)
    if world > 1:
        if rank == 0:
            for src in range(1, world):
                fs = [f for f in range(nf) if owner[f] == src and not top[f]]
                total = sum(_r_size(sym, f) for f in fs)
                if total == 0:
                    continue
                buf = torch.empty(total, dtype=torch.float64, device=dev)
                dist.recv(buf, src=src)
                flat = buf.cpu().numpy()
                off = 0
                for f in fs:
                    sz = _r_size(sym, f)
                    Rall[f] = flat[off:off + sz]
                    off += sz
        else:
            fs = [f for f in range(nf) if owner[f] == rank and not top[f]]
            total = sum(_r_size(sym, f) for f in fs)
            if total > 0:
                flat = np.concatenate([np.asarray(R1[f]).ravel() for f in fs])
                dist.send(torch.from_numpy(flat).to(dev), dst=0)
            return None, info
    return blocks_to_csc(sym, Rall), info


def blocks_to_csc(sym, Rblocks):
    """Packed per-front R blocks -> R in compressed-column form (spqr_amd_rconvert_blocks)."""
    import scipy.sparse as sp
    L = spqr._bind()
    nf, n = sym.nf, sym.n
    Rblock = np.full(max(nf, 1), -1, dtype=np.int64)
    parts, off = [], 0
    for f in range(nf):
        if f in Rblocks:
            Rblock[f] = off
            parts.append(np.asarray(Rblocks[f], dtype=np.float64).ravel())
            off += parts[-1].size
    Rstack = np.concatenate(parts) if parts else np.zeros(1)
    Rdead = np.zeros(max(n, 1), dtype=np.int8)
    for f in range(nf):
        fm, fp = int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f])
        if fp > fm:
            Rdead[sym.Super[f] + fm:sym.Super[f] + fp] = 1
    Rp = np.zeros(n + 1, dtype=np.int64)
    L.spqr_amd_rconvert_blocks.restype = C.c_int
    L.spqr_amd_rconvert_blocks.argtypes = [C.c_void_p] * 7
    L.spqr_amd_rconvert_blocks(sym._h, Rstack.ctypes.data, Rblock.ctypes.data, Rdead.ctypes.data,
                               Rp.ctypes.data, None, None)
    Ri = np.zeros(max(int(Rp[n]), 1), dtype=np.int64)
    Rx = np.zeros(max(int(Rp[n]), 1), dtype=np.float64)
    L.spqr_amd_rconvert_blocks(sym._h, Rstack.ctypes.data, Rblock.ctypes.data, Rdead.ctypes.data,
                               Rp.ctypes.data, Ri.ctypes.data, Rx.ctypes.data)
    rows = int(sum(min(int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f])) for f in range(nf)))
    return sp.csc_matrix((Rx[:Rp[n]], Ri[:Rp[n]], Rp), shape=(rows, n))
