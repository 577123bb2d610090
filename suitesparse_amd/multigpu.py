"""Sparse QR across the GPUs of one node: the assembly tree is mapped onto the ranks
proportionally (every subtree gets a share of the ranks that matches its share of the flops, a
front whose subtree spans several ranks is factorized by the first of them), every rank
factorizes its fronts level by level, and the only data that crosses ranks are contribution
blocks, point to point, device to device (RCCL over xGMI with the nccl backend; gloo on CPUs).
The single-GPU reference has no counterpart; this is its stage / limbo hand-off
(SPQR/SPQRGPU/spqrgpu_kernel.cpp:556-611,687-724) with "a later stage" replaced by "another GPU".
One process per GPU, launched by torch.distributed.run.
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


def _children(sym, f):
    return [int(c) for c in sym.Child[sym.Childp[f]:sym.Childp[f + 1]]]


def map_tree(sym, nparts):
    """Proportional mapping of the assembly tree onto `nparts` ranks.

    A forest with a list of ranks is split in two: the ranks in halves, the subtrees into two bins
    whose weights match the halves (heaviest first, each to the bin that is further below its
    share).  A single subtree with several ranks gives its root to the first of them and splits its
    children over all of them; a forest with one rank goes to that rank whole.  A subtree so heavy
    that it would unbalance any two bins keeps all the ranks, and its light siblings are dealt to the
    least loaded ranks afterwards.

    Returns (owner, level): owner[f] = rank that factorizes front f; level[f] = number of times the
    owner changes on the longest path from a leaf up to f.  The fronts (r, L) with owner r and level L
    form one call of the subset driver on rank r; a call needs blocks only from calls of lower levels.
    """
    nf = sym.nf
    w = front_weights(sym)
    Parent = sym.Parent
    sub = w.copy()
    for f in range(nf):                         # fronts are in postorder: children before parents
        p = Parent[f]
        if p < nf:
            sub[p] += sub[f]
    owner = np.zeros(nf, dtype=np.int64)
    level = np.zeros(nf, dtype=np.int64)
    if nf == 0:
        return owner, level
    load = np.zeros(max(nparts, 1))

    def whole(f, r):
        stack = [f]
        while stack:
            g = stack.pop()
            owner[g] = r
            stack.extend(_children(sym, g))
        load[r] += sub[f]

    def forest(kids, ranks):
        work = [(list(kids), list(ranks))]
        while work:
            kids, ranks = work.pop()
            if not kids:
                continue
            if len(ranks) == 1:
                for k in kids:
                    whole(k, ranks[0])
                continue
            kids.sort(key=lambda k: -sub[k])
            total = sum(sub[k] for k in kids)
            n1 = (len(ranks) + 1) // 2
            share1 = total * n1 / len(ranks)
            if len(kids) == 1 or sub[kids[0]] > 1.25 * share1:
                # one subtree carries the forest: its root goes to the first rank, its children
                # get all the ranks; light siblings are packed onto the least loaded ranks later
                head, rest = kids[0], kids[1:]
                owner[head] = ranks[0]
                load[ranks[0]] += w[head]
                work.append((_children(sym, head), ranks))
                late.extend((k, ranks) for k in rest)
                continue
            bins = ([], [])
            fill = [0.0, 0.0]
            want = [share1, total - share1]
            for k in kids:
                b = 0 if (want[0] - fill[0]) >= (want[1] - fill[1]) else 1
                bins[b].append(k)
                fill[b] += sub[k]
            work.append((bins[0], ranks[:n1]))
            work.append((bins[1], ranks[n1:]))

    late = []
    roots = _children(sym, nf)
    forest(roots, list(range(max(nparts, 1))))
    while late:
        k, ranks = late.pop()
        r = min(ranks, key=lambda q: load[q])
        whole(k, r)
    for f in range(nf):
        for c in _children(sym, f):
            level[f] = max(level[f], level[c] + (1 if owner[c] != owner[f] else 0))
    return owner, level


def mapping_report(sym, owner, level, nparts):
    """What the mapping promises, from the flop weights alone: the share of the flops in fronts whose
    subtree spans several ranks (factorized by one GPU while others may idle), and the speed-up bound
    total / critical path when every rank runs its calls in level order and a call starts once the
    calls it takes blocks from have ended."""
    nf = sym.nf
    w = front_weights(sym)
    total = float(w.sum())
    if nf == 0 or total <= 0:
        return dict(top_flops_share=0.0, speedup_bound=1.0, loads=[0.0] * nparts)
    spans = np.zeros(nf, dtype=bool)            # subtree of f holds fronts of another owner
    for f in range(nf):
        for c in _children(sym, f):
            if spans[c] or owner[c] != owner[f]:
                spans[f] = True
    nlev = int(level.max()) + 1
    cost = np.zeros((nparts, nlev))
    np.add.at(cost, (owner, level), w)
    # call (r, L) starts when (r, L-1) has ended and every call that hands it a block has ended
    need = {}
    for f in range(nf):
        p = sym.Parent[f]
        if p < nf and (owner[p] != owner[f] or level[p] != level[f]):
            need.setdefault((int(owner[p]), int(level[p])), set()).add((int(owner[f]), int(level[f])))
    end = np.zeros((nparts, nlev))
    for L in range(nlev):
        for r in range(nparts):
            start = end[r, L - 1] if L > 0 else 0.0
            for (rs, Ls) in need.get((r, L), ()):
                start = max(start, end[rs, Ls])
            end[r, L] = start + cost[r, L]
    crit = float(end.max())
    return dict(top_flops_share=float(w[spans].sum() / total), speedup_bound=total / crit if crit > 0 else 1.0,
                loads=[float(x) / total for x in cost.sum(axis=1)])


def _c_size(sym, f):
    return int(sym.Cm[f]) * int((sym.Rp[f + 1] - sym.Rp[f]) - (sym.Super[f + 1] - sym.Super[f]))


def _r_size(sym, f):
    rk = min(int(sym.Fm[f]), int(sym.Super[f + 1] - sym.Super[f]))
    return rk * (rk + 1) // 2 + rk * (int(sym.Rp[f + 1] - sym.Rp[f]) - rk)


import os as _os

BAND = int(_os.environ.get("GPUQR_CUT_BAND", "2048"))      # rows per band of a contribution block on the wire


def _c_shape(sym, f):
    return int(sym.Cm[f]), int((sym.Rp[f + 1] - sym.Rp[f]) - (sym.Super[f + 1] - sym.Super[f]))


def _band_sizes(sym, f):
    """Elements of every band of front f's contribution block as it travels: rows [r0, r1) from column r0 on."""
    cm, cn = _c_shape(sym, f)
    return [(min(r0 + BAND, cm) - r0) * (cn - r0) for r0 in range(0, cm, BAND) if (cn - r0) > 0]


def _bands_out(sym, f, t):
    """The bands of a compact cm-by-cn block (1-D tensor) as contiguous tensors."""
    cm, cn = _c_shape(sym, f)
    if cm * cn == 0:
        return []
    t2 = t.reshape(cm, cn)
    return [t2[r0:min(r0 + BAND, cm), r0:].contiguous().reshape(-1) for r0 in range(0, cm, BAND) if (cn - r0) > 0]


def _bands_in(sym, f, pieces, device):
    """The compact cm-by-cn block rebuilt from its bands (what lies left of a band's first column is never read)."""
    import torch
    cm, cn = _c_shape(sym, f)
    out = torch.zeros(cm * cn, dtype=torch.float64, device=device)
    if cm * cn == 0:
        return out
    o2 = out.reshape(cm, cn)
    k = 0
    for r0 in range(0, cm, BAND):
        if cn - r0 <= 0:
            continue
        r1 = min(r0 + BAND, cm)
        o2[r0:r1, r0:] = pieces[k].to(device).reshape(r1 - r0, cn - r0)
        k += 1
    return out


class _Csc:
    """A's arrays in the types the C ABI wants, made once per factorization."""

    def __init__(self, A):
        import scipy.sparse as sp
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
        self.Ai = np.ascontiguousarray(A.indices, dtype=np.int64)
        self.Ax = np.ascontiguousarray(A.data, dtype=np.float64)


def _bind_subset(L):
    if getattr(L, "_subset_bound", False):
        return
    L.spqrgpu_amd_factorize_subset.restype = C.c_int
    L.spqrgpu_amd_factorize_subset.argtypes = [C.c_void_p] * 6 + [C.POINTER(C.POINTER(spqr.Numeric))]
    L.spqrgpu_amd_factorize_subset_dev.restype = C.c_int
    L.spqrgpu_amd_factorize_subset_dev.argtypes = [C.c_void_p] * 7 + [C.POINTER(C.POINTER(spqr.Numeric))]
    L.spqrgpu_amd_subset_csize.restype = C.c_int64
    L.spqrgpu_amd_subset_csize.argtypes = [C.c_void_p, C.c_void_p]
    L._subset_bound = True


def _collect(sym, num, mine_c, want_R):
    """R blocks (copies) and the offsets of the leaving contribution blocks of a finished call."""
    Rblocks, Coff = {}, {}
    nf = sym.nf
    if want_R:
        fnn = np.diff(sym.Rp)
        fpp = np.diff(sym.Super)
        stack = np.ctypeslib.as_array(num.Rstack, shape=(max(num.rsize, 1),))
        for f in np.nonzero(mine_c)[0]:
            f = int(f)
            rk = min(int(sym.Fm[f]), int(fpp[f]))
            size = rk * (rk + 1) // 2 + rk * (int(fnn[f]) - rk)
            off = num.Rblock[f]
            Rblocks[f] = stack[off:off + size].copy()
    # (only a few of a call's fronts have a block that leaves it)
    cblock = np.ctypeslib.as_array(num.Cblock, shape=(max(nf, 1),))[:nf]
    for f in np.nonzero((cblock >= 0) & (np.asarray(mine_c[:nf]) != 0))[0]:
        Coff[int(f)] = int(cblock[f])
    stats = dict(kernel_ms=float(num.gpuKernelTime), flops=int(num.gpuFlops), launches=int(num.gpuNumKernelLaunches),
                 groups=int(num.numStagesRun))
    return Rblocks, Coff, stats


def gpu_factor_subset(sym, A, mine, limbo, want_R=True):
    """spqrgpu_amd_factorize_subset on the current HIP device, contribution blocks in host memory on
    both sides.  Returns (Rblocks, Cblocks, stats): dicts front -> packed R block / compact C block."""
    L = spqr._bind()
    _bind_subset(L)
    csc = A if isinstance(A, _Csc) else _Csc(A)
    nf = sym.nf
    mine_c = np.ascontiguousarray(mine, dtype=np.int8)
    ptrs = (C.c_void_p * max(nf, 1))()
    keep = []
    for c, blk in limbo.items():
        b = np.ascontiguousarray(blk, dtype=np.float64)
        keep.append(b)
        ptrs[c] = b.ctypes.data
    out = C.POINTER(spqr.Numeric)()
    code = L.spqrgpu_amd_factorize_subset(sym._h, csc.Ap.ctypes.data, csc.Ai.ctypes.data, csc.Ax.ctypes.data,
                                          mine_c.ctypes.data, C.cast(ptrs, C.c_void_p), C.byref(out))
    if code != 0:
        raise RuntimeError("spqrgpu_amd_factorize_subset failed with code %d" % code)
    num = out.contents
    Rblocks, Coff, stats = _collect(sym, num, mine_c, want_R)
    Cblocks = {}
    for f, coff in Coff.items():
        csz = _c_size(sym, f)
        Cblocks[f] = np.ctypeslib.as_array(num.Cstack, shape=(max(coff + csz, 1),))[coff:coff + csz].copy()
    L.spqr_amd_numeric_free(out)
    return Rblocks, Cblocks, stats


def gpu_factor_subset_dev(sym, A, mine, limbo, want_R=True):
    """spqrgpu_amd_factorize_subset_dev: the incoming blocks (`limbo`: front -> torch tensor on the
    current device) are assembled from where they lie, the leaving ones are written into one new device
    tensor.  Returns (Rblocks, Cblocks, stats) with Cblocks: front -> view into that tensor."""
    import torch
    L = spqr._bind()
    _bind_subset(L)
    csc = A if isinstance(A, _Csc) else _Csc(A)
    nf = sym.nf
    mine_c = np.ascontiguousarray(mine, dtype=np.int8)
    ptrs = (C.c_void_p * max(nf, 1))()
    for c, t in limbo.items():
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == _c_size(sym, c)
        ptrs[c] = t.data_ptr() if t.numel() > 0 else None
    csize = int(L.spqrgpu_amd_subset_csize(sym._h, mine_c.ctypes.data))
    cout = torch.empty(max(csize, 1), dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
    torch.cuda.synchronize()            # the incoming blocks were written on torch's streams
    out = C.POINTER(spqr.Numeric)()
    code = L.spqrgpu_amd_factorize_subset_dev(sym._h, csc.Ap.ctypes.data, csc.Ai.ctypes.data, csc.Ax.ctypes.data,
                                              mine_c.ctypes.data, C.cast(ptrs, C.c_void_p), cout.data_ptr(), C.byref(out))
    if code != 0:
        raise RuntimeError("spqrgpu_amd_factorize_subset_dev failed with code %d" % code)
    num = out.contents
    Rblocks, Coff, stats = _collect(sym, num, mine_c, want_R)
    Cblocks = {f: cout[coff:coff + _c_size(sym, f)] for f, coff in Coff.items()}
    L.spqr_amd_numeric_free(out)
    return Rblocks, Cblocks, stats


def factorize_distributed(sym, A, factor_subset=None, gather_R=True, trace=None):
    """Factorize A = QR over the ranks of the default torch.distributed group.

    Every rank calls this with the same `sym` and `A`.  For L = 0, 1, ...: each rank factorizes its
    fronts of level L (one call of the subset driver), then the contribution blocks that leave those
    calls travel to the rank that owns the parent, all of a level's transfers in one batch of
    point-to-point sends and receives.  With the nccl backend the blocks go device to device; with
    gloo they are staged through host tensors.  `factor_subset` (sym, A, mine, limbo) -> (R, C, stats)
    replaces the GPU driver by a host one working on numpy blocks (the CPU tests pass the oracle).
    With gather_R, rank 0 also collects every R block and returns R (scipy CSC); other ranks return
    None for it.  `trace` (a callable taking a string) is told where the call is, for post-mortems of
    multi-process runs.  Returns (R or None, info dict).
    """
    if trace is None:
        def trace(msg):
            pass
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
        nccl = dist.get_backend() == "nccl"
    else:
        rank, world, nccl = 0, 1, False
    host_blocks = factor_subset is not None      # numpy blocks, host driver
    cdev = torch.device("cuda", torch.cuda.current_device()) if (nccl or not host_blocks) else torch.device("cpu")
    wire = cdev if nccl else torch.device("cpu")        # where a tensor must live to be sent

    nf = sym.nf
    Parent = sym.Parent
    csc = A if (host_blocks or isinstance(A, _Csc)) else _Csc(A)
    # the map of the tree onto the ranks depends on the symbolic object and the number of ranks only:
    # made once, kept with the object (a caller that factorizes again pays for the numeric work alone)
    cache = sym.__dict__.setdefault("_rank_maps", {})
    if world not in cache:
        owner, level = map_tree(sym, world)
        nlev = int(level.max()) + 1 if nf > 0 else 0
        # fronts whose block leaves the call that makes it, by level
        leaving = [[] for _ in range(nlev)]
        for f in range(nf):
            p = Parent[f]
            if p < nf and (owner[p] != owner[f] or level[p] != level[f]):
                leaving[level[f]].append(f)
        static = dict(levels=nlev,
                      n_cut=sum(1 for L in leaving for f in L if owner[Parent[f]] != owner[f]),
                      cut_bytes=8 * sum(sum(_band_sizes(sym, f)) for L in leaving for f in L if owner[Parent[f]] != owner[f]),
                      cut_bytes_rectangles=8 * sum(_c_size(sym, f) for L in leaving for f in L if owner[Parent[f]] != owner[f]))
        static.update(mapping_report(sym, owner, level, world))
        cache[world] = (owner, level, nlev, leaving, static)
    owner, level, nlev, leaving, static = cache[world]
    info = dict(rank=rank, world=world, calls=[], **static)

    held = {}           # front -> its contribution block, waiting here for the parent's call
    Rall = {}
    for L in range(nlev):
        mine = (owner == rank) & (level == L)
        Cout = {}
        trace("level %d: %d fronts" % (L, int(mine.sum())))
        if mine.any():
            # the blocks this call takes in: children of its fronts that another call made (kept with the map)
            inc = cache.setdefault(("incoming", world, rank, L), None)
            if inc is None:
                inc = [c for f in np.nonzero(mine)[0] for c in _children(sym, int(f)) if not mine[c]]
                cache[("incoming", world, rank, L)] = inc
            limbo = {c: held[c] for c in inc}
            if host_blocks:
                R, Cout, st = factor_subset(sym, csc, mine, limbo)
            else:
                R, Cout, st = gpu_factor_subset_dev(sym, csc, mine, limbo, want_R=gather_R)
            for c in limbo:
                del held[c]
            if gather_R:
                Rall.update(R)
            info["calls"].append(dict(level=L, fronts=int(mine.sum()), **st))
        # ---- the blocks that left level L, to whoever factorizes the parent ------------------------
        # A contribution block is upper trapezoidal (entries left of the diagonal are never read,
        # packAssemble.cu:47-55), so it travels as bands of BAND rows, each from its own first column on:
        # a little over half of the cm-by-cn rectangle for a square block.
        ops, landing = [], []
        for f in leaving[L]:                            # increasing f on every rank: the pairs match up
            src, dst = int(owner[f]), int(owner[Parent[f]])
            if src == rank and dst == rank:
                held[f] = Cout[f]
            elif src == rank:
                blk = Cout[f]
                t = torch.from_numpy(np.ascontiguousarray(blk)) if host_blocks else blk
                for piece in _bands_out(sym, f, t):
                    ops.append(dist.P2POp(dist.isend, piece.to(wire), dst))
            elif dst == rank:
                pieces = [torch.empty(nel, dtype=torch.float64, device=wire) for nel in _band_sizes(sym, f)]
                landing.append((f, pieces))
                for piece in pieces:
                    ops.append(dist.P2POp(dist.irecv, piece, src))
        trace("level %d: exchange, %d transfers" % (L, len(ops)))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if wire.type == "cuda":
                torch.cuda.synchronize()
        for f, pieces in landing:
            t = _bands_in(sym, f, pieces, cdev if not host_blocks else torch.device("cpu"))
            held[f] = t.numpy() if host_blocks else t

    trace("levels done")
    if not gather_R:
        return None, info

    # ---- R blocks to rank 0 (host memory; not part of the timed path) -------------------------------
    if world > 1:
        if rank == 0:
            for src in range(1, world):
                fs = [f for f in range(nf) if owner[f] == src]
                total = sum(_r_size(sym, f) for f in fs)
                if total == 0:
                    continue
                buf = torch.empty(total, dtype=torch.float64, device=wire)
                dist.recv(buf, src=src)
                flat = buf.cpu().numpy()
                off = 0
                for f in fs:
                    sz = _r_size(sym, f)
                    Rall[f] = flat[off:off + sz]
                    off += sz
        else:
            fs = [f for f in range(nf) if owner[f] == rank]
            total = sum(_r_size(sym, f) for f in fs)
            if total > 0:
                flat = np.concatenate([np.asarray(Rall[f]).ravel() for f in fs])
                dist.send(torch.from_numpy(flat).to(wire), dst=0)
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


def map_tree_c(sym, nparts):
    """spqr_amd_map_tree: the same mapping made by the library (what spqrgpu_amd_factorize_multi uses).
    Returns (owner, level, top_flops_share, speedup_bound)."""
    L = spqr._bind()
    L.spqr_amd_map_tree.restype = C.c_int
    L.spqr_amd_map_tree.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    nf = int(sym.nf)
    owner = np.zeros(max(nf, 1), np.int64)
    level = np.zeros(max(nf, 1), np.int64)
    top, bound = C.c_double(0), C.c_double(0)
    code = L.spqr_amd_map_tree(sym._h, int(nparts), owner.ctypes.data, level.ctypes.data, C.byref(top), C.byref(bound))
    if code != 0:
        raise RuntimeError("spqr_amd_map_tree failed with code %d" % code)
    return owner[:nf], level[:nf], float(top.value), float(bound.value)


def factorize_multi(sym, A, ndev, devices=None, want_R=True, Sx=None):
    """spqrgpu_amd_factorize_multi: one process, ndev parts of the assembly tree on the HIP devices
    `devices` (default 0 .. ndev-1; parts may share a device).  Returns a result like spqr.factorize.
    With Sx (the values of S = A(P,Q) in row form, what an SPQR adapter holds) A is not looked at."""
    import scipy.sparse as sp
    L = spqr._bind()
    L.spqrgpu_amd_factorize_multi.restype = C.c_int
    L.spqrgpu_amd_factorize_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                              C.POINTER(C.POINTER(spqr.Numeric))]
    devs = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
    out = C.POINTER(spqr.Numeric)()
    if Sx is not None:
        Sx = np.ascontiguousarray(Sx, dtype=np.float64)
        assert Sx.size == sym.anz
        code = L.spqrgpu_amd_factorize_multi(sym._h, None, None, Sx.ctypes.data, int(ndev),
                                             None if devs is None else devs.ctypes.data, C.byref(out))
    else:
        csc = A if isinstance(A, _Csc) else _Csc(A)
        code = L.spqrgpu_amd_factorize_multi(sym._h, csc.Ap.ctypes.data, csc.Ai.ctypes.data, csc.Ax.ctypes.data, int(ndev),
                                             None if devs is None else devs.ctypes.data, C.byref(out))
    if code != 0:
        raise RuntimeError("spqrgpu_amd_factorize_multi failed with code %d" % code)
    num = out.contents
    n = int(sym.n)
    res = spqr.NumericResult()
    res.rank = int(num.sumfrank)
    res.maxfrank = int(num.maxfrank)
    res.gpuKernelTime = float(num.gpuKernelTime)
    res.gpuNumKernelLaunches = int(num.gpuNumKernelLaunches)
    res.numStagesRun = int(num.numStagesRun)
    res.R = None
    if want_R:
        Rp = np.zeros(n + 1, dtype=np.int64)
        L.spqr_amd_rconvert(sym._h, out, Rp.ctypes.data, None, None)
        Ri = np.zeros(max(int(Rp[n]), 1), dtype=np.int64)
        Rx = np.zeros(max(int(Rp[n]), 1), dtype=np.float64)
        L.spqr_amd_rconvert(sym._h, out, Rp.ctypes.data, Ri.ctypes.data, Rx.ctypes.data)
        res.R = sp.csc_matrix((Rx[:Rp[n]], Ri[:Rp[n]], Rp), shape=(max(res.rank, 0), n))
    L.spqr_amd_numeric_free(out)
    return res
