"""gpuqr_factor_sparse_noalloc against gpuqr_factor_sparse, through the C ABI: the same fronts must
come out bit for bit (same kernels, same task records; only where the records and the VT tiles
live differs), on two calls in a row, the second with a smaller input, so that what the first call
left in the fixed workspace is seen not to leak into the second.  Then the size limit."""
import ctypes as C

import numpy as np
import pytest

import suitesparse_amd as ss
from suitesparse_amd import _lib, engine, problems, spqr

pytestmark = pytest.mark.gpu


def _bind():
    L = _lib.lib()
    L.gpuqr_init.restype = C.c_int
    L.gpuqr_factor_sparse_noalloc.argtypes = [C.POINTER(_lib.FrontC), C.c_int64, C.POINTER(_lib.Stats)]
    L.gpuqr_factor_sparse_noalloc.restype = C.c_int
    L.gpuqr_noalloc_workspace_bytes.restype = C.c_size_t
    L.gpuqr_noalloc_last_need.restype = C.c_size_t
    return L


def _sparse_batch(shape, leaf, with_map=False):
    """The fronts of a small multifrontal problem as one engine batch (S assembly, pack assembly,
    staircases), built from the C ABI's own assembly maps.  Returns (make, nbytes): make(buf) gives the
    Front list for a device arena of nbytes bytes."""
    import torch
    dev = torch.device("cuda:0")
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=leaf), gpuMemorySize=2**32)
    Stair, Rjmap, Rimap, cpuS = spqr.spqrgpu_buildAssemblyMaps(sym, spqr.s_values(sym, A))
    nf = sym.nf
    fn = np.diff(sym.Rp)[:nf]
    fp = np.diff(sym.Super)[:nf]
    fm = sym.Fm[:nf]
    off = np.zeros(nf + 1, dtype=np.int64)
    soff = np.zeros(nf + 1, dtype=np.int64)
    order = [int(f) for f in sym.Post[:nf]]
    rel = {f: k for k, f in enumerate(order)}
    for k, f in enumerate(order):
        off[k + 1] = off[k] + fm[f] * fn[f]
        soff[k + 1] = soff[k] + (sym.Sp[sym.Sleft[sym.Super[f + 1]]] - sym.Sp[sym.Sleft[sym.Super[f]]])
    dS = torch.from_numpy(cpuS.view(np.float64).copy()).to(dev)       # 16-byte records as pairs of 8-byte words
    dRj = torch.from_numpy(Rjmap.copy()).to(dev)
    dRi = torch.from_numpy(Rimap.copy()).to(dev)
    keep = [dS, dRj, dRi, Stair]

    def make(buf):
        fronts = []
        for k, f in enumerate(order):
            pid = int(sym.Parent[f])
            fr = ss.Front(k, rel[pid] if pid < nf else ss.EMPTY, int(fm[f]), int(fn[f]), int(fp[f]))
            fr.rank = min(int(fm[f]), int(fp[f]))
            fr.Stair = Stair[sym.Rp[f]:sym.Rp[f + 1]]
            fr.gpuF = buf.data_ptr() + 8 * int(off[k])
            m = fr.sparseMeta
            m.isSparse = True
            m.nc = int(sym.Childp[f + 1] - sym.Childp[f])
            m.gpuS = dS.data_ptr() + 16 * int(soff[k])
            m.Scount = int(soff[k + 1] - soff[k])
            m.cn = int(fn[f] - fp[f])
            m.cm = int(sym.Cm[f])
            if pid < nf:
                m.pn = int(fn[pid])
                m.gpuRjmap = dRj.data_ptr() + 4 * int(sym.RjmapOffsets[f])
                m.gpuRimap = dRi.data_ptr() + 4 * int(sym.RimapOffsets[f])
                if with_map:        # the host copy of the row map: contribution blocks then travel row block by row block
                    m.cpuRimap = Rimap[int(sym.RimapOffsets[f]):int(sym.RimapOffsets[f]) + m.cm]
                m.gpuC = fr.gpuF + 8 * (fr.rank * int(fn[f]) + int(fp[f]))
            fronts.append(fr)
        return fronts

    return make, int(off[nf]), keep


def _dense_batch(mats):
    import torch
    dev = torch.device("cuda:0")
    flat = torch.from_numpy(np.concatenate([np.ascontiguousarray(A).ravel() for A in mats])).to(dev)

    def make(buf):
        fronts, o = [], 0
        for k, A in enumerate(mats):
            fr = ss.Front(k, ss.EMPTY, *A.shape)
            fr.gpuF = buf.data_ptr() + 8 * o
            o += A.size
            fronts.append(fr)
        return fronts

    return make, flat


def _both(L, make, init):
    """Run both entry points on their own copy of the arena `init`; return the two results."""
    import torch
    outs = []
    for noalloc in (False, True):
        buf = init.clone()
        torch.cuda.synchronize()
        keep = []
        fronts = make(buf)
        arr = engine._pack(fronts, keep)
        st = _lib.Stats()
        if noalloc:
            code = L.gpuqr_factor_sparse_noalloc(arr, len(fronts), C.byref(st))
        else:
            code = L.gpuqr_factor_sparse(0, arr, len(fronts), None, None, None, C.byref(st))
        assert code == _lib.GPUQR_SUCCESS, (noalloc, code)
        assert st.numLaunches > 0
        torch.cuda.synchronize()
        outs.append(buf.cpu().numpy())
    return outs


def test_noalloc_matches_the_allocating_entry_point_twice_then_refuses_too_much():
    import torch
    dev = torch.device("cuda:0")
    L = _bind()
    assert L.gpuqr_init() == _lib.GPUQR_SUCCESS
    assert L.gpuqr_noalloc_workspace_bytes() == 32 << 20

    # first call: a multifrontal batch (S assembly, contribution blocks, staircases, 3 levels of tree)
    make, nwords, keep = _sparse_batch((24, 24), leaf=4)
    ref, got = _both(L, make, torch.zeros(nwords, dtype=torch.float64, device=dev))
    assert np.any(ref != 0)
    assert np.array_equal(ref, got)
    need1 = L.gpuqr_noalloc_last_need()

    # second call: a smaller input (two dense fronts), right behind the first
    rng = np.random.default_rng(11)
    make2, flat = _dense_batch([rng.standard_normal((150, 70)), rng.standard_normal((40, 90))])
    ref2, got2 = _both(L, make2, flat)
    assert np.array_equal(ref2, got2)
    assert 0 < L.gpuqr_noalloc_last_need() < need1

    # and once more the first, to see that the smaller call left nothing behind that matters
    ref3, got3 = _both(L, make, torch.zeros(nwords, dtype=torch.float64, device=dev))
    assert np.array_equal(ref3, ref) and np.array_equal(got3, ref)

    # the same batch with the host copies of the row maps (the parents then start on the rows their
    # children have finished): another schedule, the same arithmetic, in both entry points
    make_m, nwords_m, keep_m = _sparse_batch((24, 24), leaf=4, with_map=True)
    ref_m, got_m = _both(L, make_m, torch.zeros(nwords_m, dtype=torch.float64, device=dev))
    assert np.array_equal(ref_m, ref) and np.array_equal(got_m, ref)

    # what the entry point is named for: with the device set up (gpuqr_init, and the calls above), a
    # call allocates nothing: no host allocation through the library's hooks, no device memory
    MALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t)
    CALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_size_t)
    FREE = C.CFUNCTYPE(None, C.c_void_p)
    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = C.c_void_p, [C.c_size_t]
    libc.calloc.restype, libc.calloc.argtypes = C.c_void_p, [C.c_size_t, C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    calls = {"malloc": 0, "calloc": 0}

    def counting_malloc(sz):
        calls["malloc"] += 1
        return libc.malloc(sz)

    def counting_calloc(a, b):
        calls["calloc"] += 1
        return libc.calloc(a, b)

    cbs = (MALLOC(counting_malloc), CALLOC(counting_calloc), FREE(lambda p: libc.free(p)))
    L.gpuqr_set_allocator.argtypes = [MALLOC, CALLOC, FREE]
    buf = torch.zeros(nwords_m, dtype=torch.float64, device=dev)
    keep_n = []
    arr = engine._pack(make_m(buf), keep_n)
    st = _lib.Stats()
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    try:
        L.gpuqr_set_allocator(*cbs)
        code = L.gpuqr_factor_sparse_noalloc(arr, arr._length_, C.byref(st))
    finally:
        L.gpuqr_set_allocator(MALLOC(0), CALLOC(0), FREE(0))
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert code == _lib.GPUQR_SUCCESS
    assert calls == {"malloc": 0, "calloc": 0}, calls
    assert free_after == free_before
    assert np.array_equal(buf.cpu().numpy(), ref)
    # (the allocating entry point under the same hooks, on a batch it has not seen, does call them: the
    # hooks are live)
    make_a, flat_a = _dense_batch([rng.standard_normal((130, 50)), rng.standard_normal((77, 33))])
    keep_a = []
    arr2 = engine._pack(make_a(flat_a.clone()), keep_a)
    try:
        L.gpuqr_set_allocator(*cbs)
        assert L.gpuqr_factor_sparse(0, arr2, arr2._length_, None, None, None, C.byref(st)) == _lib.GPUQR_SUCCESS
    finally:
        L.gpuqr_set_allocator(MALLOC(0), CALLOC(0), FREE(0))
    torch.cuda.synchronize()
    assert calls["malloc"] + calls["calloc"] > 0

    # beyond the fixed workspace: refused before anything is launched, the front is untouched
    m, n = 16384, 4096
    big = torch.ones(m * n, dtype=torch.float64, device=dev)
    fr = ss.Front(0, ss.EMPTY, m, n)
    fr.gpuF = big.data_ptr()
    arr = engine._pack([fr], [])
    assert L.gpuqr_factor_sparse_noalloc(arr, 1, None) == 3          # GPUQR_TOOLARGE
    assert L.gpuqr_noalloc_last_need() > L.gpuqr_noalloc_workspace_bytes()
    torch.cuda.synchronize()
    assert bool((big == 1).all())
