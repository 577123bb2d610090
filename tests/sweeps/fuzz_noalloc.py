"""Randomised sweep of gpuqr_factor_sparse_noalloc against gpuqr_factor_sparse through the C ABI: front
batches of multifrontal grid problems of random shape (with and without the host row map) and random
dense batches, one after another in one process so that every call finds what the one before left in
the fixed workspace; the reduced fronts must be equal bit for bit, and a batch beyond the workspace must
be refused with GPUQR_TOOLARGE before anything runs.  Exit code 1 on failure.

    python tests/sweeps/fuzz_noalloc.py [cases] [first_seed]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from suitesparse_amd import _lib, engine                                  # noqa: E402
from tests.test_zz_no_alloc import _bind, _dense_batch, _sparse_batch     # noqa: E402

COUNT = {"sparse": 0, "dense": 0, "refused": 0}


def one_case(seed, L):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    if rng.random() < 0.6:
        shape = (int(rng.integers(3, 60)), int(rng.integers(3, 60))) if rng.random() < 0.6 else tuple(int(x) for x in rng.integers(2, 13, 3))
        make, nwords, keep = _sparse_batch(shape, leaf=int(rng.integers(2, 7)), with_map=bool(rng.random() < 0.5))
        init = torch.zeros(max(nwords, 1), dtype=torch.float64, device=dev)
        what, kind = "grid %s" % (shape,), "sparse"
    else:
        mats = [rng.standard_normal((int(rng.integers(1, 700)), int(rng.integers(1, 400)))) for _ in range(int(rng.integers(1, 6)))]
        make, init = _dense_batch(mats)
        what, kind = "dense %s" % [a.shape for a in mats], "dense"
    outs = []
    for noalloc in (False, True):
        buf = init.clone()
        torch.cuda.synchronize()
        keepalive = []
        fronts = make(buf)
        arr = engine._pack(fronts, keepalive)
        st = _lib.Stats()
        if noalloc:
            code = L.gpuqr_factor_sparse_noalloc(arr, len(fronts), C.byref(st))
            if code == 3:                       # GPUQR_TOOLARGE: must be what the sizes say
                COUNT["refused"] += 1
                if L.gpuqr_noalloc_last_need() <= L.gpuqr_noalloc_workspace_bytes():
                    return ["seed %d: %s refused although it needs %d bytes" % (seed, what, L.gpuqr_noalloc_last_need())]
                return []
        else:
            code = L.gpuqr_factor_sparse(0, arr, len(fronts), None, None, None, C.byref(st))
        if code != _lib.GPUQR_SUCCESS:
            return ["seed %d: %s: code %d (noalloc %s)" % (seed, what, code, noalloc)]
        torch.cuda.synchronize()
        outs.append(buf.cpu().numpy())
    COUNT[kind] += 1
    if not np.array_equal(outs[0], outs[1]):
        return ["seed %d: %s: the two entry points differ by %.2e" % (seed, what, np.abs(outs[0] - outs[1]).max())]
    return []


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    L = _bind()
    assert L.gpuqr_init() == _lib.GPUQR_SUCCESS
    t0, bad = time.time(), []
    for s in range(first, first + cases):
        f = one_case(s, L)
        for line in f:
            print("FAIL", line, flush=True)
        bad += f
    print("%d cases from seed %d: %d failures in %.0f s; checked: %s" % (cases, first, len(bad), time.time() - t0, COUNT))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
