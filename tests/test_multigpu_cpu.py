"""world_size-2 (and 3) gloo runs of the subtree-parallel sparse QR: partition, exchange of
contribution blocks and assembly of R are the product's; the per-rank numeric work on CPU is the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from suitesparse_amd import multigpu, problems, spqr
from tests.util import r_rel_diff, rownorm_sign

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shape,leaf", [((24, 24), 3), ((12, 12, 12), 3)])
def test_mapping_covers_tree_and_never_loses_to_one_rank(shape, leaf):
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=leaf), gpuMemorySize=2**34)
    nf = sym.nf
    w = multigpu.front_weights(sym)
    for parts in (1, 2, 3, 4, 8):
        owner, level = multigpu.map_tree(sym, parts)
        assert owner.min() >= 0 and owner.max() < parts
        for f in range(nf):
            p = sym.Parent[f]
            if p < nf:
                # a call takes blocks only from calls of lower levels, or from itself
                assert level[p] >= level[f] + (1 if owner[p] != owner[f] else 0)
        rep = multigpu.mapping_report(sym, owner, level, parts)
        assert abs(sum(rep["loads"]) - 1.0) < 1e-9
        assert rep["speedup_bound"] >= 1.0 - 1e-12           # splitting never does worse than one rank
        assert rep["speedup_bound"] <= parts + 1e-9
        if parts == 1:
            assert rep["top_flops_share"] == 0.0 and level.max() == 0
        else:
            load = np.array([w[owner == r].sum() for r in range(parts)])
            assert load.min() > 0                            # nobody idles on a balanced grid
            assert rep["speedup_bound"] > 1.15               # and the split pays
            # the root's rank carries the chain of separators above its subtree, no more
            assert load.max() <= rep["top_flops_share"] * w.sum() + 1.5 * w.sum() / parts


def test_mapping_of_a_chain_stays_on_one_rank():
    """A tree with nothing to run side by side: everything lands on one rank and the bound says 1."""
    n = 40
    A = sp.diags([np.ones(n), np.ones(n - 1)], [0, 1], shape=(n, n), format="csc") + sp.eye(n, format="csc")
    sym = spqr.analyze(sp.vstack([A, sp.eye(n)]).tocsc(), None, gpuMemorySize=2**30)
    owner, level = multigpu.map_tree(sym, 4)
    rep = multigpu.mapping_report(sym, owner, level, 4)
    assert abs(rep["speedup_bound"] - 1.0) < 0.35


@pytest.mark.parametrize("world,shape,band", [(2, "14x14", 2048), (3, "7x7x7", 5), (4, "9x9x9", 2048), (2, "9x9x9", 7)])
def test_distributed_factorization_gloo(tmp_path, world, shape, band):
    """(band: rows per band of a contribution block on the wire; the small values cut the test's blocks
    into several upper-trapezoid bands, as the 2048 of production cuts a separator's block)"""
    out = str(tmp_path / "R.npz")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "mgpu_worker.py"), "gloo", out, shape]
    env = dict(os.environ, OMP_NUM_THREADS="1", GPUQR_CUT_BAND=str(band))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    z = np.load(out)
    R = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"])).toarray()
    assert int(z["n_cut"]) >= world - 1 and int(z["levels"]) >= 2
    dims = tuple(int(x) for x in shape.split("x"))
    A = problems.stacked_ls(dims)
    Rl = np.linalg.qr(A[:, z["Qfill"]].toarray(), mode="r")
    assert r_rel_diff(rownorm_sign(R), rownorm_sign(Rl)) <= 1e-12


@pytest.mark.parametrize("shape,leaf,relax", [((24, 24), 4, 20), ((10, 10, 10), 3, -1), ((16, 16, 16), 4, -1), ((40, 9), 3, 0)])
@pytest.mark.parametrize("parts", [1, 2, 3, 4, 8])
def test_library_mapping_is_the_python_mapping(shape, leaf, relax, parts):
    """spqr_amd_map_tree (what the one-process multi-GPU entry point uses) against multigpu.map_tree and
    mapping_report: same owners, same levels, same bounds."""
    from suitesparse_amd import multigpu, problems, spqr
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=leaf), relax=relax, gpuMemorySize=2**33)
    owner, level = multigpu.map_tree(sym, parts)
    rep = multigpu.mapping_report(sym, owner, level, parts)
    o2, l2, top, bound = multigpu.map_tree_c(sym, parts)
    assert np.array_equal(owner, o2) and np.array_equal(level, l2)
    assert abs(top - rep["top_flops_share"]) <= 1e-12 and abs(bound - rep["speedup_bound"]) <= 1e-9 * bound
