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


def test_partition_covers_tree_and_balances():
    A = problems.stacked_ls((24, 24))
    sym = spqr.analyze(A, problems.nested_dissection((24, 24), leaf=3), gpuMemorySize=2**34)
    for parts in (1, 2, 4, 8):
        owner, top = multigpu.partition_subtrees(sym, parts)
        assert owner.min() >= 0 and owner.max() < parts
        # a front that is not top has the same owner as all its descendants
        for f in range(sym.nf):
            p = sym.Parent[f]
            if p < sym.nf and not top[p]:
                assert owner[f] == owner[p] and not top[f]
        # top fronts form an ancestor-closed set on rank 0
        for f in range(sym.nf):
            if top[f]:
                assert owner[f] == 0
                assert sym.Parent[f] == sym.nf or top[sym.Parent[f]]
        if parts > 1:
            # below the cut the ranks are balanced; the top set stays on rank 0 by design
            w = multigpu.front_weights(sym)
            load = np.array([w[owner == r].sum() for r in range(parts)])
            assert load.min() > 0
            assert load.max() <= max(1.5 * load.mean(), 1.01 * w[top].sum())


@pytest.mark.parametrize("world,shape", [(2, "14x14"), (3, "7x7x7")])
def test_distributed_factorization_gloo(tmp_path, world, shape):
    out = str(tmp_path / "R.npz")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "mgpu_worker.py"), "gloo", out, shape]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    z = np.load(out)
    R = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"])).toarray()
    assert int(z["n_cut"]) >= world and int(z["n_top"]) >= 1
    dims = tuple(int(x) for x in shape.split("x"))
    A = problems.stacked_ls(dims)
    Rl = np.linalg.qr(A[:, z["Qfill"]].toarray(), mode="r")
    assert r_rel_diff(rownorm_sign(R), rownorm_sign(Rl)) <= 1e-12
