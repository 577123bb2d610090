"""Full-size parity of the dense path through size-independent properties (the oracle is too slow
here): for every shape of the benchmark sweep, R'R = A'A to 1e-12 (the bar of
gpuqrengine_demo.cpp:98-100), R upper trapezoidal inside each panel block, R_kk >= 0, and R equal
to LAPACK's R (torch.linalg.qr on the device) up to row signs for the tall shapes."""
import numpy as np
import pytest

import suitesparse_amd as ss

pytestmark = pytest.mark.gpu

FM = [96, 192, 384, 768, 1536, 3072, 4096]
FN = [32, 64, 128, 256, 512, 1024, 2048]


def _run(shapes, seed):
    import torch
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    total = sum(m * n for m, n in shapes)
    src = torch.rand(total, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    buf = src.clone()
    fronts, off = [], 0
    for k, (m, n) in enumerate(shapes):
        f = ss.Front(k, ss.EMPTY, m, n)
        f.gpuF = buf.data_ptr() + off * 8
        off += m * n
        fronts.append(f)
    st = ss.QREngineStats()
    assert ss.GPUQREngine(0, fronts, len(fronts), stats=st) == ss.QRENGINE_SUCCESS
    torch.cuda.synchronize()
    out, off = [], 0
    for m, n in shapes:
        out.append((src[off:off + m * n].view(m, n), buf[off:off + m * n].view(m, n)))
        off += m * n
    return out, st


def test_every_benchmark_shape_gram_identity_and_lapack_r():
    import torch
    shapes = [(m, n) for m in FM for n in FN]
    res, st = _run(shapes, 7)
    assert st.flopsActual > 0
    for (m, n), (A, F) in zip(shapes, res):
        k = min(m, n)
        R = torch.triu(F[:k, :])
        G = A.T @ A
        err = (R.T @ R - G).abs().sum(0).max() / G.abs().sum(0).max()       # 1-norms, as the demo
        assert err <= 1e-12, (m, n, float(err))
        d = torch.diagonal(R)
        # a column with nothing below the diagonal keeps its sign (sigma = 0, factorize_vt.cu:272-279):
        # the last row of a front with m <= n
        assert bool((d[:k - 1] >= 0).all()) and (m <= n or float(d[k - 1]) >= 0)
        for c0 in range(0, k, 32):                                           # clean diagonal blocks
            blk = F[c0:c0 + 32, c0:c0 + 32]
            assert float(torch.tril(blk, -1).abs().max()) == 0.0 if blk.shape[0] > 1 else True
        if m >= n:
            Rl = torch.linalg.qr(A, mode="r").R
            sgn = torch.sign(torch.diagonal(Rl))
            sgn[sgn == 0] = 1
            Rl = Rl * sgn[:, None]
            assert float((R[:n] - Rl).abs().max() / Rl.abs().max()) <= 1e-11, (m, n)


def test_factorizing_r_again_returns_r():
    """QR of an upper triangular R with non-negative diagonal is R itself (idempotence)."""
    import torch
    shapes = [(4096, 512), (1536, 1024), (768, 768)]
    res, _ = _run(shapes, 9)
    dev = res[0][1].device
    Rs = [torch.triu(F[:min(m, n), :]).contiguous() for (m, n), (_, F) in zip(shapes, res)]
    total = sum(R.numel() for R in Rs)
    buf = torch.empty(total, dtype=torch.float64, device=dev)
    fronts, off = [], 0
    for k, R in enumerate(Rs):
        buf[off:off + R.numel()] = R.reshape(-1)
        f = ss.Front(k, ss.EMPTY, R.shape[0], R.shape[1])
        f.gpuF = buf.data_ptr() + off * 8
        off += R.numel()
        fronts.append(f)
    assert ss.GPUQREngine(0, fronts, len(fronts)) == ss.QRENGINE_SUCCESS
    torch.cuda.synchronize()
    off = 0
    for R in Rs:
        R2 = torch.triu(buf[off:off + R.numel()].view(R.shape))
        off += R.numel()
        assert float((R2 - R).abs().max() / R.abs().max()) <= 1e-13


@pytest.mark.parametrize("grid,leaf", [((256, 256), 8), ((32, 32, 32), 8), ((1024, 1024), 8), ((64, 64, 64), 8)])
def test_sparse_least_squares_residual_at_size(grid, leaf):
    """The checks of qrdemo_gpu.cpp:92-113 at a size the dense reference cannot follow: the solution
    from R and Q'b satisfies the normal equations, ||A'(Ax-b)|| / (||A|| ||b||) at rounding level, and
    rank = n."""
    from suitesparse_amd import problems, spqr
    A = problems.stacked_ls(grid)
    m, n = A.shape
    b = np.ones(m)
    x, R, sym = spqr.qr_solve(A, b, problems.nested_dissection(grid, leaf=leaf), relax=20)
    x = x[:, 0]
    r = A @ x - b
    anorm = abs(A).sum(axis=0).max()
    assert np.linalg.norm(A.T @ r) <= 1e-12 * anorm * np.linalg.norm(b) * 10
    assert R.shape[0] == n + 1 and np.all(np.abs(R.diagonal()[:n]) > 0)



def _bench_order(A, shape):
    """The column order bench.py uses: the reference's METIS order when its build is here."""
    from oracle import ref
    from suitesparse_amd import problems
    if ref.available():
        return ref.metis_order_child(A)
    return problems.nested_dissection(shape, leaf=8)


@pytest.mark.parametrize("shape,mem_gib", [((1024, 1024), 200), ((96, 96, 96), 200), ((128, 128, 128), 200)],
                         ids=["config3-1024x1024", "3d-96", "config4-128-staged"])
def test_r_at_the_sizes_the_bench_times(shape, mem_gib):
    """What bench.py times, checked at the size it is timed: the 2D 1024^2 grid (config 3), the 3D 96^3
    grid and the 3D 128^3 grid (config 4: four stages at 200 GiB declared, the 70 GB of R on a pageable
    stack), in the bench's column order and with SPQR's own fronts.  The check is the reference's
    check_r_factor (SPQR/Tcov/qrtest.cpp:1218-1257: R'R against (AE)'(AE), bar 1e-10) through a random
    vector, then the checks of qrdemo_gpu.cpp:92-113 with the right-hand side of the demo (b = ones)
    solved by the seminormal equations with one step of refinement: rank = n and
    ||A'(Ax - b)|| / (||A|| ||b||) at rounding level."""
    import scipy.sparse as sp
    from suitesparse_amd import multigpu, problems, spqr
    A = problems.stacked_ls(shape)
    m, n = A.shape
    Q = _bench_order(A, shape)
    sym = spqr.analyze(A, Q, relax=-1, gpuMemorySize=mem_gib * 2**30)
    res = spqr.factorize(sym, multigpu._Csc(A), want_R=False, keep_num=True)
    try:
        assert res.rank == n
        err = spqr.gram_probe(sym, res, A, seed=3)
        assert err <= 1e-12, err
        # min ||Ax - b||: R'R x = A'b on the packed blocks (two triangular solves), one refinement step
        b = np.ones(m)
        Qf = np.asarray(sym.Qfill)

        def solve(rhs):                     # (R'R) y = rhs in the factorization's column order
            w = spqr.rtsolve(sym, res, rhs)
            return spqr.rsolve(sym, res, w, permute=False)[:, 0]

        g = (A.T @ b)[Qf]
        y = solve(g)
        x = np.zeros(n)
        x[Qf] = y
        r = b - A @ x
        y += solve((A.T @ r)[Qf])
        x[Qf] = y
        r = A @ x - b
        anorm = abs(A).sum(axis=0).max()
        assert np.linalg.norm(A.T @ r) <= 1e-12 * anorm * np.linalg.norm(b) * 10
    finally:
        res.free()
