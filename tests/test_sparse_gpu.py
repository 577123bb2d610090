"""GPU parity of the sparse path: S-assembly, pack-assembly, staging/limbo and the numeric
driver against the CPU oracle (same symbolic structure) and LAPACK (R is unique up to row signs)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from suitesparse_amd import problems, spqr
from tests.util import r_rel_diff, rownorm_sign

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    yield "2d-12", problems.stacked_ls((12, 12)), problems.nested_dissection((12, 12), leaf=3)
    yield "2d-40", problems.stacked_ls((40, 40)), problems.nested_dissection((40, 40), leaf=4)
    yield "2d-33x17", problems.stacked_ls((33, 17)), problems.nested_dissection((33, 17), leaf=4)
    yield "3d-9", problems.stacked_ls((9, 9, 9)), problems.nested_dissection((9, 9, 9), leaf=2)
    rng = np.random.default_rng(4)
    R = sp.random(300, 120, density=0.03, random_state=4, format="csc") + sp.vstack(
        [sp.identity(120), sp.csc_matrix((180, 120))])
    yield "random", sp.csc_matrix(R), rng.permutation(120)
    yield "natural-order", problems.stacked_ls((10, 10)), None
    # sizes at which fronts have several panels, several 256-row chunks and merge levels
    yield "3d-14", problems.stacked_ls((14, 14, 14)), problems.nested_dissection((14, 14, 14), leaf=3)
    yield "2d-72", problems.stacked_ls((72, 72)), problems.nested_dissection((72, 72), leaf=6)


CASES = list(_cases())


def _dense_R(res, n):
    R = np.zeros((n, n))
    Rd = res.R.toarray()
    R[:Rd.shape[0]] = Rd
    return R


@pytest.mark.parametrize("name,A,Q", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("relax", [0, 20])
def test_sparse_qr_one_stage(name, A, Q, relax):
    sym = spqr.analyze(A, Q, relax=relax, gpuMemorySize=2**34)
    assert sym.numStages == 1
    res = spqr.factorize(sym, A)
    n = sym.n
    assert res.rank == n and res.numStagesRun == 1
    R = _dense_R(res, n)
    # vs LAPACK on the permuted dense matrix
    Rl = np.linalg.qr(A[:, sym.Qfill].toarray(), mode="r")
    assert r_rel_diff(rownorm_sign(R), rownorm_sign(Rl)) <= 1e-12
    # vs the CPU oracle: same fronts, same maps
    Farena, FOff = oracle.multifrontal(sym, spqr.s_values(sym, A))
    Ro = oracle.r_from_fronts(sym, Farena, FOff)
    assert r_rel_diff(rownorm_sign(R), rownorm_sign(Ro)) <= 1e-12
    # the pattern of R is the symbolic one: nnz == rnz and nothing outside the front patterns
    assert res.R.nnz == sym.rnz
    # ||R'R - (AQ)'(AQ)||_1 / ||A'A||_1 (check_r_factor, SPQR/Tcov/qrtest.cpp:1218-1257)
    AQ = A[:, sym.Qfill].toarray()
    G = AQ.T @ AQ
    assert np.linalg.norm(R.T @ R - G, 1) / np.linalg.norm(G, 1) <= 1e-12


@pytest.mark.parametrize("shape,leaf", [((512, 512), 8), ((56, 56, 56), 8)])
def test_large_grids_by_size_independent_properties(shape, leaf):
    """Sizes at which no dense comparison is possible (fronts of thousands of columns, many rounds of
    row-block hand-overs, R of 10^7..10^8 entries): R'R y = (AQ)'(AQ) y for random y, which pins every
    entry of R'R without forming it; ||R y|| = ||A Q y||; R upper triangular with the symbolic pattern;
    the same R from a second factorization that replays the schedule."""
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=leaf), relax=20, gpuMemorySize=2**36)
    assert sym.numStages == 1
    res = spqr.factorize(sym, A)
    n = sym.n
    assert res.rank == n
    R = res.R.tocsr()
    assert R.shape == (n, n) and R.nnz == sym.rnz
    coo = R.tocoo()
    assert (coo.row <= coo.col).all()
    assert (R.diagonal()[:-1] > 0).all()
    AQ = A[:, sym.Qfill[:n]].tocsr()
    rng = np.random.default_rng(9)
    for _ in range(3):
        y = rng.standard_normal(n)
        Ry, Ay = R @ y, AQ @ y
        assert abs(np.linalg.norm(Ry) - np.linalg.norm(Ay)) <= 1e-12 * np.linalg.norm(Ay)
        u, v = R.T @ Ry, AQ.T @ Ay
        assert np.linalg.norm(u - v) <= 1e-12 * np.linalg.norm(v)
    again = spqr.factorize(sym, A)
    assert (again.R != res.R).nnz == 0


@pytest.mark.parametrize("name,A,Q", CASES[:4], ids=[c[0] for c in CASES[:4]])
@pytest.mark.parametrize("mem", [0, 10_700_000, 11_200_000])
@pytest.mark.parametrize("host_limbo", [False, True])
def test_sparse_qr_staged_equals_one_stage(name, A, Q, mem, host_limbo, monkeypatch):
    """Shrinking the declared device memory forces several stages and the "limbo" wait of
    contribution blocks between them (the reference tests staging the same way,
    SPQR/Tcov/qrtest.cpp:2660-2684): on the host as in the reference, or on the device when it has
    the room.  The factor must not depend on the staging."""
    if host_limbo:
        monkeypatch.setenv("GPUQR_HOST_LIMBO", "1")
    one = spqr.factorize(spqr.analyze(A, Q, gpuMemorySize=2**34), A)
    sym = spqr.analyze(A, Q, gpuMemorySize=mem)
    if sym.status == 2:
        pytest.skip("infeasible at this size")
    res = spqr.factorize(sym, A)
    if mem == 0:
        assert res.numStagesRun == sym.nf + 1
    assert res.rank == one.rank
    assert (res.R != one.R).nnz == 0 or np.abs((res.R - one.R)).max() <= 1e-13 * np.abs(one.R).max()


def test_infeasible_memory_is_reported():
    A = problems.stacked_ls((12, 12))
    sym = spqr.analyze(A, None, gpuMemorySize=1)
    assert sym.status == 2 and sym.numStages == 0
    with pytest.raises(RuntimeError):
        spqr.factorize(sym, A)


def test_rank_deficient_hits_the_wall():
    """Structurally rank-deficient input: pivot columns past the bottom of a front are dead
    (spqrgpu_kernel.cpp:763-771); rank = sum of front ranks."""
    A = sp.csc_matrix(np.array([[1., 2, 0, 0], [0, 0, 0, 3.], [0, 0, 0, 0]]))
    sym = spqr.analyze(A, None, gpuMemorySize=2**30)
    res = spqr.factorize(sym, A)
    assert res.rank < 4 and res.rank == int(sum(min(sym.Fm[f], sym.Super[f + 1] - sym.Super[f]) for f in range(sym.nf)))
    assert res.Rdead.sum() == 4 - res.rank


@pytest.mark.parametrize("mem", [0, 14_000_000, 2**34])
@pytest.mark.parametrize("dev_blocks", [False, True])
def test_subset_driver_in_groups_and_pieces(mem, dev_blocks):
    """The subset driver on one process: the tree cut into three pieces (two subtrees, then the rest)
    that are factorized by three calls, blocks handed from call to call on the host or on the device;
    with little declared memory every call runs in several groups.  Same R as the whole-tree driver."""
    import torch
    from suitesparse_amd import multigpu
    shape = (9, 9, 9)
    A = problems.stacked_ls(shape)
    Q = problems.nested_dissection(shape, leaf=2)
    one = spqr.factorize(spqr.analyze(A, Q, gpuMemorySize=2**34), A)
    sym = spqr.analyze(A, Q, gpuMemorySize=mem)
    nf = sym.nf
    owner, level = multigpu.map_tree(sym, 3)
    pieces = [(owner == r) & (level == L) for L in range(int(level.max()) + 1) for r in range(3)]
    pieces = [p for p in pieces if p.any()]
    assert len(pieces) >= 3
    held, Rall, groups = {}, {}, 0
    for mine in pieces:
        limbo = {c: held[c] for f in np.nonzero(mine)[0] for c in multigpu._children(sym, int(f)) if not mine[c]}
        if dev_blocks:
            Rb, Cb, st = multigpu.gpu_factor_subset_dev(sym, A, mine, limbo)
            Cb = {f: t.clone() for f, t in Cb.items()}
        else:
            Rb, Cb, st = multigpu.gpu_factor_subset(sym, A, mine, limbo)
        groups += st["groups"]
        held.update(Cb)
        Rall.update(Rb)
    torch.cuda.synchronize()
    assert len(Rall) == nf
    if mem == 0:
        assert groups == nf                      # every front a group of its own
    elif mem < 2**30:
        assert len(pieces) < groups < nf         # the declared memory forced several groups per call
    else:
        assert groups == len(pieces)
    R = multigpu.blocks_to_csc(sym, Rall)
    assert R.shape == one.R.shape and R.nnz == one.R.nnz
    assert np.abs((R - one.R)).max() <= 1e-13 * np.abs(one.R).max()


@pytest.mark.parametrize("grid,leaf", [((24, 24), 4), ((8, 8, 8), 2)])
def test_least_squares_through_appended_rhs(grid, leaf):
    """x = R \\ (Q'b) with Q'b obtained by appending b to A (SuiteSparseQR.cpp:256-260), checked
    against the dense least-squares solution and the normal-equations residual of qrdemo_gpu.cpp:92-113."""
    A = problems.stacked_ls(grid)
    m, n = A.shape
    rng = np.random.default_rng(12)
    B = np.column_stack([np.ones(m), rng.standard_normal(m)])
    x, R, sym = spqr.qr_solve(A, B, problems.nested_dissection(grid, leaf=leaf), relax=20)
    xref = np.linalg.lstsq(A.toarray(), B, rcond=None)[0]
    assert np.abs(x - xref).max() <= 1e-11 * np.abs(xref).max()
    r = A @ x - B
    assert np.abs(A.T @ r).max() <= 1e-11 * np.abs(A).max() * np.abs(B).max() * m ** 0.5


def test_pageable_r_stack_fallback(monkeypatch):
    """When the system refuses the pinned R stack the driver copies through bounce buffers into
    pageable memory: same R."""
    grid = (30, 30)
    A = problems.stacked_ls(grid)
    sym = spqr.analyze(A, problems.nested_dissection(grid, leaf=4), relax=20, gpuMemorySize=2**34)
    r0 = spqr.factorize(sym, A)
    monkeypatch.setenv("GPUQR_PAGEABLE_R", "1")
    r1 = spqr.factorize(sym, A)
    assert (r0.R != r1.R).nnz == 0


def test_pageable_r_stack_with_stages_copies_beside_the_next_stage(monkeypatch):
    """Several stages and a pageable stack: the R of a stage is copied out by a helper thread while
    the next stage runs (two device blocks take turns); same R as the pinned, one-stage run."""
    grid = (40, 40)
    A = problems.stacked_ls(grid)
    Q = problems.nested_dissection(grid, leaf=4)
    one = spqr.factorize(spqr.analyze(A, Q, relax=0, gpuMemorySize=2**34), A)
    sym = spqr.analyze(A, Q, relax=0, gpuMemorySize=10_700_000)
    assert sym.numStages > 2
    monkeypatch.setenv("GPUQR_PAGEABLE_R", "1")
    r1 = spqr.factorize(sym, A)
    monkeypatch.setenv("GPUQR_SERIAL_R", "1")
    r2 = spqr.factorize(sym, A)
    assert r1.numStagesRun == sym.numStages
    assert (r1.R != r2.R).nnz == 0
    assert (r1.R != one.R).nnz == 0 or np.abs((r1.R - one.R)).max() <= 1e-13 * np.abs(one.R).max()


@pytest.mark.parametrize("grid,leaf,mem", [((48, 48), 4, 2**34), ((12, 12, 12), 2, 2**34), ((48, 48), 4, 20_000_000)])
def test_contribution_blocks_sent_row_block_by_row_block_change_nothing(monkeypatch, grid, leaf, mem):
    """With the host copy of the row map the engine sends a child's rows to the parent as they become
    final and starts the parent's panels early.  Same arithmetic in the same order: R must be
    bit-identical to the factorization where every parent waits for its whole children, in fewer
    launches (more tasks share a round)."""
    A = problems.stacked_ls(grid)
    Q = problems.nested_dissection(grid, leaf=leaf)
    # (a schedule is kept per symbolic object: two objects, two schedules)
    monkeypatch.setenv("GPUQR_NO_PIPELINE", "1")
    whole = spqr.factorize(spqr.analyze(A, Q, relax=20, gpuMemorySize=mem), A)
    monkeypatch.delenv("GPUQR_NO_PIPELINE")
    early = spqr.factorize(spqr.analyze(A, Q, relax=20, gpuMemorySize=mem), A)
    assert early.rank == whole.rank == A.shape[1]
    assert (early.R != whole.R).nnz == 0
    assert early.gpuNumKernelLaunches < whole.gpuNumKernelLaunches


@pytest.mark.parametrize("grid,leaf,mem,mode", [((30, 30), 4, 2**34, 0), ((10, 10, 10), 2, 2**34, 0),
                                                ((30, 30), 4, 2**22, 0), ((30, 30), 4, 2**34, 1)])
def test_r_packed_inside_the_engine_equals_r_packed_afterwards(monkeypatch, grid, leaf, mem, mode):
    """R blocks packed and sent by the engine while later fronts still run (page-locked stack) are
    bit for bit the blocks packed in one pass after the engine returned; also when the tree is cut
    into stages and when the work-queue kernel runs the schedule."""
    from suitesparse_amd import _lib
    A = problems.stacked_ls(grid)
    sym = spqr.analyze(A, problems.nested_dissection(grid, leaf=leaf), relax=20, gpuMemorySize=mem)
    _lib.lib().gpuqr_set_mode(mode)
    try:
        r0 = spqr.factorize(sym, A)
        r0b = spqr.factorize(sym, A)
        monkeypatch.setenv("GPUQR_LATE_R", "1")
        r1 = spqr.factorize(sym, A)
    finally:
        _lib.lib().gpuqr_set_mode(0)
    assert (r0.R != r1.R).nnz == 0 and (r0b.R != r1.R).nnz == 0
    assert r0.R.nnz == r1.R.nnz


def test_staged_refactorization_plans_nothing():
    """A factorization in several stages is several engine calls; from the second factorization on each
    of them must find the schedule it made the first time.  That needs the same front descriptors,
    and those of a later stage name the device blocks in which the contributions of earlier stages
    wait: the driver hands those blocks out of one buffer, first-fit, so they land at the same
    addresses every time."""
    import ctypes as C
    from suitesparse_amd import _lib
    L = _lib.lib()
    L.gpuqr_schedules_built.restype = C.c_int64
    A = problems.stacked_ls((20, 20, 20))
    Q = problems.nested_dissection((20, 20, 20), leaf=3)
    sym = spqr.analyze(A, Q, relax=-1, gpuMemorySize=80_000_000)
    assert sym.numStages >= 4
    n0 = L.gpuqr_schedules_built()
    r1 = spqr.factorize(sym, A)
    n1 = L.gpuqr_schedules_built()
    assert n1 - n0 == r1.numStagesRun >= sym.numStages
    A2 = A.copy()
    A2.data = A2.data * 1.5
    r2 = spqr.factorize(sym, A2)
    r3 = spqr.factorize(sym, A)
    assert L.gpuqr_schedules_built() == n1                  # nothing planned again
    assert (r3.R != r1.R).nnz == 0
    assert abs(abs(r2.R) - 1.5 * abs(r1.R)).max() <= 1e-12 * abs(r2.R).max()


def test_repeated_factorizations_reuse_the_schedule_and_follow_the_values(monkeypatch):
    """A second factorization on the same symbolic object replays the engine's schedule; it must
    track new values, survive other engine calls in between, and agree with the uncached path."""
    grid = (30, 30)
    A = problems.stacked_ls(grid)
    Q = problems.nested_dissection(grid, leaf=4)
    sym = spqr.analyze(A, Q, relax=20, gpuMemorySize=2**34)
    r1 = spqr.factorize(sym, A)
    A2 = A.copy()
    A2.data = A2.data * np.linspace(0.5, 1.5, A2.nnz)
    r2 = spqr.factorize(sym, A2)                                    # schedule replayed, new values
    sym_b = spqr.analyze(problems.stacked_ls((12, 12)), None, gpuMemorySize=2**34)
    spqr.factorize(sym_b, problems.stacked_ls((12, 12)))            # another problem in between
    r3 = spqr.factorize(sym, A)
    monkeypatch.setenv("GPUQR_NO_PLAN_CACHE", "1")
    r1n = spqr.factorize(sym, A)
    r2n = spqr.factorize(sym, A2)
    assert (r1.R != r1n.R).nnz == 0 and (r3.R != r1n.R).nnz == 0
    assert (r2.R != r2n.R).nnz == 0 and (r2.R != r1.R).nnz > 0


def _config1_matrix():
    """Stand-in for ash219 (219 x 85, 0/1 entries, two per row), which is not in the reference tree:
    same shape and density from a seeded generator."""
    rng = np.random.default_rng(219)
    rows = np.repeat(np.arange(219), 2)
    cols = np.concatenate([rng.choice(85, size=2, replace=False) for _ in range(219)])
    return sp.csc_matrix((np.ones(438), (rows, cols)), shape=(219, 85))


def test_config1_small_rectangular_least_squares():
    """The plumbing case of the reference's qrdemo (SPQR/Demo/qrdemo.cpp: min ||Ax - b||, b = ones)
    on the ash219 stand-in: R against LAPACK and the oracle, and the residual checks the demo prints."""
    A = _config1_matrix()
    m, n = A.shape
    assert np.linalg.matrix_rank(A.toarray()) == n
    sym = spqr.analyze(A, None, relax=0, gpuMemorySize=2**30)
    res = spqr.factorize(sym, A)
    assert res.rank == n
    R = _dense_R(res, n)
    Rl = np.linalg.qr(A[:, sym.Qfill].toarray(), mode="r")
    assert r_rel_diff(rownorm_sign(R), rownorm_sign(Rl)) <= 1e-12
    Farena, FOff = oracle.multifrontal(sym, spqr.s_values(sym, A))
    assert r_rel_diff(rownorm_sign(R), rownorm_sign(oracle.r_from_fronts(sym, Farena, FOff))) <= 1e-12
    b = np.ones((m, 1))
    x, _, _ = spqr.qr_solve(A, b, None, relax=0)
    xref = np.linalg.lstsq(A.toarray(), b, rcond=None)[0]
    assert np.abs(x - xref).max() <= 1e-10 * max(1.0, np.abs(xref).max())
    r = A @ x - b
    assert np.abs(A.T @ r).max() <= 1e-10 * np.abs(A).sum()          # ||A'(Ax-b)||, qrdemo's second check


def test_config5_dependent_columns_show_on_the_diagonal():
    """Duplicated and nearly dependent columns.  The device path has no rank detection (the reference
    turns its GPU off when a tolerance is asked for, SPQR/Source/spqr_analyze.cpp:120-125), so the
    factorization runs structurally: R'R = A'A still holds to rounding, and the dependent columns are
    exactly those whose diagonal entry falls under SPQR's default tolerance
    20 (m+n) eps max||A(:,j)|| (SPQR/Source/spqr_tol.cpp:23-24)."""
    grid = (24, 24)
    A0 = problems.stacked_ls(grid)
    n0 = A0.shape[1]
    rng = np.random.default_rng(5)
    dup = rng.choice(n0, size=12, replace=False)
    near = rng.choice(n0, size=12, replace=False)
    Anear = A0[:, near].toarray()
    Anear = Anear + 1e-14 * rng.standard_normal(Anear.shape) * (Anear != 0)
    A = sp.csc_matrix(sp.hstack([A0, A0[:, dup], sp.csc_matrix(Anear)]))
    m, n = A.shape
    Q = np.concatenate([problems.nested_dissection(grid, leaf=4), np.arange(n0, n)])
    sym = spqr.analyze(A, Q, relax=0, gpuMemorySize=2**32)
    res = spqr.factorize(sym, A)
    R = _dense_R(res, n)
    AQ = A[:, sym.Qfill].toarray()
    G = AQ.T @ AQ
    assert np.linalg.norm(R.T @ R - G, 1) / np.linalg.norm(G, 1) <= 1e-12
    tol = 20.0 * (m + n) * np.finfo(float).eps * np.sqrt((AQ * AQ).sum(axis=0)).max()
    dead = np.abs(np.diag(R)) <= tol
    assert dead.sum() == 24
    assert np.linalg.matrix_rank(AQ, tol=tol) == n - 24
    # the dead ones are the appended copies (they come last in the column order)
    assert set(np.nonzero(dead)[0]) == set(np.nonzero(sym.Qfill >= n0)[0])


@pytest.mark.parametrize("mem", [2**34, 14_000_000, 0])
@pytest.mark.parametrize("pageable", [False, True])
def test_spqr_binding_from_arrays_gives_the_same_r(mem, pageable, monkeypatch):
    """What an SPQR build does (adapter/spqrgpu_kernel.cpp): the symbolic object from QRsym's arrays,
    the values as Sx, R written onto the caller's stack.  Same bits as the direct path; same stage
    count; the stack head moves by exactly nnz of the R blocks."""
    if pageable:
        monkeypatch.setenv("GPUQR_PAGEABLE_R", "1")
    shape = (9, 9, 9)
    A = problems.stacked_ls(shape)
    Q = problems.nested_dissection(shape, leaf=2)
    sym = spqr.analyze(A, Q, relax=20, gpuMemorySize=mem)
    direct = spqr.factorize(sym, A)
    sym2 = spqr.symbolic_from_arrays(sym)
    out = spqr.spqrgpu_kernel(sym2, spqr.s_values(sym, A), stack_room=sym.rnz + 7)
    assert out["code"] == 0
    assert out["used"] == sym.rnz and out["sumfrank"] == direct.rank and out["maxfrank"] == direct.maxfrank
    assert np.array_equal(out["Rdead"][:sym.n], direct.Rdead)
    assert np.all(out["stack"][sym.rnz:] == 0)                  # nothing written past the blocks
    # blocks lie in postorder, stage after stage (spqrgpu_kernel.cpp:399-440)
    order = np.argsort(out["Rblock"][:sym.nf])
    assert np.array_equal(order, sym.Post[:sym.nf])
    from suitesparse_amd import multigpu
    Rb = {}
    for f in range(sym.nf):
        sz = multigpu._r_size(sym, f)
        Rb[f] = out["stack"][out["Rblock"][f]:out["Rblock"][f] + sz]
    R = multigpu.blocks_to_csc(sym, Rb)
    assert R.shape == direct.R.shape and (R != direct.R).nnz == 0
    # too little room on the stack is refused
    assert spqr.spqrgpu_kernel(sym2, spqr.s_values(sym, A), stack_room=sym.rnz - 1)["code"] == 1


def _rank_case(name):
    rng = np.random.default_rng(17)
    if name == "config5-24":
        A, _, _ = problems.rank_deficient_ls(24, 12, 12)
        return A, problems.nested_dissection((24, 24), leaf=4), 24
    if name == "config5-256":
        A, _, _ = problems.rank_deficient_ls(256, 64, 64)
        return A, problems.nested_dissection((256, 256), leaf=8), 128
    if name == "zero-and-duplicate-columns":
        B = sp.random(160, 60, density=0.08, random_state=3, format="lil")
        B[:60, :] += sp.identity(60)
        B[:, 7] = 0                                 # a column of zeros
        B[:, 20] = B[:, 19]                         # neighbours in the ordering
        B[:, 41] = 2.5 * B[:, 3]                    # far apart
        B[:, 59] = B[:, 58]                         # the last column
        return sp.csc_matrix(B), None, 4
    if name == "wide":
        # more columns than rows: pivot columns run out of rows, and two die before that
        B = sp.random(30, 45, density=0.3, random_state=5, format="lil")
        B[:, 4] = B[:, 2]
        B[:, 5] = -B[:, 2]
        return sp.csc_matrix(B), rng.permutation(45), None
    if name == "row-left-by-a-dead-column":
        # column 0 is negligible and column 1 lives in the one row that column 0 starts: a factorization
        # that spends that row on column 0 finds nothing left of column 1, the CPU rule leaves the row
        # to column 1; the duplicate of column 5 dies for good
        B = np.zeros((40, 12))
        B[:, 2:] = rng.uniform(-1, 1, size=(40, 10))     # (row 0 is dense: one front holds everything)
        B[0, 0], B[0, 1] = 1e-13, 3.0                    # (the tolerance is 8.5e-13 here)
        B[:, 6] = B[:, 5]
        return sp.csc_matrix(B), None, 2
    raise KeyError(name)


@pytest.mark.parametrize("name", ["config5-24", "zero-and-duplicate-columns", "wide", "config5-256",
                                  "row-left-by-a-dead-column"])
def test_rank_detection_on_the_device_matches_the_cpu_rule(name):
    """Config 5: rank-deficient least squares at SPQR's default tolerance.  The device path
    (spqrgpu_amd_factorize_tol) must give the CPU rule's answer (the oracle restates
    SPQR/Source/spqr_front.cpp:424-479): the same dead columns, the same rank, the same squeezed R
    (pattern: one row per live pivot, dead columns cut at their would-be pivot row; values to 1e-12),
    and R'R = (AQ)'(AQ) to the tolerance."""
    A, Q, expect_dead = _rank_case(name)
    A = sp.csc_matrix(A)
    m, n = A.shape
    sym = spqr.analyze(A, Q, relax=20, gpuMemorySize=2**36, rank_detection=True)
    tol = spqr.default_tol(A)
    res = spqr.factorize(sym, A, tol=tol)
    Ro, Rdead_o, rank_o, _, normE_o = oracle.multifrontal_tol(sym, spqr.s_values(sym, A), tol)
    assert res.rank == rank_o
    assert np.array_equal(res.Rdead, Rdead_o)
    # the norm of what the dead columns dropped (cc->SPQR_norm_E_fro): rounding noise for exact
    # dependencies, so it is compared on the scale of the tolerance
    assert abs(res.norm_E_fro - normE_o) <= 0.05 * tol
    if expect_dead is not None:
        assert int(res.Rdead.sum()) == expect_dead + max(0, n - m) * 0 and res.rank == n - int(res.Rdead.sum())
        assert res.numPasses >= 2
    R = res.R.tocsc()
    assert R.shape == Ro.shape == (rank_o, n)
    # same pattern (structural: explicit zeros count) ...
    R.sort_indices()
    Ro.sort_indices()
    assert np.array_equal(R.indptr, Ro.indptr) and np.array_equal(R.indices, Ro.indices)
    # ... and the same numbers up to the sign of a row
    scale = np.abs(Ro.data).max()
    assert np.abs(np.abs(R.data) - np.abs(Ro.data)).max() <= 1e-12 * scale
    # R'R = (AQ)'(AQ) up to what the dead columns dropped
    AQ = A[:, sym.Qfill]
    G = (AQ.T @ AQ)
    E = (R.T @ R - G)
    assert abs(E).sum(axis=0).max() <= 10 * tol * abs(G).sum(axis=0).max() ** 0.5 + 1e-12 * abs(G).sum(axis=0).max()


def test_randomised_sweep_of_both_numeric_paths():
    """tests/sweeps/fuzz_parity.py, a bounded run: random sparse matrices (tall, wide, empty rows and columns),
    orderings, amalgamation and declared device memory through the numeric driver against LAPACK, and
    each one again with dependent columns through the tolerance path against the CPU rule's restatement
    (rank, Rdead, pattern, R'R on the live rows; decisions that rounding alone can tip are counted, not
    failed).  Seeds 5130 and 20557 are the two that the sweep first caught: a run of dead columns behind
    the last independent row of a wide front (two more found per pass, passes exhausted), and guesses
    that chased each other (a column left dead that the rule keeps)."""
    fz = _load_script("fuzz_parity")
    fz.TOL_EVERY = 1
    bad = []
    for seed in [5130, 20557, 20848, 21067] + list(range(70000, 70150)):
        bad += fz.one_case(seed)
    assert not bad, bad[:5]
    assert fz.COUNT["tol"] >= 140 and fz.COUNT["tol_dead"] >= 100 and fz.COUNT["max_passes"] <= 12, fz.COUNT


def _load_script(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", "sweeps", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_randomised_sweep_of_the_other_paths():
    """tests/sweeps/fuzz_paths.py, a bounded run: staged runs bit-identical to the one-stage R, the tree in
    pieces for 2-4 ranks (device or host blocks), Q'B and solves from a kept factorization, solves with
    appended right-hand sides; tests/sweeps/fuzz_dense.py: dense batches against LAPACK, twice bit for bit,
    and through the work-queue kernel."""
    fz = _load_script("fuzz_paths")
    bad = []
    for seed in range(80000, 80060):
        bad += fz.one_case(seed)
    assert not bad, bad[:5]
    assert fz.COUNT["staged"] >= 100 and fz.COUNT["pieces"] == 60 and fz.COUNT["kept"] >= 50, fz.COUNT
    fd = _load_script("fuzz_dense")
    for seed in range(80000, 80030):
        bad += fd.one_case(seed)
    assert not bad, bad[:5]


def test_rank_detection_needs_its_analysis_and_costs_nothing_at_full_rank():
    A = problems.stacked_ls((20, 20))
    Q = problems.nested_dissection((20, 20), leaf=4)
    plain = spqr.analyze(A, Q, gpuMemorySize=2**34)
    with pytest.raises(RuntimeError):
        spqr.factorize(plain, A, tol=1e-10)              # fronts not sized for dying columns: refused (3)
    sym = spqr.analyze(A, Q, gpuMemorySize=2**34, rank_detection=True)
    res = spqr.factorize(sym, A, tol=spqr.default_tol(A))
    ref = spqr.factorize(plain, A)
    assert res.numPasses == 1 and res.rank == sym.n and res.Rdead.sum() == 0
    assert np.abs(res.R - ref.R).max() <= 1e-13 * np.abs(ref.R).max()
    assert (spqr.factorize(sym, A, tol=-1.0).R != spqr.factorize(sym, A).R).nnz == 0


@pytest.mark.parametrize("shape,leaf,nrhs", [((24, 24), 4, 3), ((9, 9, 9), 2, 1), ((14, 14, 14), 3, 17), ((40, 40), 4, 32)])
def test_h_kept_on_the_device_qtmult_and_solve(shape, leaf, nrhs):
    """Factor once, keep H on the device, then push right-hand sides through it that were not known at
    factorization time: C = (Q'B)(0:n,:) against a dense Q (row signs fixed by R's diagonal), and
    x = R \\ C against the least-squares solution; twice, with different B, on the same object; and
    the same R as the ordinary factorization."""
    A = problems.stacked_ls(shape)
    Q = problems.nested_dissection(shape, leaf=leaf)
    sym = spqr.analyze(A, Q, relax=20, gpuMemorySize=2**34)
    m, n = A.shape
    kept = spqr.KeptFactorization(sym, A, nrhs=nrhs)
    plain = spqr.factorize(sym, A)
    assert kept.rank == n
    assert np.abs(kept.R - plain.R).max() <= 1e-13 * np.abs(plain.R).max()
    AQ = A[:, sym.Qfill].toarray()
    Qd, Rd = np.linalg.qr(AQ, mode="reduced")
    sgn = np.where(np.diag(Rd) < 0, -1.0, 1.0)          # the device's R has a non-negative diagonal
    rng = np.random.default_rng(3)
    for trial in range(2):
        k = nrhs if trial == 0 else max(1, nrhs // 2)
        B = rng.standard_normal((m, k))
        Cq = kept.qtmult(B)
        ref = (Qd * sgn).T @ B
        assert np.abs(Cq - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1.0)
        x = kept.solve(B)
        xl = np.linalg.lstsq(A.toarray(), B, rcond=None)[0]
        assert np.abs(x - xl).max() <= 1e-10 * max(np.abs(xl).max(), 1.0)
    kept.close()


def test_basic_solution_of_a_rank_deficient_problem_on_the_packed_blocks():
    """spqr_amd_rsolve on a squeezed R (SPQR/Source/spqr_rsolve.cpp:198-221: dead pivot columns get
    zeros, the live ones are solved for): [A b] factorized with the default tolerance, b appended as the
    reference's GPU path does; the live part of x must be the least-squares solution on the live
    columns of A."""
    A, Q, _ = _rank_case("zero-and-duplicate-columns")
    A = sp.csc_matrix(A)
    m, n = A.shape
    b = np.random.default_rng(11).standard_normal((m, 2))
    AB, Qab = spqr.append_rhs(A, b, Q)
    sym = spqr.analyze(AB, Qab, relax=20, gpuMemorySize=2**34, rank_detection=True)
    assert np.array_equal(np.sort(sym.Qfill[:n]), np.arange(n))
    res = spqr.factorize(sym, AB, tol=spqr.default_tol(A), keep_num=True)
    x = spqr.rsolve(sym, res, res.R.tocsc()[:, n:].toarray(), nacols=n)
    dead_cols = sym.Qfill[:n][res.Rdead[:n] != 0]
    res.free()
    assert len(dead_cols) == 4 and np.all(x[dead_cols] == 0)
    live = np.setdiff1d(np.arange(n), dead_cols)
    xl = np.linalg.lstsq(A[:, live].toarray(), b, rcond=None)[0]
    assert np.abs(x[live] - xl).max() <= 1e-10 * max(np.abs(xl).max(), 1.0)


@pytest.mark.parametrize("shape,leaf", [((14, 14), 3), ((6, 6, 6), 2)])
def test_residual_of_the_factorization_itself_through_kept_h(shape, leaf):
    """||A E - Q R|| / ||A|| with the device's own Q: the kept Householder vectors applied to the columns
    of A E must reproduce R (Q'(A E) = [R; 0], of which the entry point returns the first n rows), so the
    residual is formed without any other factorization in the loop.  Bar: 1e-12 in the Frobenius norm."""
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=leaf), relax=20, gpuMemorySize=2**34)
    m, n = A.shape
    assert n <= 256
    kept = spqr.KeptFactorization(sym, A, nrhs=n)
    AE = A[:, sym.Qfill[:n]].toarray()
    QtAE = kept.qtmult(AE)                   # rows 0..n-1 of Q'(A E)
    R = kept.R.toarray()
    assert np.linalg.norm(QtAE - R) <= 1e-12 * np.linalg.norm(AE)
    # and Q is orthogonal on these columns: the rows of Q'(A E) below n carry nothing, i.e. the norm is kept
    assert abs(np.linalg.norm(QtAE) - np.linalg.norm(AE)) <= 1e-12 * np.linalg.norm(AE)
    kept.close()


def test_h_kept_refuses_what_it_cannot_serve():
    A = problems.stacked_ls((12, 12))
    Q = problems.nested_dissection((12, 12), leaf=3)
    staged = spqr.analyze(A, Q, gpuMemorySize=0)              # one front per stage: the fronts are not all resident
    with pytest.raises(RuntimeError):
        spqr.KeptFactorization(staged, A, nrhs=1)
    rd = spqr.analyze(A, Q, gpuMemorySize=2**33, rank_detection=True)
    with pytest.raises(RuntimeError):
        spqr.KeptFactorization(rd, A, nrhs=1)
    with pytest.raises(RuntimeError):
        spqr.KeptFactorization(spqr.analyze(A, Q, gpuMemorySize=2**33), A, nrhs=1000)


@pytest.mark.parametrize("shape,leaf,parts", [((20, 20), 3, 2), ((10, 10, 10), 3, 3), ((16, 16, 16), 4, 4), ((24, 24, 24), 4, 8)])
def test_one_process_multi_device_entry_point_gives_the_single_device_r(shape, leaf, parts):
    """spqrgpu_amd_factorize_multi (C, one process): the tree mapped onto `parts` devices by the library's
    own spqr_amd_map_tree, one subset call per (part, level), blocks handed over on the device.  With one
    GPU on the box every part names device 0 (the parts take turns on it); R must be the single-device
    R bit for bit, with SPQR's own fronts."""
    from suitesparse_amd import multigpu
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=leaf), relax=-1, gpuMemorySize=2**34)
    res0 = spqr.factorize(sym, A)
    owner, level, top, bound = multigpu.map_tree_c(sym, parts)
    res = multigpu.factorize_multi(sym, A, parts, devices=[0] * parts)
    assert res.rank == res0.rank == sym.n
    assert (res.R != res0.R).nnz == 0
    # once more on the same symbolic object (its per-device copies, maps and schedules are kept with it)
    res_again = multigpu.factorize_multi(sym, A, parts, devices=[0] * parts)
    assert (res_again.R != res0.R).nnz == 0
    # the same with the values in S order and no A at all: what the SPQR adapter's cc->numGPU branch passes
    res_sx = multigpu.factorize_multi(sym, None, parts, devices=[0] * parts, Sx=spqr.s_values(sym, A))
    assert (res_sx.R != res0.R).nnz == 0
    if parts > 1 and len(set(owner.tolist())) > 1:
        assert res.numStagesRun >= len(set(zip(owner.tolist(), level.tolist())))
