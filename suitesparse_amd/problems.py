"""Synthetic least-squares problems and fill-reducing orderings for them.

Inputs only: seeded generators of the matrices named by the benchmark
configurations (stacked grid Laplacians, rank-deficient variants) and a
geometric nested-dissection column ordering that stands in for METIS on a box
without it.  Nothing here factorizes anything.
"""
import numpy as np
import scipy.sparse as sp


def grid_laplacian(shape):
    """5-point (2D) / 7-point (3D) Dirichlet Laplacian on a regular grid, CSC."""
    dims = len(shape)
    n = int(np.prod(shape))
    idx = np.arange(n).reshape(shape)
    rows = [idx.ravel()]
    cols = [idx.ravel()]
    vals = [np.full(n, 2.0 * dims)]
    for d in range(dims):
        lo = [slice(None)] * dims
        hi = [slice(None)] * dims
        lo[d] = slice(0, shape[d] - 1)
        hi[d] = slice(1, shape[d])
        a = idx[tuple(lo)].ravel()
        b = idx[tuple(hi)].ravel()
        rows += [a, b]
        cols += [b, a]
        vals += [np.full(a.size, -1.0), np.full(a.size, -1.0)]
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    return A.tocsc()


def stacked_ls(shape, lam=0.1):
    """A = [L ; sqrt(lam) I]: 2n-by-n overdetermined least-squares matrix (CSC)."""
    L = grid_laplacian(shape)
    n = L.shape[0]
    A = sp.vstack([L, np.sqrt(lam) * sp.identity(n, format="csc")]).tocsc()
    A.sort_indices()
    return A


def rank_deficient_ls(k=256, ndup=64, nnear=64, seed=5, lam=0.1):
    """2D stacked LS at k-by-k with ndup exactly duplicated columns and nnear columns equal
    to a neighbour plus 1e-14 noise; returns (A, dup_pairs, near_pairs)."""
    rng = np.random.default_rng(seed)
    A = stacked_ls((k, k), lam).tolil()
    n = A.shape[1]
    picks = rng.choice(n - 1, size=ndup + nnear, replace=False)
    dup, near = [], []
    for t, j in enumerate(picks):
        src = j + 1
        col = A[:, src].toarray().ravel()
        if t >= ndup:
            nzr = np.nonzero(col)[0]
            col[nzr] += 1e-14 * rng.standard_normal(nzr.size)
            near.append((int(j), int(src)))
        else:
            dup.append((int(j), int(src)))
        A[:, j] = col.reshape(-1, 1)
    A = A.tocsc()
    A.sort_indices()
    return A, dup, near


def nested_dissection(shape, leaf=8, sepwidth=2):
    """Geometric nested-dissection ordering of the nodes of a regular grid.

    Separators are `sepwidth` planes wide: for a least-squares matrix whose rows
    are grid stencils, A'A couples nodes two apart, so a 2-wide separator is what
    decouples the two halves of the column elimination tree.  Returns a permutation
    (new position -> node index), separators last at every level.
    """
    dims = len(shape)
    idx = np.arange(int(np.prod(shape))).reshape(shape)
    out = []

    def rec(lo, hi):
        ext = [h - l for l, h in zip(lo, hi)]
        d = int(np.argmax(ext))
        if ext[d] <= max(leaf, sepwidth + 2):
            out.append(idx[tuple(slice(l, h) for l, h in zip(lo, hi))].ravel())
            return
        mid = lo[d] + (ext[d] - sepwidth) // 2
        l_hi = list(hi); l_hi[d] = mid
        r_lo = list(lo); r_lo[d] = mid + sepwidth
        s_lo = list(lo); s_lo[d] = mid
        s_hi = list(hi); s_hi[d] = mid + sepwidth
        rec(lo, l_hi)
        rec(r_lo, hi)
        out.append(idx[tuple(slice(l, h) for l, h in zip(s_lo, s_hi))].ravel())

    rec([0] * dims, list(shape))
    perm = np.concatenate(out).astype(np.int64)
    assert perm.size == idx.size
    return perm
