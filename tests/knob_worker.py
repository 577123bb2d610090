"""Worker of test_schedule_switches_in_a_fresh_process: the library reads its schedule switches
(GPUQR_LOOKAHEAD_TILES, GPUQR_NO_PIPELINE, ...) once per process, so a switch is exercised by a
process of its own.  Factorizes a dense batch and one sparse problem and stores the results."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import suitesparse_amd as ss
    from suitesparse_amd import problems, spqr
    rng = np.random.default_rng(5)
    shapes = [(1024, 384), (700, 900), (257, 257), (3000, 200)]
    mats = [rng.standard_normal(s) for s in shapes]
    fronts = []
    for k, A in enumerate(mats):
        m, n = A.shape
        f = ss.Front(k, ss.EMPTY, m, n)
        f.F = np.ascontiguousarray(A)
        f.cpuR = np.zeros((min(m, n), n))
        fronts.append(f)
    stats = ss.QREngineStats()
    assert ss.GPUQREngine(0, fronts, len(fronts), stats=stats) == ss.QRENGINE_SUCCESS
    out = {"dense%d" % k: np.triu(np.array(f.cpuR)) for k, f in enumerate(fronts)}
    out["dense_launches"] = np.array(stats.numLaunches)
    grid = (12, 12, 12)
    A = problems.stacked_ls(grid)
    sym = spqr.analyze(A, problems.nested_dissection(grid, leaf=2), relax=20, gpuMemorySize=2**34)
    r = spqr.factorize(sym, A)
    R = r.R.tocsc()
    out.update(sp_data=R.data, sp_indices=R.indices, sp_indptr=R.indptr, sp_launches=np.array(r.gpuNumKernelLaunches))
    return out


if __name__ == "__main__":
    np.savez(sys.argv[1], **run())
