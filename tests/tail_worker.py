"""Worker of test_end_of_a_pageable_stack_page_locked (a process of its own: the page-locking limit and the
size from which pageable stacks are kept are read once per process).  A staged factorization whose R
stack is larger than the limit: the groups whose R fits under the limit at the end of the stack send
it from inside the engine, the groups before them through the bounce buffers.  Three runs that must
give the same bits: that mixed hand-off, the same again on the kept stack, and everything packed and
copied after each group (GPUQR_LATE_R=1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from suitesparse_amd import problems, spqr
    shape = tuple(int(x) for x in sys.argv[1].split("x"))
    mem = int(sys.argv[2])
    A = problems.stacked_ls(shape)
    Q = problems.nested_dissection(shape, leaf=3)
    sym = spqr.analyze(A, Q, relax=-1, gpuMemorySize=mem)
    one = spqr.analyze(A, Q, relax=-1, gpuMemorySize=2**34)
    print("stages", sym.numStages, "R bytes", 8 * int(sym.rnz), flush=True)
    assert sym.numStages >= 4
    runs = []
    for late in (False, False, True):
        if late:
            os.environ["GPUQR_LATE_R"] = "1"
        r = spqr.factorize(sym, A)
        assert r.numStagesRun >= sym.numStages
        runs.append(r.R.tocsc())
        del r
    os.environ.pop("GPUQR_LATE_R", None)
    ref = spqr.factorize(one, A).R.tocsc()
    for R in runs:
        assert R.shape == ref.shape and (R != ref).nnz == 0
    print("same bits in all runs", flush=True)


if __name__ == "__main__":
    main()
