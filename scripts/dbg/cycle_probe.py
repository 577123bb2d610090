"""Does it matter to processes started later how often this process has given its streams and device
caches back (gpuqr_finalize / spqrgpu_amd_finalize, what the test suite does after every module) and
taken them again?  CYCLES rounds of {small factorization, finalize}, then the two-rank test in this
process (its ranks are fresh child processes).  Usage: python scripts/dbg/cycle_probe.py [CYCLES]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)

import pytest  # noqa: E402
import torch  # noqa: E402,F401
from suitesparse_amd import _lib, problems, spqr  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
A = problems.stacked_ls((12, 12, 12))
Q = problems.nested_dissection((12, 12, 12), leaf=3)
L = _lib.lib()
for c in range(cycles):
    sym = spqr.analyze(A, Q, gpuMemorySize=2**33)
    r = spqr.factorize(sym, A).rank
    del sym
    L.spqrgpu_amd_finalize()
    L.gpuqr_finalize()
    torch.cuda.empty_cache()
    print("cycle %d: rank %d" % (c, r), flush=True)
rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", "--timeout", "200", "--timeout-method", "thread",
                  "tests/test_child_processes_gpu.py::test_tree_mapped_ranks_on_one_gpu"])
print("pytest exit", rc, flush=True)
sys.exit(int(rc))
