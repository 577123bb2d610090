"""Does the age of a process that holds a GPU context matter to processes started later?  One small
factorization here, then IDLE seconds of sleep (a line every 10 s), then the two-rank test in this
process (its ranks are fresh child processes).  Usage: python scripts/dbg/age_probe.py [IDLE]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)

import pytest  # noqa: E402

idle = int(sys.argv[1]) if len(sys.argv) > 1 else 290
t0 = time.time()
import torch  # noqa: E402
from suitesparse_amd import problems, spqr  # noqa: E402

A = problems.stacked_ls((12, 12, 12))
sym = spqr.analyze(A, problems.nested_dissection((12, 12, 12), leaf=3), gpuMemorySize=2**33)
print("rank", spqr.factorize(sym, A).rank, "pid", os.getpid(), flush=True)
while time.time() - t0 < idle:
    time.sleep(10)
    print("idle %.0f s" % (time.time() - t0), flush=True)
rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", "--timeout", "200", "--timeout-method", "thread",
                  "tests/test_child_processes_gpu.py::test_tree_mapped_ranks_on_one_gpu"])
print("pytest exit", rc, "after %.0f s" % (time.time() - t0), flush=True)
sys.exit(int(rc))
