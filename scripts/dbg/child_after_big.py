"""Does a child process survive its first host-to-device copy after this process ran a large factorization?"""
import gc, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from suitesparse_amd import _lib, problems, spqr

CHILD_TORCH = "import torch; x = torch.zeros(1 << 22).cuda(); torch.cuda.synchronize(); print('torch child ok')"
CHILD_OURS = ("import sys; sys.path.insert(0, %r); from suitesparse_amd import problems, spqr; "
              "A = problems.stacked_ls((20, 20)); s = spqr.analyze(A, problems.nested_dissection((20, 20), leaf=3), relax=0, gpuMemorySize=2**34); "
              "r = spqr.factorize(s, A); print('our child ok', r.rank)") % ROOT

def child(tag, code):
    rc = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    txt = " | ".join(l for l in rc.stdout.decode().splitlines() if "amdgpu" not in l)
    print(tag, "rc", rc.returncode, txt[-300:], flush=True)
    import torch
    free, total = torch.cuda.mem_get_info()
    print("      parent sees %.1f of %.1f GB device memory free" % (free / 1e9, total / 1e9), flush=True)

shape = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "56x56x56").split("x"))
A = problems.stacked_ls(shape)
sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=8), relax=20, gpuMemorySize=2**36)
res = spqr.factorize(sym, A)
print("parent factorized", shape, "rank", res.rank, "launches", res.gpuNumKernelLaunches, "mode", os.environ.get("GPUQR_MODE"), flush=True)
child("after big, objects alive: torch", CHILD_TORCH)
