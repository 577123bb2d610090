"""One-off check: R of a 3D grid from 4 ranks sharing one GPU (gloo exchange, device blocks) against the
single-process factorization: the same tasks run either way, so the two must agree bit for bit."""
import os, subprocess, sys, tempfile
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from suitesparse_amd import problems, spqr
shape = sys.argv[1] if len(sys.argv) > 1 else "32x32x32"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dims = tuple(int(x) for x in shape.split("x"))
out = os.path.join(tempfile.mkdtemp(), "R.npz")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
       "--master-port", "29733", os.path.join(ROOT, "tests", "mgpu_worker.py"), "gloo-gpu", out, shape]
rc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=500)
print("workers rc", rc.returncode, rc.stdout.decode()[-1500:] if rc.returncode else "")
z = np.load(out)
R = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
A = problems.stacked_ls(dims)
sym = spqr.analyze(A, problems.nested_dissection(dims, leaf=3), relax=0, gpuMemorySize=2**34)
one = spqr.factorize(sym, A).R.tocsc()
print("levels", int(z["levels"]), "cut blocks", int(z["n_cut"]), "bound", float(z["bound"]))
print("nnz", R.nnz, one.nnz, "differing entries", (R != one).nnz, "max |diff|", abs(R - one).max() if R.nnz else 0)
