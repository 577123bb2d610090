"""GPU tests whose work is done by child processes that open the GPU themselves: ranks of a
tree-mapped factorization sharing the box's one GPU, and a first-call-of-a-process case.

They live in a module of their own that sorts early among the GPU tests on purpose.  On the test boxes a
process that has been through most of the GPU suite (the 128^3 run, the full-size comparisons with the
reference, hundreds of factorizations) makes every process started after it die of a segmentation fault
inside the HSA runtime when it creates its first hardware queue, until the old process exits: queue
creation throws, and the runtime's clean-up then reads an unset scratch node.  Nothing visible from user
space separates the two states of the old process (device memory free to the last GiB, no page-locked
memory, same mappings, threads, descriptors and inheritable attributes; age alone, repeated release of
its streams, or any single test module do not do it), so the child processes are started while this
process is young."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from suitesparse_amd import problems, spqr
from tests.util import r_rel_diff, rownorm_sign

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,shape", [(2, "20x20"), (2, "10x10x10"), (3, "9x9x9")])
def test_tree_mapped_ranks_on_one_gpu(tmp_path, world, shape):
    """Several processes (gloo exchange) sharing the test box's one GPU: the tree is mapped onto the
    ranks, each factorizes its fronts level by level with spqrgpu_amd_factorize_subset_dev (contribution
    blocks stay on the device on both sides of a call), the blocks that change rank travel point to
    point.  R against LAPACK."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "R.npz")
    # this process is about to share its GPU with the workers: give back the device workspaces, streams
    # and events that earlier tests left cached in it
    from suitesparse_amd import _lib
    _lib.lib().spqrgpu_amd_finalize()
    _lib.lib().gpuqr_finalize()
    # (a port of its own per case: the store of the previous case may not have let go of its port yet)
    port = 29500 + (os.getpid() % 2000) + 2000 * (1 + ["20x20", "10x10x10", "9x9x9"].index(shape))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "mgpu_worker.py"), "gloo-gpu", out, shape]
    # own session, so that a hang can be ended together with the workers it left behind; output to
    # files, so that nothing blocks on a pipe a surviving grandchild still holds
    import signal
    trace = str(tmp_path / "trace")
    with open(str(tmp_path / "out.txt"), "w") as fo:
        from tests.util import fresh_process_attrs
        proc = subprocess.Popen(cmd, stdout=fo, stderr=subprocess.STDOUT, start_new_session=True,
                                preexec_fn=fresh_process_attrs,
                                env=dict(os.environ, MGPU_TRACE=trace, GPUQR_BACKTRACE="1",
                                         **({"AMD_LOG_LEVEL": "4", "HSAKMT_DEBUG_LEVEL": "7"} if os.environ.get("MGPU_AMD_LOG") else {})))
        try:
            proc.wait(timeout=100)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait(timeout=20)
            report = open(str(tmp_path / "out.txt")).read()[-3000:]
            for r in range(world):
                try:
                    report += "\n-- rank %d --\n%s" % (r, open("%s.%d" % (trace, r)).read())
                except OSError:
                    report += "\n-- rank %d: no trace --" % r
            raise AssertionError("multi-rank run hung:\n" + report)
    stdout, stderr = open(str(tmp_path / "out.txt")).read(), ""
    if proc.returncode != 0:
        from tests.util import gpu_state_report
        report = gpu_state_report(trace, world)
        try:                            # the whole output of the ranks, for reading after the run
            os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
            with open(os.path.join(root, "gpurun_out", "mgpu_fail_%d_%s.txt" % (world, shape)), "w") as f:
                f.write(stdout + "\n" + report + "\n" + "\n".join("%s=%s" % kv for kv in sorted(os.environ.items())
                                                                 if kv[0].startswith(("GPUQR", "HIP", "HSA", "ROC", "AMD", "MKL", "OMP", "GPU"))))
        except OSError:
            pass
        raise AssertionError(stdout[-5000:] + "\n" + report)
    z = np.load(out)
    R = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"])).toarray()
    dims = tuple(int(x) for x in shape.split("x"))
    A = problems.stacked_ls(dims)
    Rl = np.linalg.qr(A[:, z["Qfill"]].toarray(), mode="r")
    assert int(z["n_cut"]) >= world - 1 and int(z["levels"]) >= 2
    assert r_rel_diff(rownorm_sign(R), rownorm_sign(Rl)) <= 1e-12


def test_four_ranks_give_the_single_process_r_bit_for_bit(tmp_path):
    """Four processes on the one GPU, a 3D grid with three levels of the rank map: the ranks run the
    same tasks on the same numbers as one process does, contribution blocks are copied, never summed,
    so R must agree bit for bit with the single-process factorization."""
    import os
    import signal
    import subprocess
    import sys
    from suitesparse_amd import _lib
    _lib.lib().spqrgpu_amd_finalize()
    _lib.lib().gpuqr_finalize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, shape = str(tmp_path / "R.npz"), "24x24x24"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + (os.getpid() % 2000) + 9000),
           os.path.join(root, "tests", "mgpu_worker.py"), "gloo-gpu", out, shape]
    with open(str(tmp_path / "out.txt"), "w") as fo:
        proc = subprocess.Popen(cmd, stdout=fo, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            proc.wait(timeout=150)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait(timeout=20)
            raise AssertionError("multi-rank run hung:\n" + open(str(tmp_path / "out.txt")).read()[-3000:])
    assert proc.returncode == 0, open(str(tmp_path / "out.txt")).read()[-6000:]
    z = np.load(out)
    R = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    dims = (24, 24, 24)
    A = problems.stacked_ls(dims)
    sym = spqr.analyze(A, problems.nested_dissection(dims, leaf=3), relax=0, gpuMemorySize=2**34)
    one = spqr.factorize(sym, A).R.tocsc()
    assert int(z["levels"]) >= 3 and int(z["n_cut"]) >= 3
    assert R.shape == one.shape and R.nnz == one.nnz and (R != one).nnz == 0


def test_stage_without_r_between_two_stages_in_a_fresh_process():
    """One front per stage, pageable stack, fronts without rows among them (a wide matrix): the R of a
    stage is copied out by a helper thread from one of two device blocks that take turns; a stage that
    has no R never waited for that thread, and the stage after it packed into the block still being
    read.  Only the first call of a process is slow enough to show it, hence the child process."""
    import subprocess
    import sys
    env = dict(os.environ, GPUQR_PAGEABLE_R="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sweeps", "dbg", "binding_race.py"), "10371", "5000000"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("shape,mem,limit_gib", [("20x20x20", 80_000_000, 0.0112), ("96x96", 20_000_000, 0.0028), ("96x96", 10_000_000, 0.0028)])
def test_end_of_a_pageable_stack_page_locked(shape, mem, limit_gib):
    """An R stack beyond the page-locking limit is pageable; the driver page-locks as much of its end as
    the limit allows, and the groups whose R lies there send it from inside the engine while the groups
    before them go through the bounce buffers.  Exercised at a small size by a process with a tiny limit
    (read once per process): same bits as the one-stage run, and the log says the mixed hand-off ran."""
    import subprocess
    import sys
    env = dict(os.environ, GPUQR_PIN_LIMIT_GIB=str(limit_gib), GPUQR_KEEP_PAGEABLE_FROM_MB="0", GPUQR_VERBOSE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tail_worker.py"), shape, str(mem)],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "same bits in all runs" in r.stdout
    said = [ln for ln in r.stderr.splitlines() if "goes out from inside the engine" in ln]
    assert len(said) == 2, r.stderr[-3000:]          # the two runs without GPUQR_LATE_R
