import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # the attributes child processes inherit, as they are before any test ran (for comparison with the
    # report a failing child-process test writes)
    try:
        from tests.util import process_attrs
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "attrs_at_session_start.txt"), "w") as f:
            f.write(process_attrs() + "\n")
    except Exception:
        pass


# PyTorch ships its own copy of the HIP runtime.  A process that loads the engine (linked against
# /opt/rocm's runtime) first and PyTorch second ends up with two runtimes, and the second one finds
# no device.  Tests that use torch tensors as device buffers therefore need torch loaded first.
try:
    import torch  # noqa: F401
except ImportError:
    pass


@pytest.fixture(autouse=True, scope="module")
def _release_device_workspaces_between_modules():
    """The library keeps its device workspaces (grow-only, up to the declared device memory) for the
    next call.  A test module that starts child processes on the same GPU needs that memory back: a
    process that finds the device full dies inside the runtime at its first copy instead of getting an
    error.  So every module hands the workspaces back when it is done."""
    yield
    try:
        import torch
        if not torch.cuda.is_available():
            return
        from suitesparse_amd import _lib
        L = _lib.lib()
        L.spqrgpu_amd_finalize()
        L.gpuqr_finalize()
        torch.cuda.empty_cache()
        if os.environ.get("MGPU_AMD_LOG"):
            # debugging aid: what this process still holds after each module
            from tests.util import gpu_state_report
            with open(os.path.join(ROOT, "gpurun_out", "state_by_module.txt"), "a") as f:
                f.write("==== after %s\n%s\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?"), gpu_state_report().split("parent attributes")[0]))
    except Exception:
        pass
