import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# PyTorch ships its own copy of the HIP runtime.  A process that loads the engine (linked against
# /opt/rocm's runtime) first and PyTorch second ends up with two runtimes, and the second one finds
# no device.  Tests that use torch tensors as device buffers therefore need torch loaded first.
try:
    import torch  # noqa: F401
except ImportError:
    pass
