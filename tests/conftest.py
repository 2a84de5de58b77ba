import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture
def torch_kernels(monkeypatch):
    """CPU test seam: swap the kernel table for the torch restatement (oracle/flat_ref.py) so that the
    product's autograd layer and modules can be exercised without a GPU.  Never used by -m gpu tests."""
    import cgc_net_amd.kernels as kernels
    from oracle.flat_ref import TorchKernels
    monkeypatch.setattr(kernels, '_instance', TorchKernels())
    yield
