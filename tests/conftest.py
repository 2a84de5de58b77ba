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


@pytest.fixture(params=['exact', 'split', 'half'])
def gemm_mode(request, monkeypatch):
    """Run a model-level GPU test three times: with the exact fp32 GEMM, with CGC_GEMM_SPLIT_BF16 (csrc/gemm_split.hip: the big
    products as six bf16 MFMA pairs per fp32 product) and with CGC_GEMM_SPLIT_F16 (csrc/gemm_half.hip: three fp16 pairs of scaled
    operands) -- same test body, same bars.  Encoders built inside the test pick the mode up from the environment
    (network.default_gemm_mode); the fixture also reports how many products took the 16-bit kernels."""
    import cgc_net_amd.kernels as kernels
    split = request.param != 'exact'
    monkeypatch.delenv('CGC_GEMM_SPLIT_BF16', raising=False)
    monkeypatch.setenv('CGC_GEMM_16BIT', {'exact': '0', 'split': '1', 'half': '2'}[request.param])
    K = kernels.get()
    count = lambda: int(K.lib.cgc_gemm_split_count()) + int(K.lib.cgc_gemm_half_count())
    before, before_half = count(), int(K.lib.cgc_gemm_half_count())

    class Mode(object):
        name = request.param
        is_split = split

        @staticmethod
        def launches():
            return count() - before

        @staticmethod
        def half_launches():
            return int(K.lib.cgc_gemm_half_count()) - before_half

        @staticmethod
        def check_applied(least):
            """the mode's kernels took at least ``least`` products (none in the exact mode); mode 'half': the fp16 kernel itself did --
            a product of these sizes is above the threshold below which it hands over to the bf16 kernel"""
            assert (count() - before >= least) == split, (request.param, count() - before)
            if request.param == 'half':
                assert int(K.lib.cgc_gemm_half_count()) - before_half >= least, (int(K.lib.cgc_gemm_half_count()) - before_half, least)
    yield Mode
    K.gemm_mode = kernels.GEMM_EXACT


@pytest.fixture
def forced_big_route():
    """Every product of the test takes the 128 x 128 pipelined route (cgc_gemm_tuning(11)) -- the route the 16-bit GEMM modes apply to --
    at sizes a test can afford (automatically it is taken from ~450 output tiles up: none of the reference-generated fixtures gets
    there by itself), and mode CGC_GEMM_SPLIT_F16 takes every product however small (cgc_gemm_half_min_work(0): by itself it hands
    products below ~28 k tile x k-tile steps to the bf16 kernel)."""
    import ctypes

    import cgc_net_amd.kernels as kernels
    K = kernels.get()
    old = K.lib.cgc_gemm_tuning(11)
    old_work = K.lib.cgc_gemm_half_min_work(ctypes.c_int64(0))
    yield
    K.lib.cgc_gemm_tuning(old)
    K.lib.cgc_gemm_half_min_work(ctypes.c_int64(old_work))
