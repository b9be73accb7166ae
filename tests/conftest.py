import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture
def cpu_double():
    """install the torch-CPU kernel test double for the duration of a test"""
    from surreal_amd import kernels as KN
    from cpu_kernels import TorchCpuKernels
    prev = KN.set_default_kernels(TorchCpuKernels(), 'cpu')
    yield
    KN.set_default_kernels(*prev)
