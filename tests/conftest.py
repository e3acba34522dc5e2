"""pytest configuration: registers the ``gpu`` marker and shared fixtures.

CPU-only tests (``-m "not gpu"``) cover the CPU checker against the committed golden vectors,
the host-side logic and the C-ABI surface.  ``-m gpu`` tests are the parity tests proper: they
drive the HIP kernels through the C ABI and compare with the checker.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def port():
    from oracle import binding
    return binding.load_port()


@pytest.fixture(scope="session")
def ref():
    from oracle import binding
    r = binding.load_ref()
    if r is None:
        pytest.skip("reference tree not mounted and oracle/_ref not prebuilt")
    return r


@pytest.fixture(scope="session")
def golden_scalars():
    return np.load(os.path.join(ROOT, "tests", "golden", "scalars.npz"))


@pytest.fixture(scope="session")
def golden_gemm():
    return np.load(os.path.join(ROOT, "tests", "golden", "gemm_cases.npz"))
