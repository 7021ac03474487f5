import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS_DIR = os.path.dirname(os.path.abspath(__file__))
if TESTS_DIR not in sys.path:
    sys.path.insert(0, TESTS_DIR)  # shared case modules (wave_cases.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def gpu_verifier():
    """One BatchVerifier for the whole GPU session; fails loudly (no fallback)."""
    import go_ibft_amd.verifier as V
    bv = V.BatchVerifier(device=0, max_rows=65536)
    yield bv
    bv.close()


@pytest.fixture(scope="session")
def gpu_verifier_lane():
    """Same, but pinned to the lane-per-signature kernels (IBFT_KERNEL_LANE): the default context
    uses 2/4/8-lane groups for small batches, this one keeps the G = 1 kernels covered."""
    import go_ibft_amd.verifier as V
    bv = V.BatchVerifier(device=0, max_rows=65536, kernel=V.KERNEL_LANE)
    yield bv
    bv.close()
