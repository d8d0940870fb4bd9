import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def O():
    """CPU oracle (test infrastructure)"""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def R():
    """independent oracles restated from the reference's tests"""
    from oracle import ref_test_oracle
    return ref_test_oracle


@pytest.fixture(scope="session")
def mc_amd():
    """the product package (directory montecarlo.jl_amd); building is a precondition"""
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(g.PKG_DIR, "libdqmc_hip.so")):
        g.build()
    return g.load_package()


@pytest.fixture(scope="session")
def gpu(mc_amd):
    if mc_amd.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X (no CPU fallback exists)")
    return mc_amd


def relerr(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)
