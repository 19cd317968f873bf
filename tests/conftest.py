import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import gko_oracle
    gko_oracle.lib()
    return gko_oracle


@pytest.fixture(scope="session")
def gexec():
    """Cdna4Executor on cuda:0; fails loudly if the HIP library is missing."""
    import ginkgo_amd as g
    return g.Cdna4Executor.create(0)
