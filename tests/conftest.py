import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the HIP library and the oracle are compiled (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g

    g.build()
    return g


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def G(built):
    import gradus_jl_amd

    return gradus_jl_amd


@pytest.fixture(scope="session")
def ens(G):
    """One EnsembleMI355X on device 0 for the whole GPU session; fails loudly without a GPU."""
    return G.EnsembleMI355X(0)
