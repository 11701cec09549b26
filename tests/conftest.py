import os
import sys

import pytest

# One HIP runtime per process: torch bundles its own libamdhip64, and whichever of torch / libgradus_mi355x.so
# touches the GPU first decides which copy the process runs on.  Importing torch at collection time gives every
# subset of the suite the order the full suite (and bench.py) has.  Seen on the GPU pool: library first, torch
# second => torch reports "No HIP GPUs are available".
try:
    import torch  # noqa: F401
except ImportError:      # CPU-only checkouts without torch still run the oracle / host tests
    torch = None

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


# library defaults of every launch knob (gradus_mi355x.hip, struct gr_ctx)
KNOB_DEFAULTS = {"kernel": 2, "block": 0, "refill_threshold": 0, "waves_per_simd": 0, "swizzle": 1, "lpt_lane": 0,
                 "lds": 1, "precision": 64, "lpt": 1, "pipeline": 4, "tangent_pairs": 2, "xcd_spread": 1, "sky_deal": 1}


@pytest.fixture(scope="session")
def _ens_session(G):
    """One EnsembleMI355X (one gr_ctx) on device 0 for the whole GPU session; fails loudly without a GPU."""
    return G.EnsembleMI355X(0)


@pytest.fixture
def ens(_ens_session):
    """The session's ensemble with every launch knob back at the library default, so that no test inherits the
    kernel / precision / lpt / block choice of the test that ran before it."""
    for k, v in KNOB_DEFAULTS.items():
        _ens_session.set(k, v)
    yield _ens_session
    for k, v in KNOB_DEFAULTS.items():
        _ens_session.set(k, v)
