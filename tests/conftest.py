import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def gpu_state():
    """One Opt_State for the whole GPU session.  Fails (does not skip) if the HIP library is missing."""
    import torch
    from arap_flow_amd import opt
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    st = opt.State()
    yield st
    st.close()
