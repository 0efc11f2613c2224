import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# The CPU oracle is OpenMP code.  A 1-GPU box reports all host cores but grants 16 (cgroup): without a cap
# libgomp starts one thread per reported core and the oracle runs ~10x slower.
if "OMP_NUM_THREADS" not in os.environ:
    try:
        _n = len(os.sched_getaffinity(0))
    except AttributeError:
        _n = os.cpu_count() or 1
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(_n, 16)))
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
