import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# Arithmetic of the vocoder's matrix kernels (conan_streams_opts.arith): every GPU test that compares the vocoder or the
# whole step with a reference golden or the oracle runs in both forms.
ARITHS = ("f32", "limb")


def kernels_of(st, fn):
    """Kernel names (as rocprofv3 prints them) -> launches of the matrix kernels that `fn()` enqueued on stream-set `st`
    (blocking entry points only: conan_profile_begin / _end bracket every launch with HIP events)."""
    st.profile_begin()
    fn()
    st.profile_end()
    return {r[0]: r[3] for r in st.profile_kernels()}


def assert_arith_ran(names, arith, limb_expected=True):
    """The kernels that ran match the stream-set's arithmetic: an f32 stream-set must not have launched a limb kernel, and a
    limb stream-set must have launched at least one wherever a limb kernel exists for its launch shapes."""
    limb = sorted(k for k in names if "limb" in k)
    if arith == "f32":
        assert not limb, limb
    elif limb_expected:
        assert limb, sorted(names)
