import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native libraries are build artefacts (git-ignored): build them when a fresh checkout has none
    need = [os.path.join(REPO, "marl-mass_amd", "csrc", "libmm_hip.so"), os.path.join(REPO, "oracle", "libmm_oracle.so")]
    if not all(os.path.exists(f) for f in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
