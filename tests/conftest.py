import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuauv-vision-pipeline_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def vp():
    """libvp binding with a live context; fails loudly (no skip) when the HIP library is absent."""
    import torch  # noqa: F401  -- before libvp: both must share the HIP runtime torch ships, whichever test file runs first
    from vision import _vp
    _vp.lib()
    _vp.default_context()
    return _vp
