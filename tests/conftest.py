import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The product library, built in-tree by __graft_entry__.build() (hipcc cross-compiles on CPU)."""
    from openvvc_amd import capi
    if not capi.LIB_PATH.exists():
        import __graft_entry__ as g
        g.build()
    return capi.load()
