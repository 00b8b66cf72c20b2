import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    # the .so is a build artefact (git-ignored): a fresh checkout compiles it once (hipcc cross-compiles
    # gfx950 without a GPU); an existing one is used as is
    lib_path = os.path.join(ROOT, "myrtlespeech_amd", "libms_hotpath.so")
    if not os.path.exists(lib_path):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def lib():
    """The C-ABI library handle (loads on CPU too; compute entry points need a GPU)."""
    from myrtlespeech_amd import _lib

    return _lib.load()
