import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure). Built by __graft_entry__.build() / oracle/Makefile."""
    from tests import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dynamic_vins_amd.frontend import Context
    made = []

    def make(**kw):
        c = Context(**kw)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()
