import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_SESSION_FILLER = None


def session_filler():
    """the filler that runs beside the whole `-m gpu` session (None when it is off or there is no GPU): tests that need an IDLE GPU as their reference pause it"""
    return _SESSION_FILLER


@pytest.fixture(scope="session", autouse=True)
def _gpu_contention(request):
    """The whole `-m gpu` session runs while a background stream floods the GPU with filler kernels (tests/test_contention.py::Filler): every parity test is thereby also a
    test that its result does not depend on what else the GPU is doing — the class of defect round 4 found in the accept decision, which three rounds of green GPU suites on an
    idle device had not seen.  ON BY DEFAULT for a session that selects the gpu marker (the driver's `pytest -m gpu` is the contention run); DVINS_GPU_CONTENTION=0 switches it
    off (timing-sensitive debugging), =1 forces it for any session."""
    global _SESSION_FILLER
    env = os.environ.get("DVINS_GPU_CONTENTION")
    markexpr = getattr(request.config.option, "markexpr", "") or ""
    want = env == "1" or (env != "0" and "gpu" in markexpr and "not gpu" not in markexpr)
    if not want:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from tests.test_contention import Filler
    with Filler() as f:
        _SESSION_FILLER = f
        try:
            yield
        finally:
            _SESSION_FILLER = None
        print(f"\n[contention] filler launches beside the session: {f.launched}")


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


def iterations_agree(sd, so):
    """Window-solve iteration counts of the HIP path (sd) and the oracle (so).  They are identical unless the prior's constant c0 = r0^T r0 differs:
    the reference forms r0 = S^-1/2 Q^T b over eigenvalues > 1e-8 of a matrix of norm ~1e9, so c0 carries O(1) rounding noise from the gauge directions
     that no two implementations (nor two builds of the reference) reproduce.  c0 moves neither the minimiser nor the step acceptance, only
    ceres' RELATIVE function-tolerance test |dcost| <= 1e-6 cost — by the same few percent the cost moved, which can flip the test by one iteration.
    Accepted: equal counts, or a difference of one while the two initial costs differ by a constant offset (> 1e-5 relative)."""
    if sd.iterations == so.iterations:
        return True
    off = abs(sd.initial_cost - so.initial_cost)
    return abs(sd.iterations - so.iterations) == 1 and off > 1e-5 * max(abs(so.initial_cost), 1e-300)
