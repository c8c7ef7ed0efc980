import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Bounded look-back spins: the library's default bound (2^22 polls, seconds per give-up)
# is sized for production; under test a wait that long is a bug, and a chain of tiles
# timing out one after another would look like a hang. 2^16 polls is still ~0.1 s.
os.environ.setdefault("CLO_MAX_SPINS", str(1 << 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        from cl_ops_amd import _hip
        return _hip.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """Context + queue on cuda:0; GPU tests fail (not skip) if the library is
    missing, and are deselected by marker on CPU-only boxes."""
    import cl_ops_amd as clo
    if not _have_gpu():
        pytest.fail("no HIP device visible: -m gpu tests need the GPU box")
    ctx = clo.Context(0)
    q = clo.Queue(ctx, profiling=True)
    yield ctx, q
    q.close()
    ctx.close()
