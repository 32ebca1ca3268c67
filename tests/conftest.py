import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "expects_fallback: the test makes a single-launch time loop give up on purpose")


@pytest.fixture(scope="session")
def oracle32():
    import oracle
    return oracle.load("f32")


@pytest.fixture(scope="session")
def oracle64():
    import oracle
    return oracle.load("f64")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _no_silent_fallback(request):
    """A single-launch time loop that gives up is re-run with one launch per step: the results are the same, so a broken
    time loop (a group nobody owns, a hand-off that never arrives) would pass every parity test on the fall-back alone.
    GPU tests therefore fail if the library counted a fall-back they did not ask for (MIFWI_TEST_FAKE_TIMEOUT, or the
    `expects_fallback` marker)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from physicsbasedfwi2_amd import _lib
    before = _lib.load().mifwi_fallback_count()
    yield
    asked = os.environ.get("MIFWI_TEST_FAKE_TIMEOUT") or request.node.get_closest_marker("expects_fallback")
    after = _lib.load().mifwi_fallback_count()
    if not asked:
        assert after == before, "%d single-launch time loop(s) gave up and fell back to one launch per step" % (after - before)
