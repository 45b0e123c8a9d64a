import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- parity margins on record ----
# Every relative-error helper of the GPU tests reports the value it computed through note_margin(); with MFVI_MARGINS=<file> the
# session writes one line per test: the number of comparisons and the LARGEST relative error any of them saw (whatever tolerance it
# was held to — the tolerances are in the tests).  scripts/parity_margins.sh runs the GPU suite this way and files the result under
# profiles/.
_MARGINS = {}


def note_margin(value, what="relerr"):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    n, worst, kind = _MARGINS.get(test, (0, 0.0, what))
    _MARGINS[test] = (n + 1, max(worst, float(value)), kind if worst >= float(value) else what)


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("MFVI_MARGINS")
    if not path or not _MARGINS:
        return
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write("# test\tcomparisons\tlargest relative error seen\tkind of the largest\n")
        for test in sorted(_MARGINS):
            n, worst, kind = _MARGINS[test]
            f.write("%s\t%d\t%.3e\t%s\n" % (test, n, worst, kind))
