import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Parity margins observed by the GPU tests (max error / tolerance per check), written to gpurun_out/parity_margins.json
# at the end of the session so the numbers behind "green" can be read and quoted.
MARGINS = {}


def record_margin(key, value, tol):
    cur = MARGINS.get(key)
    if cur is None or value > cur["value"]:
        MARGINS[key] = {"value": float(value), "tol": float(tol)}


def pytest_sessionfinish(session, exitstatus):
    if not MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_margins.json"), "w") as f:
            json.dump(MARGINS, f, indent=1, sort_keys=True)
    except OSError:
        pass
