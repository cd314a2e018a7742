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


def pytest_sessionfinish(session, exitstatus):
    """Measured worst case of every stated tolerance of this session (tests/_tol.py) -> gpurun_out/tolerance_report.json."""
    try:
        import json

        from _tol import FACTS, MEASURED
    except Exception:
        return
    if not MEASURED:
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    rows = {k: dict(v, ratio=(round(v["tol"] / v["max"], 2) if v["max"] > 0 else None)) for k, v in sorted(MEASURED.items())}
    if FACTS:
        rows["_facts"] = dict(sorted(FACTS.items()))
    with open(os.path.join(out_dir, "tolerance_report.json"), "w") as fh:
        json.dump(rows, fh, indent=1)
