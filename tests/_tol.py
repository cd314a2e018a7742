"""
Tolerance bookkeeping of the -m gpu parity tests: every comparison against a stated tolerance goes through within(),
which records the largest value seen per label.  At the end of a session tests/conftest.py writes the table to
gpurun_out/tolerance_report.json (copied to profiles/ per round): the stated tolerances are kept at <= ~3 x the measured
worst case, so a regression of that size fails instead of hiding under a loose bound (parity soft spots).
"""
MEASURED = {}


def within(value, tol, label):
    value, tol = float(value), float(tol)
    rec = MEASURED.setdefault(label, {"max": 0.0, "tol": tol, "n": 0})
    rec["max"] = max(rec["max"], value)
    rec["tol"] = max(rec["tol"], tol)
    rec["n"] += 1
    assert value <= tol, "%s: measured %.4g > tolerance %.4g" % (label, value, tol)
    return True


FACTS = {}


def note(label, value):
    """A measured fact that is not a tolerance (e.g. the fraction of frames a comparison leaves out: 0 since round 4)."""
    FACTS[label] = value
