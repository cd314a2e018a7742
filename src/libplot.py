"""Import shim: ``from libplot import lp`` -> headless-safe stand-in (magphase_amd.libplot)."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from magphase_amd.libplot import lp  # noqa: E402,F401
