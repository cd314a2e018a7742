"""Import shim: the reference's scripts add <repo>/src to sys.path and import libutils; this forwards to magphase_amd.libutils."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from magphase_amd.libutils import *  # noqa: E402,F401,F403
