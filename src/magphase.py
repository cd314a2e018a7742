"""Import shim: the reference's scripts add <repo>/src to sys.path and import magphase; this forwards to magphase_amd.magphase."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from magphase_amd.magphase import *  # noqa: E402,F401,F403
