"""Import shim: the reference's scripts add <repo>/src to sys.path and import libaudio; this forwards to magphase_amd.libaudio."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from magphase_amd.libaudio import *  # noqa: E402,F401,F403
