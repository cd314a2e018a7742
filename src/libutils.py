"""Import shim: ``import libutils as lu`` -> magphase_amd.libutils."""
from magphase_amd.libutils import *  # noqa: F401,F403
