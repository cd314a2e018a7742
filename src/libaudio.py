"""Import shim: ``import libaudio as la`` -> magphase_amd.libaudio."""
from magphase_amd.libaudio import *  # noqa: F401,F403
