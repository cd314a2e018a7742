"""Import shim: ``import magphase as mp`` (reference scripts add <repo>/src to sys.path) -> the MI355X implementation."""
from magphase_amd.magphase import *  # noqa: F401,F403
