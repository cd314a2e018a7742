"""Import shim: ``from libplot import lp`` -> headless-safe stand-in."""
from magphase_amd.libplot import lp  # noqa: F401
