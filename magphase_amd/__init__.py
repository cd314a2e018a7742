"""
magphase_amd -- MI355X (gfx950) implementation of the MagPhase vocoder's per-frame analysis / synthesis hot
path behind the reference's own Python API.  See DESIGN.md; the compute lives in csrc/ (HIP) and is reached
through the C ABI of include/magphase_hip.h.
"""
__all__ = ["magphase", "libaudio", "libutils", "libplot", "hostmath", "engine", "synthetic"]
