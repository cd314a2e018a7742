"""
Builds libmagphase_hip.so (hipcc, gfx950 only) in-tree next to this file.

    python -m magphase_amd.build

-fno-slp-vectorize: the SLP vectoriser packs the butterflies into v_pk_*_f32 pairs, which have the same
fp32 rate as the scalar ops on gfx950 but need register pairing moves -- the kernels spill without it.
-Wno-inline-asm: the LDS-DMA copies set M0 in inline asm and name it as a clobber; the compiler warns about the reserved
register once per template instantiation (the M0 write is immediately consumed by the following instruction of the same
asm statement, nothing else relies on it).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "magphase_hip.hip")
SRCS = [SRC, os.path.join(HERE, "csrc", "magphase_comp.hip"), os.path.join(HERE, "csrc", "magphase_f64.hip"),
        os.path.join(HERE, "csrc", "magphase_epochs.hip"), os.path.join(HERE, "csrc", "magphase_host.cpp"),
        os.path.join(HERE, "csrc", "magphase_plan.cpp"), os.path.join(HERE, "csrc", "magphase_mtjump.cpp")]
DEPS = SRCS + [os.path.join(HERE, "csrc", "wave_fft.hpp"), os.path.join(HERE, "csrc", "wave_fft_f64.hpp"),
               os.path.join(HERE, "csrc", "mpx_common.hpp"),
               os.path.join(os.path.dirname(HERE), "include", "magphase_hip.h")]
LIB = os.path.join(HERE, "libmagphase_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-inline-asm", "-shared", "-fPIC", "-pthread"]


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def is_stale():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=True, extra_flags=()):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path()] + FLAGS + list(extra_flags) + SRCS + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra_flags=[a for a in sys.argv[1:] if a.startswith(("-R", "-D"))])
