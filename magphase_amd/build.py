"""
Builds libmagphase_hip.so (hipcc, gfx950 only) in-tree next to this file.

    python -m magphase_amd.build [--force] [-D... / -R... extra flags]

One object file per translation unit under magphase_amd/_obj/ (compiled in parallel, rebuilt only when the unit or a
header it includes is newer), then one link.  build() returns the library path and says what it did ("compiled N of M
units" / "reused").

-fno-slp-vectorize: the SLP vectoriser packs the butterflies into v_pk_*_f32 pairs, which have the same
fp32 rate as the scalar ops on gfx950 but need register pairing moves -- the kernels spill without it.
-Wno-inline-asm: the LDS-DMA copies set M0 in inline asm and name it as a clobber; the compiler warns about the reserved
register once per template instantiation (the M0 write is immediately consumed by the following instruction of the same
asm statement, nothing else relies on it).
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ["magphase_hip.hip", "magphase_comp.hip", "magphase_f64.hip", "magphase_epochs.hip", "magphase_probe.hip", "magphase_merlin.hip", "magphase_noise.hip",
         "magphase_host.cpp", "magphase_plan.cpp", "magphase_mtjump.cpp"]
SRCS = [os.path.join(CSRC, u) for u in UNITS]
SRC = SRCS[0]
HEADERS = [os.path.join(CSRC, "wave_fft.hpp"), os.path.join(CSRC, "wave_fft_f64.hpp"),
           os.path.join(CSRC, "mpx_common.hpp"), os.path.join(CSRC, "host_pool.hpp"),
           os.path.join(os.path.dirname(HERE), "include", "magphase_hip.h")]
# The marshalling layer of the batch API (csrc/magphase_pyhost.cpp): a CPython extension beside the C-ABI library, linked
# against it.  It needs Python.h; where that is missing the batch API marshals in Python (slower, same results).
PYHOST_SRC = os.path.join(CSRC, "magphase_pyhost.cpp")
PYHOST = os.path.join(HERE, "_mpx_pyhost.so")
DEPS = SRCS + HEADERS + [PYHOST_SRC]
LIB = os.path.join(HERE, "libmagphase_hip.so")
OBJ_DIR = os.path.join(HERE, "_obj")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-inline-asm", "-fPIC", "-pthread"]
FLAGS = CFLAGS + ["-shared"]   # (kept: tools/ build variants with the full flag list)
last_action = None             # what the last build() call did (read by __graft_entry__.build)


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def is_stale():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(d) > t for d in DEPS):
        return True
    return _python_include() is not None and (not os.path.isfile(PYHOST) or os.path.getmtime(PYHOST) < t)


def _python_include():
    import sysconfig

    inc = sysconfig.get_paths().get("include")
    return inc if inc and os.path.isfile(os.path.join(inc, "Python.h")) else None


def build_pyhost(lib, verbose=True):
    """_mpx_pyhost.so next to `lib` (linked against it, found through $ORIGIN); None when Python.h is not installed."""
    inc = _python_include()
    if inc is None:
        if verbose:
            print("magphase_amd.build: no Python.h: _mpx_pyhost not built (the batch API marshals in Python)", flush=True)
        return None
    out = os.path.join(os.path.dirname(lib), "_mpx_pyhost.so")
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + inc, PYHOST_SRC, "-o", out,
           "-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib), "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def _obj_for(src, extra_flags):
    tag = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8] if extra_flags else "std"
    return os.path.join(OBJ_DIR, "%s.%s.o" % (os.path.basename(src), tag))


def _unit_stale(src, obj):
    if not os.path.isfile(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + HEADERS)


def build(force=False, verbose=True, extra_flags=(), out=None):
    """Compiles what is out of date and links.  out: alternative library path (variant builds of the tools)."""
    global last_action
    lib = out or LIB
    extra_flags = list(extra_flags)
    if not force and out is None and not extra_flags and not is_stale():
        last_action = "reused (library newer than every source)"
        if verbose:
            print("magphase_amd.build: %s: %s" % (last_action, lib), flush=True)
        return lib
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = [_obj_for(s, extra_flags) for s in SRCS]
    todo = [(s, o) for s, o in zip(SRCS, objs) if force or _unit_stale(s, o)]

    def compile_one(so):
        s, o = so
        cmd = [hipcc_path()] + CFLAGS + extra_flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, r.stderr[-4000:]))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        logs = list(ex.map(compile_one, todo))
    if any("-R" in f for f in extra_flags):   # resource-usage remarks asked for: show them
        sys.stderr.write("".join(logs))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if out is None:
        build_pyhost(lib, verbose=verbose)
    last_action = "compiled %d of %d units, linked" % (len(todo), len(SRCS))
    if verbose:
        print("magphase_amd.build: %s: %s" % (last_action, lib), flush=True)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra_flags=[a for a in sys.argv[1:] if a.startswith(("-R", "-D"))])
