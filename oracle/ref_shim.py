"""
TEST INFRASTRUCTURE ONLY -- loader for the *real* reference (runs only where /root/reference exists).

The reference (CSTR-Edinburgh/magphase, /root/reference/src/{libutils,libaudio,magphase}.py) is
Python-2.7 source.  This module imports it *unmodified, in memory* under Python 3 so that
(a) the numpy restatement in oracle/magphase_oracle.py can be validated against it and
(b) golden vectors can be generated from it (oracle/gen_golden.py -> tests/golden/*.npz).

Nothing of the reference is copied into the repo: the source text is read from
/root/reference at run time, transformed in memory and exec'd.  This file never travels
into a product path; it is imported only by oracle/gen_golden.py and by the
"reference present" tests (skipped on the GPU box, where /root/reference does not exist).

Recipe:
  1. lib2to3 refactor (xrange, ``raise X, "m"``, ConfigParser, unicode, print).
  2. AST rewrite of every ``a / b`` into ``__py2div__(a, b)``: Python-2 semantics, i.e.
     floor division when both operands are integers (the reference relies on it for
     indices: magphase.py:59, :841; libaudio.py:123).
  3. exec into fresh modules registered as ``libutils``, ``libaudio``, ``magphase``.
  4. ``np.float`` / ``np.complex`` aliases (removed in numpy 2; used at magphase.py:557).
  5. a stub ``soundfile`` module (16-bit PCM mono wav reader/writer).
  6. ``la.reaper`` is replaced by a writer of a REAPER-format ``.est`` file from epochs the
     caller supplies (REAPER itself is an external binary absent here: parity unpinned).
  7. ``la.sp_to_mcep`` is replaced by the SPTK-3.9 ``mcep`` restatement of
     oracle/magphase_oracle.py (SPTK is an external binary absent here: parity unpinned).
"""
import ast
import os
import sys
import types
import wave

import numpy as np

REF_ROOT = os.environ.get("MAGPHASE_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src")


def reference_available():
    return os.path.isfile(os.path.join(REF_SRC, "magphase.py"))


# ----------------------------------------------------------------------------- py2 division
def __py2div__(a, b):
    def _is_int(x):
        if isinstance(x, (bool, np.bool_)):
            return True
        if isinstance(x, (int, np.integer)):
            return True
        if isinstance(x, np.ndarray) and (np.issubdtype(x.dtype, np.integer) or x.dtype == bool):
            return True
        return False

    if _is_int(a) and _is_int(b):
        return a // b
    return a / b


class _DivRewriter(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(
                ast.Call(func=ast.Name(id="__py2div__", ctx=ast.Load()), args=[node.left, node.right], keywords=[]),
                node,
            )
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            # a /= b  ->  a = __py2div__(a, b)   (targets in the reference are plain names/subscripts)
            load_target = ast.parse(ast.unparse(node.target), mode="eval").body
            return ast.copy_location(
                ast.Assign(
                    targets=[node.target],
                    value=ast.Call(func=ast.Name(id="__py2div__", ctx=ast.Load()), args=[load_target, node.value], keywords=[]),
                ),
                node,
            )
        return node


def _py2_to_py3(src, name):
    from lib2to3 import refactor

    fixers = refactor.get_fixers_from_package("lib2to3.fixes")
    tool = refactor.RefactoringTool(fixers)
    if not src.endswith("\n"):
        src += "\n"
    return str(tool.refactor_string(src, name))


def _load_module(name, extra_globals=None):
    path = os.path.join(REF_SRC, name + ".py")
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        src = f.read()
    src3 = _py2_to_py3(src, name)
    tree = ast.parse(src3, filename=path)
    tree = _DivRewriter().visit(tree)
    ast.fix_missing_locations(tree)
    code = compile(tree, path, "exec")
    mod = types.ModuleType(name)
    mod.__file__ = path
    mod.__dict__["__py2div__"] = __py2div__
    if extra_globals:
        mod.__dict__.update(extra_globals)
    sys.modules[name] = mod
    exec(code, mod.__dict__)
    return mod


# ----------------------------------------------------------------------------- soundfile stub
def _wav_read(path):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2 and w.getnchannels() == 1, "stub handles 16-bit mono only"
        fs = w.getframerate()
        raw = w.readframes(w.getnframes())
    v = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    return v, fs


def _wav_write(path, data, fs, *a, **k):
    v = np.clip(np.asarray(data, dtype=np.float64), -1.0, 1.0 - 1.0 / 32768)
    pcm = np.round(v * 32768.0).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(fs))
        w.writeframes(pcm.tobytes())


def _install_soundfile_stub():
    sf = types.ModuleType("soundfile")
    sf.read = _wav_read
    sf.write = _wav_write
    sys.modules["soundfile"] = sf
    return sf


# ----------------------------------------------------------------------------- epochs
_EPOCHS = {}


def set_epochs(wav_file, v_pm_sec, v_voi):
    """Register the epochs that the patched la.reaper will 'detect' for wav_file."""
    _EPOCHS[os.path.realpath(wav_file)] = (np.asarray(v_pm_sec, dtype=np.float64), np.asarray(v_voi))


def write_est_file(path, v_pm_sec, v_voi):
    """REAPER .est text layout as read at libaudio.py:421-431: 7 header lines, then 'time voiced f0'."""
    with open(path, "w") as f:
        f.write("EST_File Track\nDataType ascii\nNumFrames %d\nNumChannels 1\n" % len(v_pm_sec))
        f.write("FrameShift 0.0\nVoicingEnabled true\nEST_Header_End\n")
        for t, v in zip(v_pm_sec, v_voi):
            f.write("%.6f %d 0.0\n" % (t, int(v)))


def _fake_reaper(in_wav_file, out_est_file):
    key = os.path.realpath(in_wav_file)
    if key not in _EPOCHS:
        raise RuntimeError("ref_shim: no epochs registered for %s (call set_epochs first)" % in_wav_file)
    write_est_file(out_est_file, *_EPOCHS[key])


# ----------------------------------------------------------------------------- public
_LOADED = None


def load_reference():
    """Returns (magphase, libaudio, libutils) modules of the real reference, shimmed."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REF_SRC)
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "complex"):
        np.complex = complex
    _install_soundfile_stub()
    lu = _load_module("libutils")
    la = _load_module("libaudio")
    # libplot is skipped (matplotlib Qt4Agg); nothing on the hot path imports it at module level.
    mp = _load_module("magphase")
    la.reaper = _fake_reaper
    from oracle import magphase_oracle as orc  # SPTK restatement lives with the oracle

    def _sp_to_mcep(m_sp, n_coeffs=60, alpha=0.77, in_type=3, fft_len=0):
        return orc.sptk_mcep(m_sp, n_coeffs=n_coeffs, alpha=alpha, in_type=in_type, fft_len=fft_len)

    la.sp_to_mcep = _sp_to_mcep
    _LOADED = (mp, la, lu)
    return _LOADED
