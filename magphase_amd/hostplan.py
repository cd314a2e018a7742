"""
Batch planners in the library (csrc/magphase_plan.cpp) behind numpy-friendly wrappers.  They are the same float64 /
integer arithmetic as the numpy forms in hostmath.py / engine.py (which remain: they are what a failed native call falls
back to -- raising the exceptions the reference's arithmetic would -- and what tests/test_host_plans.py compares these
against, bit for bit), for a whole batch per call instead of ~60 numpy calls per utterance.
MAGPHASE_NATIVE_PLAN=0 disables them.
"""
import os

import numpy as np

from . import _lib
from .hostmath import OLA_RUN_DTYPE


class PlanFallback(Exception):
    """The native planner declined (an utterance the numpy form raises on, or planners disabled): use the numpy form."""


def enabled():
    return os.environ.get("MAGPHASE_NATIVE_PLAN", "1") != "0"


def _cat(arrs, dtype):
    if len(arrs) == 1:
        return np.ascontiguousarray(arrs[0], dtype=dtype).reshape(-1)
    return np.concatenate([np.asarray(a, dtype=dtype).reshape(-1) for a in arrs])


def plan_analysis(pm_sec_list, voi_list, n_smpls, fs_list, sig_off):
    """-> dict(pos, pm, left, right (int64[F]), f0 (float64[F]), frame_off (int64[U+1])) for the batch."""
    if not enabled():
        raise PlanFallback()
    lib = _lib.load()
    U = len(pm_sec_list)
    pm_sec = _cat(pm_sec_list, np.float64)
    voi = _cat(voi_list, np.float64)
    sizes = [int(np.size(p)) for p in pm_sec_list]
    if [int(np.size(v)) for v in voi_list] != sizes:
        raise PlanFallback()
    ep_off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    E = int(ep_off[-1])
    n_smpls = np.ascontiguousarray(n_smpls, dtype=np.int64)
    fs = np.ascontiguousarray(fs_list, dtype=np.float64)
    sig_off = np.ascontiguousarray(sig_off, dtype=np.int64)
    pos, pm, left, right = (np.empty(max(E, 1), dtype=np.int64) for _ in range(4))
    f0 = np.empty(max(E, 1), dtype=np.float64)
    frame_off = np.empty(U + 1, dtype=np.int64)
    F = int(lib.mpx_host_plan_analysis(U, pm_sec.ctypes.data, voi.ctypes.data, ep_off.ctypes.data, n_smpls.ctypes.data,
                                       fs.ctypes.data, sig_off.ctypes.data, pos.ctypes.data, pm.ctypes.data,
                                       left.ctypes.data, right.ctypes.data, f0.ctypes.data, frame_off.ctypes.data))
    if F < 0:
        raise PlanFallback()
    return dict(pos=pos[:F], pm=pm[:F], left=left[:F], right=right[:F], f0=f0[:F], frame_off=frame_off)


def plan_synthesis(f0_list, fs, fft_len, b_const_rate, b_voi_ap_win):
    """f0_list: exp(lf0) per utterance.  -> dict of the per-frame tables of CompressedSynthesisPlan for the batch."""
    if not enabled():
        raise PlanFallback()
    lib = _lib.load()
    U = len(f0_list)
    f0 = _cat(f0_list, np.float64)
    row_off = np.concatenate(([0], np.cumsum([int(np.size(f)) for f in f0_list]))).astype(np.int64)
    cap = 2 * int(row_off[-1]) + 2
    i64 = lambda: np.empty(cap, dtype=np.int64)      # noqa: E731
    i32 = lambda: np.empty(cap, dtype=np.int32)      # noqa: E731
    o = dict(v_shift=i64(), v_pm=i64(), npos=i64(), nleft=i32(), nright=i32(), wtype=i32(), voiced=i32(), row0=i32(),
             row1=i32(), rowt=np.empty(cap, dtype=np.float64), win_l=i32(), win_r=i32(), pm_rel=i64())
    frame_off = np.empty(U + 1, dtype=np.int64)
    ns_len, out_start, out_len = (np.empty(max(U, 1), dtype=np.int64) for _ in range(3))
    F = int(lib.mpx_host_plan_synthesis(
        U, f0.ctypes.data, row_off.ctypes.data, float(fs), int(fft_len), int(bool(b_const_rate)), int(bool(b_voi_ap_win)),
        cap, o["v_shift"].ctypes.data, o["v_pm"].ctypes.data, o["npos"].ctypes.data, o["nleft"].ctypes.data,
        o["nright"].ctypes.data, o["wtype"].ctypes.data, o["voiced"].ctypes.data, o["row0"].ctypes.data,
        o["row1"].ctypes.data, o["rowt"].ctypes.data, o["win_l"].ctypes.data, o["win_r"].ctypes.data,
        o["pm_rel"].ctypes.data, frame_off.ctypes.data, ns_len.ctypes.data, out_start.ctypes.data, out_len.ctypes.data))
    if F < 0:
        raise PlanFallback()
    o = {k: v[:F] for k, v in o.items()}
    o.update(frame_off=frame_off, ns_len=ns_len[:U], out_start=out_start[:U], out_len=out_len[:U], row_off=row_off)
    return o


def plan_lossless_synthesis(f0_list, fs_list, fft_len):
    """-> dict(v_pm, pm_rel (int64[F]), frame_off (int64[U+1]), out_start, out_len (int64[U]))."""
    if not enabled():
        raise PlanFallback()
    lib = _lib.load()
    U = len(f0_list)
    if U == 0:
        raise PlanFallback()
    f0 = _cat(f0_list, np.float64)
    frame_off = np.concatenate(([0], np.cumsum([int(np.size(f)) for f in f0_list]))).astype(np.int64)
    fs = np.ascontiguousarray(fs_list, dtype=np.float64)
    F = int(frame_off[-1])
    v_pm, pm_rel = np.empty(max(F, 1), dtype=np.int64), np.empty(max(F, 1), dtype=np.int64)
    out_start, out_len = np.empty(U, dtype=np.int64), np.empty(U, dtype=np.int64)
    rc = int(lib.mpx_host_plan_lossless_synthesis(U, f0.ctypes.data, frame_off.ctypes.data, fs.ctypes.data, int(fft_len),
                                                  v_pm.ctypes.data, pm_rel.ctypes.data, out_start.ctypes.data,
                                                  out_len.ctypes.data))
    if rc < 0:
        raise PlanFallback()
    return dict(v_pm=v_pm[:F], pm_rel=pm_rel[:F], frame_off=frame_off, out_start=out_start, out_len=out_len)


def ola_runs(pm_rel_cat, frame_off, starts, out_lens, out_offs, fft_len, n_slots, weights=None):
    """hostmath.ola_runs (default equal-share mode) on the concatenated frame positions -> (runs, slot_off, slot_runs)."""
    if not enabled():
        raise PlanFallback()
    lib = _lib.load()
    frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
    U = int(frame_off.size - 1)
    total = int(frame_off[-1])
    n_slots = max(1, int(n_slots))
    from . import hostmath as hm

    gcuts = np.ascontiguousarray(hm.slot_cuts(total, n_slots, weights))
    pm_rel = np.ascontiguousarray(pm_rel_cat, dtype=np.int64)
    starts, out_lens, out_offs = (np.ascontiguousarray(a, dtype=np.int64) for a in (starts, out_lens, out_offs))
    cap = U + int(gcuts.size) + 1
    runs = np.zeros(cap, dtype=OLA_RUN_DTYPE)
    n = int(lib.mpx_host_ola_runs(U, pm_rel.ctypes.data, frame_off.ctypes.data, starts.ctypes.data, out_lens.ctypes.data,
                                  out_offs.ctypes.data, int(fft_len), gcuts.ctypes.data, int(gcuts.size),
                                  runs.ctypes.data, cap))
    if n < 0:
        raise PlanFallback()
    runs = runs[:n]
    ns = gcuts.size - 1
    slot_of = np.clip(np.searchsorted(gcuts, runs["frame_begin"], side="right") - 1, 0, ns - 1)
    slot_off = np.searchsorted(slot_of, np.arange(ns + 1), side="left").astype(np.int64)
    return runs, slot_off, np.arange(runs.size, dtype=np.int64)


# ----------------------------------------------------------------------------------------------------------------------
# Whole-launch planners (mpx_host_plan_analysis_batch / mpx_host_plan_synthesis_batch) through the marshalling layer
# _mpx_pyhost (csrc/magphase_pyhost.cpp): the utterance list is walked once in native code, the interpreter lock is
# released for the call.  Used by Engine.prepare_analysis / prepare_synthesis; the list-based functions above stay as the
# generic path (any array-like input) and as what the tests compare against.
# ----------------------------------------------------------------------------------------------------------------------
_PYHOST = False


def pyhost():
    """The _mpx_pyhost extension module, or None (not built: no Python.h at build time; MAGPHASE_PYHOST=0)."""
    global _PYHOST
    if _PYHOST is False:
        _PYHOST = None
        if enabled() and os.environ.get("MAGPHASE_PYHOST", "1") != "0":
            try:
                _lib.load()                      # the extension links against the C-ABI library
                from . import _mpx_pyhost
                _PYHOST = _mpx_pyhost
            except Exception:
                _PYHOST = None
    return _PYHOST


# order of the device tables mpx_host_plan_synthesis_batch lays out in its `desc` buffer: (name, numpy dtype)
SYNTH_TABLES = (("utt_frame_off", np.int32), ("npos", np.int64), ("nleft", np.int32), ("nright", np.int32),
                ("wtype", np.int32), ("voiced", np.int32), ("tile_first", np.int32), ("row0", np.int32),
                ("row1", np.int32), ("rowt", np.float32), ("win_l", np.int32), ("win_r", np.int32), ("pm_rel", np.int32),
                ("out_start", np.int32), ("out_off", np.int64), ("runs", np.uint8), ("slot_off", np.int32),
                ("slot_runs", np.int32))


def synth_desc_bytes(n_rows, n_utts, n_slots, b_const_rate):
    """Upper bound of the bytes mpx_host_plan_synthesis_batch writes into `desc`."""
    cap = (2 * int(n_rows) + 2 * int(n_utts)) if b_const_rate else int(n_rows)
    runs = int(n_utts) + int(n_slots) + 1
    return 52 * cap + 4 * (int(n_rows) // 31 + 3) + 16 * (int(n_utts) + 1) + 60 * runs + 4 * (int(n_slots) + 1) + 256 * 20
