"""
Device plumbing for the HIP hot path: PyTorch-ROCm tensors are the allocator/stream provider, every
computation is a libmagphase_hip.so call (ctypes, include/magphase_hip.h).  One Engine per GPU/process.

Data layout in HBM (all float32, row-major):
  sig      [sum_u n_u]          PCM of the batch's utterances, concatenated
  pos/left/right [F_tot]        per-frame epoch index into sig (int64) and Hann half lengths (int32)
  mag/real/imag  [F_tot x H]    lossless features, H = N/2+1 (same layout as the reference's arrays)
  frames   [F_tot x N]          epoch-centred time-domain frames (scratch between IFFT and PSOLA)
  pcm_out  [sum_u len_u]        resynthesised PCM, concatenated
"""
import os

import numpy as np

from . import _lib, hostmath as hm


def _torch():
    import torch

    return torch


class Engine:
    def __init__(self, device=None):
        torch = _torch()
        self.lib = _lib.load()  # raises if the HIP library is missing: no fallback
        if not torch.cuda.is_available():
            raise _lib.MagphaseHipError("magphase_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                                        "there is no CPU fallback")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self._tables = {}

    # ------------------------------------------------------------------ helpers
    def stream_ptr(self):
        return _torch().cuda.current_stream(self.device).cuda_stream

    def empty(self, shape, dtype=None):
        torch = _torch()
        return torch.empty(shape, dtype=dtype or torch.float32, device=self.device)

    def to_device(self, arr, dtype):
        torch = _torch()
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype))
        return t.to(self.device, non_blocking=False)

    def tables(self, fft_len):
        if fft_len not in self._tables:
            torch = _torch()
            nbytes = self.lib.mpx_tables_bytes(int(fft_len))
            if nbytes == 0:
                raise ValueError("fft_len %r not supported by the HIP path (2048 or 4096)" % (fft_len,))
            t = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.mpx_tables_init(self.stream_ptr(), int(fft_len), t.data_ptr()), "mpx_tables_init")
            self._tables[fft_len] = t
        return self._tables[fft_len]

    # ------------------------------------------------------------------ kernels
    def analysis_frames(self, fft_len, sig, pos, left, right, out=None):
        """sig f32[n], pos i64[F], left/right i32[F] (device) -> (mag, real, imag) f32[F x H] (device)."""
        torch = _torch()
        nfr = int(pos.numel())
        H = fft_len // 2 + 1
        if out is None:
            out = tuple(self.empty((nfr, H)) for _ in range(3))
        tab = self.tables(fft_len)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib.mpx_analysis_frames(self.stream_ptr(), int(fft_len), tab.data_ptr(), sig.data_ptr(),
                                             pos.data_ptr(), left.data_ptr(), right.data_ptr(), nfr,
                                             out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()),
                "mpx_analysis_frames")
        return out

    def synthesis_lossless_frames(self, fft_len, mag, real, imag, out=None):
        torch = _torch()
        nfr = int(mag.shape[0])
        if out is None:
            out = self.empty((nfr, fft_len))
        tab = self.tables(fft_len)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib.mpx_synthesis_lossless_frames(self.stream_ptr(), int(fft_len), tab.data_ptr(), mag.data_ptr(),
                                                       real.data_ptr(), imag.data_ptr(), nfr, out.data_ptr()),
                "mpx_synthesis_lossless_frames")
        return out

    def ola_gather(self, fft_len, frames, utt_frame_off, pm_rel, out_start, out_off, max_out_len, total_out, out=None):
        torch = _torch()
        if out is None:
            out = self.empty((int(total_out),))
        n_utts = int(out_start.numel())
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib.mpx_ola_gather(self.stream_ptr(), int(fft_len), frames.data_ptr(), n_utts,
                                        utt_frame_off.data_ptr(), pm_rel.data_ptr(), out_start.data_ptr(),
                                        out_off.data_ptr(), int(max_out_len), out.data_ptr()),
                "mpx_ola_gather")
        return out


    def synth_ola_slots(self):
        torch = _torch()
        with torch.cuda.device(self.device):
            return int(self.lib.mpx_synth_ola_slots())

    def synthesis_lossless_ola(self, fft_len, mag, real, imag, plan, strips):
        """plan: LosslessSynthesisPlan (chunk + slot tables resident on this device)."""
        torch = _torch()
        tab = self.tables(fft_len)
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib.mpx_synthesis_lossless_ola(self.stream_ptr(), int(fft_len), tab.data_ptr(), mag.data_ptr(),
                                                    real.data_ptr(), imag.data_ptr(), plan.chunks.data_ptr(),
                                                    int(plan.n_chunks), plan.slot_off.data_ptr(),
                                                    plan.slot_chunks.data_ptr(), int(plan.n_slots),
                                                    plan.pm_rel.data_ptr(), int(plan.territory), strips.data_ptr()),
                "mpx_synthesis_lossless_ola")
        return strips

    def ola_fixup(self, fft_len, territory, strips, utt_chunk_off, strip_id, out_start, out_off, max_out_len,
                  total_out, out=None):
        torch = _torch()
        if out is None:
            out = self.empty((int(total_out),))
        n_utts = int(out_start.numel())
        with torch.cuda.device(self.device):
            _lib.check(
                self.lib.mpx_ola_fixup(self.stream_ptr(), int(fft_len), int(territory), strips.data_ptr(), n_utts,
                                       utt_chunk_off.data_ptr(), strip_id.data_ptr(), out_start.data_ptr(),
                                       out_off.data_ptr(), int(max_out_len), out.data_ptr()),
                "mpx_ola_fixup")
        return out


_ENGINES = {}


def get_engine(device=None):
    torch = _torch()
    if device is None:
        if not torch.cuda.is_available():
            return Engine()  # raises the loud error
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    if key not in _ENGINES:
        _ENGINES[key] = Engine(device)
    return _ENGINES[key]


# ======================================================================================================
# Batch plans: host fp64 index math -> descriptor tensors resident in HBM
# ======================================================================================================
class LosslessAnalysisPlan:
    """
    Frame descriptors of a batch of utterances for mpx_analysis_frames.
    utts: list of (v_sig float array in [-1,1) or int16 PCM, fs, v_pm_sec, v_voi).  All must share fft_len.
    Host math follows magphase.py:2877-2879 (pm_sec*fs), libaudio.py:435-447, magphase.py:77-98, :2198-2199.
    """

    def __init__(self, engine, utts, fft_len=None):
        self.engine = engine
        sigs, pos, left, right = [], [], [], []
        self.v_shift, self.v_f0, self.fs, self.n_frames, self.n_smpls, self.v_pm = [], [], [], [], [], []
        off = 0
        for (v_sig, fs, v_pm_sec, v_voi) in utts:
            v_sig = np.asarray(v_sig)
            if v_sig.dtype == np.int16:
                v_sig = v_sig.astype(np.float32) / np.float32(32768.0)
            n = v_sig.shape[0]
            N = fft_len if fft_len is not None else hm.define_fft_len(fs)
            if not hasattr(self, "fft_len"):
                self.fft_len = N
            elif N != self.fft_len:
                raise ValueError("all utterances of a plan must share fft_len (bucket by sample rate)")
            pm_sec, voi = hm.clean_epochs(v_pm_sec, v_voi, check_len_smpls=n, fs=fs)
            pm, lft, rgt = hm.frame_bounds(pm_sec * fs, n)
            sigs.append(v_sig.astype(np.float32, copy=False))
            pos.append(pm + off)
            left.append(lft)
            right.append(rgt)
            self.v_shift.append(lft)
            self.v_pm.append(pm)
            self.v_f0.append(hm.shift_to_f0(lft, voi, fs))
            self.fs.append(fs)
            self.n_frames.append(pm.size)
            self.n_smpls.append(n)
            off += n
        self.total_frames = int(sum(self.n_frames))
        self.frame_off = np.concatenate(([0], np.cumsum(self.n_frames))).astype(np.int64)
        self.long_frame_lens = [(l + r + 1)[(l + r + 1) > self.fft_len].tolist() for l, r in zip(left, right)]
        e = engine
        self.sig = e.to_device(np.concatenate(sigs) if sigs else np.zeros(0), np.float32)
        self.pos = e.to_device(np.concatenate(pos) if pos else np.zeros(0), np.int64)
        self.left = e.to_device(np.concatenate(left) if left else np.zeros(0), np.int32)
        self.right = e.to_device(np.concatenate(right) if right else np.zeros(0), np.int32)
        self.total_smpls = int(off)

    def run(self, out=None):
        return self.engine.analysis_frames(self.fft_len, self.sig, self.pos, self.left, self.right, out=out)


class LosslessSynthesisPlan:
    """
    PSOLA bookkeeping for a batch: per utterance v_f0 (float64) -> shift -> pm (magphase.py:1771-1772, Q2/Q3)
    -> ola() offsets and trimming (magphase.py:34-62).  All float64/int host math; device gets int tables.
    """

    def __init__(self, engine, f0_list, fs_list, fft_len, territory=None):
        self.engine = engine
        self.fft_len = fft_len
        self.territory = int(territory) if territory else int(os.environ.get("MAGPHASE_OLA_TERRITORY", fft_len))
        pm_rel, starts, lens, nfr = [], [], [], []
        self.v_pm = []
        for v_f0, fs in zip(f0_list, fs_list):
            v_pm = np.cumsum(hm.f0_to_shift(np.asarray(v_f0, dtype=np.float64), fs)).astype(int)
            rel, start, out_len = hm.ola_plan(v_pm, fft_len)
            self.v_pm.append(v_pm)
            pm_rel.append(rel)
            starts.append(start)
            lens.append(out_len)
            nfr.append(v_pm.size)
        self.out_len = [int(x) for x in lens]
        self.out_off_host = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        self.total_out = int(self.out_off_host[-1])
        self.max_out_len = int(max(lens)) if lens else 0
        self.total_frames = int(sum(nfr))
        e = engine
        self.utt_frame_off = e.to_device(np.concatenate(([0], np.cumsum(nfr))), np.int32)
        self.pm_rel = e.to_device(np.concatenate(pm_rel) if pm_rel else np.zeros(0), np.int32)
        self.out_start = e.to_device(np.asarray(starts), np.int32)
        self.out_off = e.to_device(self.out_off_host, np.int64)
        self._build_chunks(pm_rel, nfr)

    def _build_chunks(self, pm_rel_list, nfr):
        """Territories of the OLA buffer -> chunks (include/magphase_hip.h: mpx_synthesis_lossless_ola)."""
        rows, terr_off, owner_all = hm.ola_chunks(pm_rel_list, self.fft_len, self.territory)
        e = self.engine
        self.n_chunks = int(rows.shape[0])
        self.chunks = e.to_device(rows, np.int32)
        self.utt_chunk_off = e.to_device(np.asarray(terr_off), np.int32)
        self.strip_id = e.to_device(owner_all, np.int32)
        self.strip_floats = self.n_chunks * (self.territory + self.fft_len)
        n_slots = e.synth_ola_slots() if hasattr(e, "synth_ola_slots") else 1280
        slot_off, slot_chunks = hm.balance_chunks(rows[:, 1] - rows[:, 0], n_slots)
        self.n_slots = int(slot_off.size - 1)
        self.slot_off = e.to_device(slot_off, np.int32)
        self.slot_chunks = e.to_device(slot_chunks, np.int32)

    def run(self, mag, real, imag, strips=None, out=None):
        """Fused path: k_synth_ola (per-chunk LDS overlap-add) + k_ola_fixup."""
        e = self.engine
        if strips is None:
            strips = e.empty((self.strip_floats,))
        e.synthesis_lossless_ola(self.fft_len, mag, real, imag, self, strips)
        return e.ola_fixup(self.fft_len, self.territory, strips, self.utt_chunk_off, self.strip_id, self.out_start,
                           self.out_off, self.max_out_len, self.total_out, out=out)

    def run_unfused(self, mag, real, imag, frames=None, out=None):
        """Two-kernel form: frames to HBM, then the ascending-order gather (bit-for-bit the reference's sum order)."""
        e = self.engine
        frames = e.synthesis_lossless_frames(self.fft_len, mag, real, imag, out=frames)
        return e.ola_gather(self.fft_len, frames, self.utt_frame_off, self.pm_rel, self.out_start, self.out_off,
                            self.max_out_len, self.total_out, out=out)
