"""
Epoch (glottal-closure instant) and voicing front end -- SURVEY.md section 8f rank 1.

The reference shells out to REAPER (libaudio.py:450-455, `reaper -s -x 400 -m 50 -a -u 0.005`), an external binary
that exists neither in the build container nor on the GPU box.  This module is NOT a port of REAPER and makes no parity
claim (**parity unpinned**; quality is checked against synthetic utterances whose epochs are known,
tests/test_epochs.py).  It produces what read_reaper_est_file returns -- epoch times in seconds and a 0/1 voicing flag
per epoch, with marks every 5 ms in unvoiced regions (REAPER's `-u 0.005`) -- from two cheap, batch-friendly stages
built from prefix sums, pooling and one small matrix product (tensor ops: they run on the MI355X when a device is
given, on the host otherwise; nothing here is on the hot path):

  1. F0 / voicing track: normalised cross-correlation on the signal low-passed and decimated to 4 kHz, 40 ms frames
     every 5 ms, lags for 60-400 Hz (REAPER's -m 50 -x 400 range), median-smoothed.
  2. Epochs by zero-frequency filtering (Murty & Yegnanarayana 2008): the differenced signal through two
     zero-frequency resonators (four cumulative sums, float64) with the local mean over ~1.5 average pitch periods
     removed after each resonator (keeps the numbers bounded for any length) and twice more at the end; the zero
     crossings of one direction are the glottal closure instants (which direction = the recording's polarity, decided
     by where the excitation energy sits).  Crossings in frames the first stage calls
     unvoiced, or with a weak slope, are dropped.
"""
import numpy as np


def _torch():
    import torch

    return torch


def _movmean_remove(y, n_win):
    """y - centred moving average over n_win (odd) samples, edges by replication; y: float64 [n]."""
    torch = _torch()
    half = n_win // 2
    yp = torch.nn.functional.pad(y.view(1, 1, -1), (half, half), mode="replicate").view(-1)
    c = torch.cumsum(yp, 0)
    c = torch.cat((torch.zeros(1, dtype=y.dtype, device=y.device), c))
    return y - (c[n_win:] - c[:-n_win]) / n_win


def f0_track(x, fs, hop_s=0.005, win_s=0.040, f_lo=60.0, f_hi=400.0, nccf_min=0.5, energy_db=-45.0):
    """
    x: float64 tensor [n].  Returns (f0 [T] (0 = unvoiced), nccf_max [T]) for frames centred at (t + 0.5) * hop_s... the
    frame t covers [t * hop, t * hop + win) of the decimated signal.
    """
    torch = _torch()
    dec = max(1, int(round(fs / 4000.0)))
    fs_d = fs / float(dec)
    xd = torch.nn.functional.avg_pool1d(x.view(1, 1, -1), kernel_size=2 * dec, stride=dec, padding=dec // 2).view(-1)
    xd = xd - xd.mean()
    hop = max(1, int(round(hop_s * fs_d)))
    win = int(round(win_s * fs_d))
    l_min, l_max = max(2, int(fs_d / f_hi)), int(np.ceil(fs_d / f_lo))
    span = win + l_max
    if xd.numel() < span + hop:
        xd = torch.nn.functional.pad(xd, (0, span + hop - xd.numel()))
    fr = xd.unfold(0, span, hop)                                   # [T x span]
    ref = fr[:, :win]
    e_ref = (ref * ref).sum(1)
    lags = torch.arange(l_min, l_max + 1, device=x.device)
    idx = lags.view(-1, 1) + torch.arange(win, device=x.device).view(1, -1)   # [L x win]
    shifted = fr[:, idx]                                           # [T x L x win]
    num = (shifted * ref.unsqueeze(1)).sum(2)
    den = torch.sqrt(e_ref.unsqueeze(1) * (shifted * shifted).sum(2)) + 1e-20
    nccf = num / den                                               # [T x L]
    # prefer the shortest lag among near-equal peaks (octave errors downwards are the common failure)
    best, _ = nccf.max(1)
    ok = nccf >= (best.unsqueeze(1) - 0.06)
    first = torch.argmax(ok.to(torch.int8), dim=1)
    # parabolic refinement around the chosen lag
    li = first.clamp(1, nccf.shape[1] - 2)
    y0, y1, y2 = (nccf.gather(1, (li + d).view(-1, 1)).view(-1) for d in (-1, 0, 1))
    delta = 0.5 * (y0 - y2) / (y0 - 2 * y1 + y2 - 1e-20)
    lag = (li + l_min).to(torch.float64) + delta.clamp(-1, 1)
    f0 = fs_d / lag
    peak = nccf.gather(1, first.view(-1, 1)).view(-1)
    e_db = 10.0 * torch.log10(e_ref / (e_ref.max() + 1e-30) + 1e-30)
    voiced = (peak > nccf_min) & (e_db > energy_db)
    # median-of-5 on the voicing decision and on F0 (isolated flips / octave slips)
    def med5(v):
        vp = torch.nn.functional.pad(v.view(1, 1, -1), (2, 2), mode="replicate").view(-1)
        return vp.unfold(0, 5, 1).median(1).values
    voiced = med5(voiced.to(torch.float64)) > 0.5
    f0 = torch.where(voiced, med5(f0), torch.zeros_like(f0))
    return f0, peak, hop / fs_d, win / fs_d


def zff_epochs(x, fs, t0_s):
    """Zero-frequency filtered signal's positive zero crossings.  x float64 [n]; returns (sample indices, slopes)."""
    torch = _torch()
    n_win = int(round(1.5 * t0_s * fs)) | 1
    dx = torch.cat((x[:1] * 0, x[1:] - x[:-1]))
    y = torch.cumsum(torch.cumsum(dx, 0), 0)
    y = _movmean_remove(y, n_win)
    y = torch.cumsum(torch.cumsum(y, 0), 0)
    for _ in range(3):
        y = _movmean_remove(y, n_win)
    # Differencing (+90 degrees) and four integrations (-360) leave the fundamental of a positive impulse train as
    # -sin(theta): the closure instant theta = 0 is the NEGATIVE-going zero crossing; for inverted recordings it is the
    # positive-going one.  Both sets are returned; the caller keeps the one that sits on the excitation energy.
    out = []
    for sgn in (-1.0, 1.0):
        z = sgn * y
        up = (z[:-1] < 0) & (z[1:] >= 0)
        idx = torch.nonzero(up).view(-1) + 1
        out.append((idx, z[idx] - z[idx - 1]))
    return out, dx


def track_epochs(v_sig, fs, device=None, unvoiced_step_s=0.005):
    """
    v_sig: float array in [-1, 1] (or int16 PCM), fs in Hz -> (v_pm_sec float64 [F], v_voi float64 [F] in {0, 1}),
    the two columns the reference reads from REAPER's .est file.  device: torch device (default: the current ROCm
    device if there is one, else the host).
    """
    torch = _torch()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    v_sig = np.asarray(v_sig)
    if v_sig.dtype.kind in "iu":
        v_sig = v_sig.astype(np.float64) / 32768.0
    x = torch.from_numpy(np.ascontiguousarray(v_sig, dtype=np.float64)).to(device)
    n = x.numel()
    dur = n / float(fs)
    x = x - x.mean()
    f0, _peak, hop_s, win_s = f0_track(x, fs)
    f0_h = f0.cpu().numpy()
    voiced_fr = f0_h > 0
    if not voiced_fr.any():
        t = np.arange(unvoiced_step_s, dur - 2.0 / fs, unvoiced_step_s)
        return np.round(t, 6), np.zeros(t.size)
    t0 = float(np.median(1.0 / f0_h[voiced_fr]))
    cands, dx = zff_epochs(x, fs, t0)
    # polarity: at a closure instant the vocal-tract response is (re-)excited -- the signal energy in the millisecond
    # after it exceeds the energy in the millisecond before; half a period later it is just decaying
    w = max(2, int(round(0.001 * fs)))
    c = torch.cat((torch.zeros(1, dtype=dx.dtype, device=dx.device), torch.cumsum(dx * dx, 0)))
    score = []
    for ci, _sl in cands:
        if ci.numel() == 0:
            score.append(-1e30)
            continue
        a = ci.clamp(w, n - w - 1)
        score.append(float(((c[a + w] - c[a]) - (c[a] - c[a - w])).mean()))
    idx, slope = cands[int(np.argmax(score))]
    idx_h, slope_h = idx.cpu().numpy(), slope.cpu().numpy()
    t_ep = idx_h / float(fs)
    # voicing of each crossing: the F0 frame whose centre is nearest
    fr_of = np.clip(np.round((t_ep - 0.5 * win_s) / hop_s).astype(int), 0, f0_h.size - 1)
    keep = voiced_fr[fr_of]
    if keep.any():
        keep &= slope_h > 0.15 * np.median(slope_h[keep])
    # the period implied by neighbouring crossings must be plausible for the local F0 (drops spurious crossings)
    t_v = t_ep[keep]
    f_v = f0_h[fr_of[keep]]
    if t_v.size > 2:
        good = np.ones(t_v.size, dtype=bool)
        d = np.diff(t_v)
        too_close = d < 0.5 / np.maximum(f_v[1:], 1.0)
        good[1:][too_close] = False
        t_v = t_v[good]
    # unvoiced marks every 5 ms wherever two consecutive voiced epochs are further apart than 2 periods of 50 Hz... i.e.
    # outside voiced runs (REAPER -u 0.005); a voiced run ends when the next epoch is > 20 ms away (1 / 50 Hz)
    pm, voi = [], []
    t_prev = 0.0
    max_gap = 1.0 / 50.0
    for k in range(t_v.size + 1):
        t_next = t_v[k] if k < t_v.size else dur
        if t_next - t_prev > max_gap:
            t = t_prev + unvoiced_step_s
            while t < t_next - 0.5 * unvoiced_step_s:
                pm.append(t)
                voi.append(0.0)
                t += unvoiced_step_s
        if k < t_v.size:
            pm.append(t_next)
            voi.append(1.0)
            t_prev = t_next
    pm = np.asarray(pm)
    voi = np.asarray(voi)
    ok = (pm > 0) & (pm * fs < n - 2)
    return np.round(pm[ok], 6), voi[ok]
