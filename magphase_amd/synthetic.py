"""
Deterministic synthetic utterances for benchmarks and parity tests.

Not part of the reference: the reference ships no benchmark inputs, and REAPER (the epoch
tracker it shells out to, libaudio.py:450-455) is out of scope, so both the signal and its
epochs/voicing flags are generated here.  Same generator for the CPU baseline and the GPU path.
"""
import numpy as np
from scipy import signal as _sps


def make_utterance(u, dur_s=5.0, fs=48000):
    """
    Returns (pcm int16[n], v_pm_sec float64[F] (6 decimals, the .est text precision), v_voi float64[F]).

    rng = RandomState(1000+u); alternating voiced (0.3-0.9 s) / unvoiced (0.1-0.3 s) segments;
    voiced F0(t) = fc*(1+0.25 sin(2 pi r t + phi)), fc~U[90,260] Hz, r~U[0.2,0.6] Hz, epochs by
    accumulating 1/F0; unvoiced epochs every 5 ms (REAPER ``-u 0.005`` convention);
    signal = impulses at voiced epochs + 0.02 N(0,1) in unvoiced regions through three 2-pole
    resonators (700/130, 1220/170, 2600/250 Hz), peak 0.5, quantised to int16.
    """
    rng = np.random.RandomState(1000 + u)
    n = int(round(dur_s * fs))
    fc = rng.uniform(90.0, 260.0)
    r = rng.uniform(0.2, 0.6)
    phi = rng.uniform(0.0, 2 * np.pi)
    exc = np.zeros(n)
    pm, voi = [], []
    t = 0.0
    voiced = bool(rng.randint(0, 2))
    t_seg_end = 0.0
    while True:
        if t >= t_seg_end:
            voiced = not voiced
            t_seg_end = t + (rng.uniform(0.3, 0.9) if voiced else rng.uniform(0.1, 0.3))
            if not voiced:
                a = int(t * fs)
                b = min(int(t_seg_end * fs), n)
                if b > a:
                    exc[a:b] += 0.02 * rng.randn(b - a)
        if voiced:
            f0 = fc * (1 + 0.25 * np.sin(2 * np.pi * r * t + phi))
            t += 1.0 / f0
        else:
            t += 0.005
        if t * fs >= n - 2:
            break
        pm.append(t)
        voi.append(1.0 if voiced else 0.0)
        if voiced:
            exc[int(round(t * fs))] += 1.0
    sig = exc
    for f_c, b_w in ((700.0, 130.0), (1220.0, 170.0), (2600.0, 250.0)):
        rad = np.exp(-np.pi * b_w / fs)
        a = [1.0, -2 * rad * np.cos(2 * np.pi * f_c / fs), rad * rad]
        sig = _sps.lfilter([1.0], a, sig)
    sig = 0.5 * sig / np.max(np.abs(sig))
    pcm = np.round(sig * 32767.0).astype(np.int16)
    v_pm_sec = np.round(np.asarray(pm, dtype=np.float64), 6)
    return pcm, v_pm_sec, np.asarray(voi, dtype=np.float64)


def pcm_to_float(pcm):
    """What soundfile.read returns for 16-bit PCM (magphase.py:2872): int16 / 32768 as float64."""
    return pcm.astype(np.float64) / 32768.0
