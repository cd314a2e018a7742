"""
Host-side float64 index arithmetic of the MagPhase hot path.

The reference computes every epoch / shift / frame index in float64 numpy with np.round (half-to-even),
truncating int casts and sequential cumsum.  "Bit-exact indices" means reproducing
that IEEE-754 op sequence, so it stays on the host in numpy and is never recomputed on the device.
Reference lines are cited per function.
"""
import warnings

import numpy as np

MAGIC = -1.0e10  # libaudio.py:17


def round_to_int(x):
    """libutils.py:131-133."""
    return np.round(x).astype(int)


def define_alpha(fs):
    """magphase.py:3279-3290."""
    table = {16000: 0.58, 22050: 0.65, 44100: 0.76, 48000: 0.77}
    if fs not in table:
        raise ValueError("Sample rate %d not supported yet." % fs)
    return table[fs]


def define_fft_len(fs):
    """magphase.py:3292-3299."""
    if fs in (22050, 16000):
        return 2048
    if fs == 8000:
        return 1024
    return 4096


def define_crossfade_params(fs):
    """magphase.py:3301-3317."""
    crsf_bw = 2000
    if fs == 48000:
        return 5000, crsf_bw
    if fs == 16000:
        return 2500, crsf_bw
    warnings.warn("Constant crsf_cf not tested nor tunned to synthesise at fs=%d Hz." % fs)
    return (4500 if fs == 44100 else 3500), crsf_bw


def clean_epochs(v_pm_sec, v_voi, check_len_smpls=-1, fs=-1):
    """libaudio.py:435-447: the two protections applied to REAPER's epoch list."""
    if (check_len_smpls > 0) and (fs == -1):
        raise ValueError("If check_len_smpls given, fs must be provided as well.")
    v_pm_sec = np.asarray(v_pm_sec, dtype=np.float64)
    v_voi = np.asarray(v_voi, dtype=np.float64)
    keep = np.hstack((True, np.diff(v_pm_sec) > 0))
    v_pm_sec, v_voi = v_pm_sec[keep], v_voi[keep]
    if check_len_smpls > 0:
        pm = round_to_int(v_pm_sec * fs)
        if pm[-1] >= (check_len_smpls - 1):
            keep2 = pm < (check_len_smpls - 1)
            v_pm_sec, v_voi = v_pm_sec[keep2], v_voi[keep2]
    return v_pm_sec, v_voi


def frame_bounds(v_pm_smpls, n_smpls):
    """
    magphase.py:77-83,90-98: epochs rounded (Q1), extended with 0 and n-1 (Q4).
    Returns (pm int64[F], left int64[F], right int64[F]); left is the reference's v_shift.
    """
    pm = round_to_int(np.asarray(v_pm_smpls))
    ext = np.hstack((0, pm, n_smpls - 1))
    return pm.astype(np.int64), (ext[1:-1] - ext[:-2]).astype(np.int64), (ext[2:] - ext[1:-1]).astype(np.int64)


def shift_to_f0(v_shift, v_voi, fs):
    """magphase.py:2198-2207 with b_smooth=False (Q2)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return v_voi * fs / v_shift.astype("float64")


def f0_to_shift(v_f0_in, fs, unv_frm_rate_ms=5):
    """magphase.py:2210-2215 (Q3)."""
    v_f0 = np.array(v_f0_in, dtype=np.float64)
    v_f0[v_f0 == 0] = 1000.0 / unv_frm_rate_ms
    return fs / v_f0


def medfilt3_batch(vectors):
    """scipy.signal.medfilt(v) (kernel 3, zero-padded ends) for a list of 1-D float64 vectors in ONE numpy pass: the
    vectors are laid end to end with a zero between them (the padding scipy adds), the median of every three neighbours is
    their sum minus their minimum and maximum -- exact (all three operands are the inputs themselves: the median is
    selected, not computed).  Returns a list of arrays equal to [signal.medfilt(v) for v in vectors] bit for bit."""
    sizes = [int(np.size(v)) for v in vectors]
    if not sizes:
        return []
    total = int(sum(sizes)) + len(sizes) + 1
    cat = np.zeros(total, dtype=np.float64)
    starts = np.cumsum([1] + [n + 1 for n in sizes[:-1]])
    for v, a, n in zip(vectors, starts, sizes):
        cat[a:a + n] = v
    lo, mid, hi = cat[:-2], cat[1:-1], cat[2:]
    med = np.maximum(np.minimum(lo, mid), np.minimum(np.maximum(lo, mid), hi))   # median of three by selection
    return [med[a - 1:a - 1 + n] for a, n in zip(starts, sizes)]


HANN_TABLE_CAP = 2048   # half lengths covered by hann_half_table (F0 >= 23.4 Hz at 48 kHz); longer halves: analytic window


def hann_half_table(cap=HANN_TABLE_CAP):
    """
    The rising halves np.hanning(2 h + 1)[0 .. h] for every half length h <= cap, laid end to end (half h at offset
    h (h + 1) / 2; (cap + 1)(cap + 2) / 2 float64 values, 16.8 MB at cap = 2048).  Built with numpy's own np.hanning -- the
    function the reference windows with (libaudio.py:70-84: both halves of a frame are rising halves, the right one
    flipped) -- so that mpx_analysis_frames_f64w multiplies by the reference's very weights.  np.hanning(1) = [1.0].
    """
    out = np.empty((cap + 1) * (cap + 2) // 2, dtype=np.float64)
    for h in range(cap + 1):
        off = h * (h + 1) // 2
        out[off:off + h + 1] = np.hanning(2 * h + 1)[:h + 1]
    return out


def ola_plan(v_pm, frmlen):
    """
    Index bookkeeping of magphase.py:34-62 (ola): returns (pm_rel int64[F], out_start, out_len) such that
    out[t] = sum_i frame_i[t + out_start - pm_rel[i]].  Python slice semantics are kept, including the
    negative-start case ``v_sig[(frmlen/2 - pm_0):]`` when the first epoch lies beyond frmlen/2.
    """
    v_pm = np.asarray(v_pm).astype(int)
    buf_len = int(v_pm[-1]) + frmlen
    start = frmlen // 2 - int(v_pm[0])
    if start < 0:
        start = max(buf_len + start, 0)
    start = min(start, buf_len)
    len1 = buf_len - start
    last_shift = int(v_pm[-1] - v_pm[-2]) if v_pm.size > 1 else int(v_pm[-1])
    stop = int(v_pm[-1]) + last_shift + 1
    if stop < 0:
        stop = max(len1 + stop, 0)
    out_len = min(len1, stop)
    return (v_pm - v_pm[0]).astype(np.int64), int(start), int(out_len)


# numpy record layout of the C struct mpx_ola_run (include/magphase_hip.h), 56 bytes
OLA_RUN_DTYPE = np.dtype([("frame_begin", "<i4"), ("frame_end", "<i4"), ("x0", "<i4"), ("head_end", "<i4"),
                          ("out_lo", "<i4"), ("out_hi", "<i4"), ("flush_end", "<i4"), ("fix_lo", "<i4"),
                          ("fix_hi", "<i4"), ("pad", "<i4"), ("out_base", "<i8"), ("strip_off", "<i8")])


def _run_cuts(rel, N, target):
    """
    Frame indices at which one utterance's frames are cut into runs of about ``target`` frames, such that only ADJACENT
    runs overlap in the OLA buffer: rel[cut_{k+1}] - rel[cut_k - 1] >= N for every run k with both neighbours
    (a frame covers [rel, rel + N)).  Returns int64 cuts, cuts[0] == 0, cuts[-1] == n.
    """
    n = int(rel.size)
    k = max(1, min(n, int(round(n / float(max(1, target))))))
    cuts = np.round(np.linspace(0, n, k + 1)).astype(np.int64)
    if k > 2:
        inner = cuts[1:-1]   # runs 1 .. k-2 have both neighbours: span from the frame before their first to the next run's first
        ok = np.all(rel[inner[1:]] - rel[inner[:-1] - 1] >= N)
    else:
        ok = True
    if ok and (k < 2 or np.all(np.diff(cuts) > 0)):
        return cuts
    # rare (tiny utterances, tiny targets): greedy left-to-right
    out = [0]
    fb = 0
    while True:
        fe = min(n, fb + max(1, int(target)))
        if fb > 0:
            while fe < n and rel[fe] - rel[fb - 1] < N:
                fe += 1
        if fe >= n:
            break
        out.append(fe)
        fb = fe
    out.append(n)
    return np.asarray(out, dtype=np.int64)


def _enforce_span(rel, N, cuts):
    """Moves (or, where there is no room, drops) cuts until every run with both neighbours satisfies
    rel[next run's first] - rel[own first - 1] >= N: a cut that comes too early is moved forward to the first frame
    that is far enough, as long as that leaves its successor a frame; otherwise the run grows into its successor.
    (Round 4: until then a violating cut was always dropped -- one slot idle and its neighbour with twice the frames;
    with shares below ~25 frames per run that made the pair kernels 45 % slower.)"""
    cuts = [int(c) for c in cuts]
    rel = np.asarray(rel)
    k = 1
    while k < len(cuts) - 2:
        need = int(rel[cuts[k] - 1]) + N
        if rel[cuts[k + 1]] < need:
            c2 = int(np.searchsorted(rel, need, side="left"))   # rel ascends within an utterance
            if c2 < cuts[k + 2]:
                cuts[k + 1] = c2     # the run takes the first frames of its successor's share
                k += 1
            else:
                del cuts[k + 1]      # no frame of the successor is far enough: the run grows into it
        else:
            k += 1
    return np.asarray(cuts, dtype=np.int64)


def slot_cuts(total, n_slots, weights=None):
    """Frame indices that deal `total` frames to the slots: equal shares, or shares in proportion to `weights` (the
    slots' relative speeds, mpx_synth_ola_slot_weights).  int64[min(n_slots, max(total, 1)) + 1], cuts[0] == 0."""
    ns = min(max(1, int(n_slots)), max(int(total), 1))
    if weights is None:
        return np.round(np.linspace(0, total, ns + 1)).astype(np.int64)
    w = np.asarray(weights, dtype=np.float64)[:ns]
    if w.size != ns or np.any(w <= 0):
        raise ValueError("slot weights must be positive, one per slot")
    return np.round(total * np.concatenate(([0.0], np.cumsum(w))) / w.sum()).astype(np.int64)


def ola_runs(pm_rel_list, starts, out_lens, out_offs, fft_len, n_slots, frames_per_run=None, weights=None):
    """
    Plans the fused overlap-add (include/magphase_hip.h: mpx_synthesis_lossless_ola).  The batch's frames, in utterance
    order, are dealt to the device's pair slots in consecutive shares: equal ones (slot s gets the frames
    [round(s F / n_slots), round((s+1) F / n_slots)) of the concatenated sequence), or, with ``weights`` (one relative
    speed per slot, mpx_synth_ola_slot_weights), shares in proportion to them -- the kernel ends when the slowest slot does.  A share that crosses an utterance boundary is two (or more) RUNS -- the end
    of one utterance and the beginning of the next; runs never cross utterances.  ``frames_per_run`` instead cuts every
    utterance on its own into runs of about that many frames and balances the slots longest-run-first (tests, tuning).
    Per run the positions are classified as head strip / final output / dropped (see mpx_ola_run) in the coordinates
    of the reference's OLA buffer (magphase.py:38-61): frame i covers [pm_rel[i], pm_rel[i] + N), the kept part is
    [start, start + out_len).  Only ADJACENT runs of an utterance may overlap: rel[next run's first frame] -
    rel[own first frame - 1] >= N for every run with both neighbours (a cut violating it is MOVED FORWARD to the first
    frame that satisfies it, _enforce_span; only a cut with no such frame left in the utterance is dropped).

    pm_rel_list: per utterance int64[F_u]; starts / out_lens: ola_plan's (out_start, out_len) per utterance;
    out_offs: int64[U+1] offsets of the utterances in pcm_out.
    Returns (runs, slot_off, slot_runs): a structured array (OLA_RUN_DTYPE) of the runs in utterance / frame order and
    the slots' work lists (slot s processes runs slot_runs[slot_off[s] : slot_off[s+1]] in that order).
    """
    N = int(fft_len)
    strip_floats = N + 64
    n_frames = np.asarray([int(np.size(r)) for r in pm_rel_list], dtype=np.int64)
    total = int(n_frames.sum())
    n_slots = max(1, int(n_slots))
    if frames_per_run:
        target, gcuts = int(frames_per_run), None
    else:
        target = max(1, -(-total // n_slots))
        gcuts = slot_cuts(total, n_slots, weights)
    recs = []
    f_base = 0
    for u, rel in enumerate(pm_rel_list):
        rel = np.asarray(rel, dtype=np.int64)
        n = int(rel.size)
        if n == 0:
            continue
        start, out_len, o0 = int(starts[u]), int(out_lens[u]), int(out_offs[u])
        if gcuts is None:
            cuts = _run_cuts(rel, N, target)
        else:   # the global cuts that fall inside this utterance
            inner = gcuts[(gcuts > f_base) & (gcuts < f_base + n)] - f_base
            cuts = _enforce_span(rel, N, np.concatenate(([0], inner, [n])))
        fb, fe = cuts[:-1], cuts[1:]
        k = fb.size
        hi = rel[fe - 1] + N                          # end of the run's last frame
        prev_hi = np.concatenate(([0], hi[:-1]))      # positions < prev_hi also get the previous run's frames
        # first position the run is responsible for: its first frame, or the end of the previous run's last frame if
        # that comes first (consecutive frames further apart than N leave a gap of zeros, which this run writes)
        lo = np.minimum(rel[fb], prev_hi)
        # element 0 at a position whose pcm_out index is a multiple of 64 (aligned 256-byte output blocks)
        x0 = lo - ((lo - start + o0) % 64)
        head_end = np.where(np.arange(k) > 0, prev_hi - x0, 0)
        if np.any(head_end > strip_floats):
            raise ValueError("ola_runs: a run's head overlap exceeds the strip (frames not in ascending position order?)")
        # final positions of run k: [prev_hi (0 for the first run), hi_k), the last run up to the end of the kept part
        own_lo = np.where(np.arange(k) > 0, prev_hi, 0)
        own_hi = hi.copy()
        own_hi[-1] = max(int(hi[-1]), start + out_len)
        out_lo = np.maximum(own_lo, start) - x0
        out_hi = np.minimum(own_hi, start + out_len) - x0
        out_hi = np.maximum(out_hi, out_lo)
        flush_end = np.maximum(hi, own_hi) - x0
        fix_lo = np.maximum(lo, start) - x0
        fix_hi = np.minimum(prev_hi, start + out_len) - x0
        fix_hi = np.where(np.arange(k) > 0, np.maximum(fix_hi, fix_lo), fix_lo)
        r = np.zeros(k, dtype=OLA_RUN_DTYPE)
        r["frame_begin"], r["frame_end"] = fb + f_base, fe + f_base
        r["x0"], r["head_end"] = x0, head_end
        r["out_lo"], r["out_hi"], r["flush_end"] = out_lo, out_hi, flush_end
        r["fix_lo"], r["fix_hi"] = fix_lo, fix_hi
        r["out_base"] = o0 + x0 - start
        recs.append(r)
        f_base += n
    runs = np.concatenate(recs) if recs else np.zeros(0, dtype=OLA_RUN_DTYPE)
    runs["strip_off"] = np.arange(runs.size, dtype=np.int64) * strip_floats
    if gcuts is None:
        slot_off, slot_runs = balance_chunks(runs["frame_end"] - runs["frame_begin"], n_slots)
    else:   # a run belongs to the share its first frame lies in; shares are consecutive, so are their runs
        ns = gcuts.size - 1
        slot_of = np.clip(np.searchsorted(gcuts, runs["frame_begin"], side="right") - 1, 0, ns - 1)
        slot_off = np.searchsorted(slot_of, np.arange(ns + 1), side="left").astype(np.int64)
        slot_runs = np.arange(runs.size, dtype=np.int64)
    return runs, slot_off, slot_runs


def balance_chunks(n_frames_per_chunk, n_slots, overhead=2):
    """
    Longest-processing-time-first assignment of chunks (already sorted longest first) to wave slots.
    cost(chunk) = frames + overhead (ring tail flush ~ 2 frames' worth of LDS/global work).
    Returns (slot_off int64[n_slots+1], slot_chunks int64[n_chunks]).
    """
    import heapq

    n_frames_per_chunk = np.asarray(n_frames_per_chunk, dtype=np.int64)
    n_slots = int(max(1, min(n_slots, max(1, n_frames_per_chunk.size))))
    heap = [(0, s) for s in range(n_slots)]
    lists = [[] for _ in range(n_slots)]
    for ci in np.argsort(-n_frames_per_chunk, kind="stable"):
        load, s_ = heapq.heappop(heap)
        lists[s_].append(int(ci))
        heapq.heappush(heap, (load + int(n_frames_per_chunk[ci]) + overhead, s_))
    slot_off = np.concatenate(([0], np.cumsum([len(l) for l in lists]))).astype(np.int64)
    slot_chunks = np.asarray([c for l in lists for c in l], dtype=np.int64)
    return slot_off, slot_chunks


# ======================================================================================================
# compressed-feature constants (float64 on the host, uploaded once per (fs, dims) as float32)
# ======================================================================================================
def warp_axis(alpha, nbins):
    """libaudio.py:612-614 / :711-715: all-pass warped frequency axis on nbins points in [0, pi]."""
    w = np.linspace(0, np.pi, num=nbins)
    ww = np.arctan((1 - alpha ** 2) * np.sin(w) / ((1 + alpha ** 2) * np.cos(w) - 2 * alpha))
    ww[ww < 0] += np.pi
    return ww


def build_mel_curve(alpha, nbins, amp=np.pi):
    """libaudio.py:711-718."""
    return warp_axis(alpha, nbins) * (amp / np.pi)


_COS_CACHE = {}


def cos_matrix(n_cep, n_spbins, alpha):
    """libaudio.py:611-619: trans[i, k] = cos(i * warp_alpha(pi k / (n_spbins - 1))), cached per configuration."""
    key = (int(n_cep), int(n_spbins), float(alpha))
    if key not in _COS_CACHE:
        _COS_CACHE[key] = np.cos(np.arange(key[0])[:, None] * warp_axis(key[2], key[1])[None, :])
    return _COS_CACHE[key]


def unwarp_matrix(ncoeffs, nbins_out, alpha):
    """
    la.sp_mel_unwarp(in_type='log') (libaudio.py:667-684) as the matrix it is: out = x @ U,
    U[ncoeffs x nbins_out].  Even extension -> real IFFT -> coefficients 1..ncoeffs-3 doubled (Q6) ->
    cosine matrix on the alpha-warped axis (libaudio.py:605-619).
    """
    eye = np.eye(ncoeffs)
    ext = np.hstack((eye, eye[:, -2:0:-1]))
    ceps = np.fft.ifft(ext).real[:, :ncoeffs]
    ceps[:, 1:(ncoeffs - 2)] *= 2
    cosm = np.cos(np.arange(ncoeffs)[:, None] * warp_axis(alpha, nbins_out)[None, :])
    return ceps @ cosm


def fbank_centres(n_melbands, nbins, alpha):
    """Band centres (linear-frequency bins) of the mel filter bank, libaudio.py:839-846 / :730-736."""
    from scipy import interpolate

    v_bins_warp = build_mel_curve(alpha, nbins, amp=np.pi)
    v_cntrs_mel = np.linspace(0, v_bins_warp[-1], n_melbands)
    f_interp = interpolate.interp1d(v_bins_warp, np.arange(nbins), kind="quadratic")
    return round_to_int(f_interp(v_cntrs_mel))


def warp_fbank_matrix(n_melbands, nbins, alpha):
    """
    la.sp_mel_warp_fbank / apply_fbank(mode='average') (libaudio.py:721-769) as W[n_melbands x nbins] acting on
    ln(mag): band b = the asymmetric Hann window (rising half of np.hanning(1 + 2 l), falling half of
    np.hanning(1 + 2 r), l / r = distances to the neighbouring band centres, first and last centre repeated)
    normalised to unit sum, placed from centre b-1 on.  Same float64 construction as the reference.
    """
    v_cntrs = fbank_centres(n_melbands, nbins, alpha)
    ext = np.r_[v_cntrs[0], v_cntrs, v_cntrs[-1]]
    w = np.zeros((n_melbands, nbins))
    for b in range(1, n_melbands + 1):
        ll, rr = int(ext[b] - ext[b - 1]), int(ext[b + 1] - ext[b])
        win = np.hstack((np.hanning(1 + 2 * ll)[:ll + 1], np.flipud(np.hanning(1 + 2 * rr)[:rr + 1])[1:]))
        win = win / np.sum(win)
        w[b - 1, ext[b - 1]:ext[b - 1] + win.size] = win
    return w


def unwarp_fbank_matrix(n_melbands, nbins, alpha):
    """
    la.sp_mel_unwarp_fbank (libaudio.py:815-864): per frame, a quadratic spline (scipy interp1d) through the band
    centres, evaluated on every bin.  For fixed knots that is linear in the frame, so it is the matrix
    U[n_melbands x nbins] = the spline of each unit vector (same scipy call as the reference).
    """
    from scipy import interpolate

    v_cntrs = fbank_centres(n_melbands, nbins, alpha)
    f_interp = interpolate.interp1d(v_cntrs, np.eye(n_melbands), kind="quadratic", axis=0)
    return f_interp(np.arange(nbins)).T.copy()


def get_num_full_mel_coeffs_from_num_phase_coeffs(freq_hz, phase_dim, alpha, fs):
    """magphase.py:2479-2487."""
    cw = 2 * np.pi * freq_hz / float(fs)
    cf_mel = np.arctan((1 - alpha ** 2) * np.sin(cw) / ((1 + alpha ** 2) * np.cos(cw) - 2 * alpha))
    if cf_mel < 0:
        cf_mel += np.pi
    return int(round_to_int(1 + (np.pi * (phase_dim - 1) / float(cf_mel))))


def phase_unwarp_matrix(phase_dim, fft_len, fs, alpha):
    """
    magphase.py:1219-1235 as a matrix [phase_dim x H]: nearest-neighbour extension phase_dim -> K (the last
    coefficient repeated) followed by the K-coefficient unwarp: rows >= phase_dim-1 of U_K fold into the last row.
    """
    cf = define_crossfade_params(fs)[0]
    k_full = get_num_full_mel_coeffs_from_num_phase_coeffs(cf, phase_dim, alpha, fs)
    u_full = unwarp_matrix(k_full, fft_len // 2 + 1, alpha)
    if k_full <= phase_dim:
        return u_full[:phase_dim].copy() if k_full == phase_dim else np.vstack(
            (u_full, np.zeros((phase_dim - k_full, u_full.shape[1]))))  # (index >= K never read: columns cut)
    u = u_full[:phase_dim].copy()
    u[phase_dim - 1] += u_full[phase_dim:].sum(axis=0)
    return u


def crossfade_lowpass_curve(nbins_half, cut_off, bw, fs):
    """libaudio.py:160-186 for (ones, zeros): 1 below bin_l, falling Hann half on [bin_l, bin_r], 0 above."""
    nfft = (nbins_half - 1) * 2
    bin_l = int(round_to_int((cut_off - bw / 2.0) * nfft / float(fs)))
    bin_r = int(round_to_int((cut_off + bw / 2.0) * nfft / float(fs)))
    bw_bin = bin_r - bin_l
    return np.hstack((np.ones(bin_l), np.hanning(2 * bw_bin + 1)[bw_bin:], np.zeros(nbins_half - bin_r - 1)))


def synthesis_bin_curves(fs, fft_len):
    """
    Per-bin constant vectors of synthesis_from_compressed (magphase.py:873-875, 917-918, 940-941, 946-952):
      per_v = tilt_voiced * sqrt(mask)      periodic component of voiced frames  (0 where mask == 0)
      ap_v  = sqrt(1 - mask)                aperiodic component of voiced frames (0 where mask == 1)
      ap_u  = tilt_unvoiced                 aperiodic component of unvoiced frames (mask == 0 there)
    tilt_voiced = 10**(melcurve(0.6, H, 2.0)/20); tilt_unvoiced = 10**((melcurve(alpha, H, 3.5) - 3.5)/20) (Q13).
    """
    half = fft_len // 2 + 1
    cf, bw = define_crossfade_params(fs)
    alpha = define_alpha(fs)
    mask = crossfade_lowpass_curve(half, cf, bw, fs)
    tilt_v = 10 ** (build_mel_curve(0.6, half, amp=2.0) / 20)
    tilt_u = 10 ** ((build_mel_curve(alpha, half, amp=3.5) - 3.5) / 20)
    return tilt_v * mask ** 0.5, (1 - mask) ** 0.5, tilt_u


# ======================================================================================================
# mel warp (compressed analysis) as a matrix -- SPTK-3.9 ``mcep -j 0`` restated (PARITY UNPINNED: external binary)
# ======================================================================================================
def freqt_matrix(n_in, order_out, alpha):
    """
    Frequency transformation of cepstra by a first-order all-pass (Tokuda et al. 1994; SPTK ``freqt``) as a matrix
    A[(order_out+1) x n_in]: mc = A @ c.  The recursion consumes the input from c[n_in-1] down to c[0]; running it on
    the identity gives the matrix.  alpha == 0 is the identity truncation.
    """
    m2 = int(order_out)
    if alpha == 0.0:
        a_mat = np.zeros((m2 + 1, n_in))
        a_mat[np.arange(min(m2 + 1, n_in)), np.arange(min(m2 + 1, n_in))] = 1.0
        return a_mat
    b = 1.0 - alpha * alpha
    g = np.zeros((n_in, m2 + 1))          # one "frame" per unit input vector
    for i in range(n_in - 1, -1, -1):
        d = g.copy()
        g[:, 0] = alpha * d[:, 0]
        g[i, 0] += 1.0                     # c1[-i] of the unit vector e_i
        if m2 >= 1:
            g[:, 1] = b * d[:, 0] + alpha * d[:, 1]
        for j in range(2, m2 + 1):
            g[:, j] = d[:, j - 1] + alpha * (d[:, j] - g[:, j - 1])
    return g.T.copy()


_WARP_CACHE = {}


def warp_matrix(nbins_out, nbins_half, alpha, nrows=None):
    """
    la.sp_mel_warp (libaudio.py:643-661) as W[nrows x nbins_half] acting on the log-periodogram:
      c[n]  = (1/N) sum_k w_k logP[k] cos(2 pi k n / N), n = 0..N/2   (real IFFT of the even extension; w_0 = w_{N/2} = 1, else 2)
      c[0] /= 2, c[N/2] /= 2 ;  mc = freqt(c, nbins_out - 1, alpha) ;  out[i] = sum_n mc[n] cos(n pi i / (nbins_out - 1))
    alpha is rounded to two decimals as on the mcep command line (libaudio.py:589).  nrows: keep only the first rows
    (the phase features are cut to phase_dim, magphase.py:2523-2524).
    """
    key = (int(nbins_out), int(nbins_half), float("%1.2f" % alpha))
    if key not in _WARP_CACHE:
        nb, half = key[0], key[1]
        n_fft = 2 * (half - 1)
        w = np.full(half, 2.0)
        w[0] = w[-1] = 1.0
        idct = np.cos(2 * np.pi * np.outer(np.arange(half), np.arange(half)) / n_fft) * w[None, :] / n_fft
        idct[0] *= 0.5
        idct[-1] *= 0.5
        a_mat = freqt_matrix(half, nb - 1, key[2])
        cosm = np.cos(np.arange(nb)[:, None] * np.linspace(0, np.pi, nb)[None, :]).T
        _WARP_CACHE[key] = cosm @ (a_mat @ idct)
    wm = _WARP_CACHE[key]
    return wm if nrows is None else wm[:nrows]


def _kappa(P, lane):
    """wave_fft.hpp kappa(lane): the output-index residue a lane holds (identity for P = 32)."""
    lane = np.asarray(lane)
    if P == 32:
        return lane
    if P == 16:
        return (lane & 15) | (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 5)
    return (lane & 7) | (((lane >> 5) & 1) << 3) | (lane & 16) | (((lane >> 3) & 1) << 5)


def fused_chunk_bins(fft_len):
    """Bin of column c of chunk q in mpx_analysis_compressed_fused's published tile: int64[P/2, 128] --
    c < 64: the low bin 64 q + c, c >= 64: its mirror M - 64 q - (c - 64)  (M = fft_len / 2; bin M/2 is not in any chunk)."""
    P = fft_len // 128
    M = fft_len // 2
    q = np.arange(P // 2)[:, None]
    c = np.arange(64)[None, :]
    return np.concatenate((64 * q + c, M - 64 * q - c), axis=1).astype(np.int64)


def pack_warp_fused(w_mag, w_ph, fft_len, n_waves=8, layout=0):
    """
    The two warp matrices ([mag_dim x H] and [phase_dim x H], float64) in the order mpx_analysis_compressed_fused's MFMA
    (v_mfma_f32_16x16x4_f32) consumes them.  A workgroup of n_waves waves (mpx_analysis_compressed_fused_waves) cuts a
    chunk's 128 columns into n_waves slices of 128 / n_waves columns = kh groups of 16:
      wpack[q][wave][h][tile][lane][e] = W_tile[16 jt + (lane & 15)][bin(q, (128 / n_waves) wave + 16 h + 4 (lane >> 4) + e)]
    -- one 16-byte load per lane, tile and 16-column group; tiles = 4 magnitude column tiles, then ceil(phase_dim / 16)
    phase tiles (shared by the real and the imaginary stream); rows past the matrix are zero.
    whalf[tile][16] = the same rows' weight of bin M/2 (added outside the chunks).  Returns (wpack, whalf) float32.

    layout = 1 (mpx_analysis_compressed_fused_layout, n_waves = 8): the magnitude product runs on v_mfma_f32_4x4x1_16b_f32 --
    per (chunk q, wave) eight magnitude fragments, then the phase tiles' fragments as above:
      wpack[q][wave][2 kg + half][lane][e] = W_mag[32 half + (lane & 31)][bin(q, 16 wave + 4 kg + e)]      (kg < 4, half < 2)
      wpack[q][wave][8 + jt][lane][e]      = W_ph[16 jt + (lane & 15)][bin(q, 16 wave + 4 (lane >> 4) + e)]
    whalf is the same in both layouts.
    """
    w_mag, w_ph = np.asarray(w_mag, dtype=np.float64), np.asarray(w_ph, dtype=np.float64)
    H = fft_len // 2 + 1
    assert w_mag.shape[1] == H and w_ph.shape[1] == H and w_mag.shape[0] <= 64 and w_ph.shape[0] <= 48
    assert n_waves in (4, 8)
    ntm, ntp = 4, (w_ph.shape[0] + 15) // 16
    tiles = []
    for src, nt in ((w_mag, ntm), (w_ph, ntp)):
        for jt in range(nt):
            t = np.zeros((16, H))
            rows = src[16 * jt:16 * jt + 16]
            t[:rows.shape[0]] = rows
            tiles.append(t)
    tiles = np.stack(tiles)                                # [T, 16, H]
    bins = fused_chunk_bins(fft_len)                       # [P/2, 128]
    lane = np.arange(64)
    li, g = lane & 15, lane >> 4
    cols, kh = 128 // n_waves, 128 // n_waves // 16
    col = (cols * np.arange(n_waves)[:, None, None, None] + 16 * np.arange(kh)[None, :, None, None]
           + 4 * g[None, None, :, None] + np.arange(4)[None, None, None, :])                    # [wave, h, lane, e]
    b = bins[:, col]                                       # [q, wave, h, lane, e]
    wpack = tiles[:, li[None, None, None, :, None], b]     # [T, q, wave, h, lane, e]
    wpack = np.ascontiguousarray(np.transpose(wpack, (1, 2, 3, 0, 4, 5)), dtype=np.float32)   # [q, wave, h, T, lane, e]
    whalf = np.ascontiguousarray(tiles[:, :, fft_len // 4], dtype=np.float32)   # [T, 16]
    if layout == 1:
        assert n_waves == 8
        wm64 = np.zeros((64, H))
        wm64[:w_mag.shape[0]] = w_mag
        colm = (16 * np.arange(8)[:, None, None] + 4 * np.arange(4)[None, :, None] + np.arange(4)[None, None, :])   # [wave, kg, e]
        bm = bins[:, colm]                                                                     # [q, wave, kg, e]
        rows = 32 * np.arange(2)[:, None] + (lane & 31)[None, :]                               # [half, lane]
        magf = wm64[rows[None, None, None, :, :, None], bm[:, :, :, None, None, :]]            # [q, wave, kg, half, lane, e]
        magf = magf.reshape(bm.shape[0], 8, 8, 64, 4)
        phf = wpack[:, :, 0, ntm:]                                                             # [q, wave, ntp, lane, e]
        out = np.concatenate((magf, phf), axis=2)
        return np.ascontiguousarray(out, dtype=np.float32).reshape(-1), whalf.reshape(-1)
    return wpack.reshape(-1), whalf.reshape(-1)


def var_to_const_rate_table(v_pm_smpls, const_rate_ms, fs):
    """
    Row/weight table of magphase.py:2219-2239 (Q15): grid arange(step, pm[-1], step); the first row is duplicated at
    t = 0 when pm[0] > 0.  Returns (row_lo, row_hi int64[Fc], t float64[Fc]) with out = (1-t)*rows[lo] + t*rows[hi].
    """
    v_pm_smpls = np.asarray(v_pm_smpls)
    step = fs * const_rate_ms / 1000
    centres = np.arange(step, v_pm_smpls[-1], step)
    if v_pm_smpls[0] > 0:
        x = np.r_[0, v_pm_smpls]
        node_row = np.r_[0, np.arange(v_pm_smpls.size)]
    else:
        x = v_pm_smpls
        node_row = np.arange(v_pm_smpls.size)
    idx = np.clip(np.searchsorted(x, centres), 1, x.size - 1)   # scipy interp1d's bracketing
    lo, hi = idx - 1, idx
    t = (centres - x[lo]) / (x[hi] - x[lo]).astype(np.float64)
    return node_row[lo].astype(np.int64), node_row[hi].astype(np.int64), t


def post_filter_tables(mag_dim, fs, av_len_at_zero=None, av_len_at_nyq=None, boost_at_zero=None, boost_at_nyq=None):
    """
    Host tables of the MagPhase post-filter (magphase.py:2300-2347, Q20): defaults per sample rate (same warnings /
    ValueError as the reference), bins with a moving average v_nx, their half lengths, and the tilt factors.
    Returns (nx_first, nx_last, half_len int64[], tilt float64[mag_dim]).
    """
    if mag_dim != 60:
        warnings.warn('Post-filter: It has been only tested with 60 dimensional mag data. '
                      'If you use another dimension, the result may be suboptimal.')
    opts = [av_len_at_zero, av_len_at_nyq, boost_at_zero, boost_at_nyq]
    if fs == 48000:
        defaults = [round_to_int(11.0 * (mag_dim / 60.0)), round_to_int(3.0 * (mag_dim / 60.0)), 1.8, 2.0]
    elif fs == 16000:
        if any(o is None for o in opts):
            warnings.warn('Post-filter: The default parameters for 16kHz sample rate have not being tunned.')
        defaults = [round_to_int(9.0 * (mag_dim / 60.0)), round_to_int(12.0 * (mag_dim / 60.0)), 2.0, 1.6]
    else:
        if any(o is None for o in opts):
            raise ValueError('Post-filter: It has only been tested with 16kHz and 48kHz sample rates.'
                             '\nProvide your own values for the options: av_len_at_zero, av_len_at_nyq, '
                             'boost_at_zero,\nboost_at_nyq if you use another sample rate')
        defaults = opts
    av0, avn, b0, bn = [d if o is None else o for o, d in zip(opts, defaults)]
    v_nx = np.arange(np.floor(av0 / 2), mag_dim - np.floor(avn / 2)).astype(int)
    v_lens = (2 * np.ceil(np.linspace(av0, avn, v_nx.size) / 2) - 1).astype(int)
    return int(v_nx[0]), int(v_nx[-1]), (v_lens // 2).astype(np.int64), np.linspace(b0, bn, mag_dim)


def hpf_tables(fs, block):
    """
    Output high-pass of synthesis_from_compressed (magphase.py:981-995): Butterworth order 4 at 40 Hz as two
    second-order sections (the same design, scipy output='sos') and, per section, the tables of the blocked scan of
    mpx_output_hpf: the direct-form-II-transposed state update z' = A z + Bx x with A = [[-a1, 1], [-a2, 0]],
    y = z0 + b0 x.  Returns (sos [2 x 6], A^block [2 x 4], G [2 x block x 2] with G[n] = [1, 0] A^n), float64.
    """
    from scipy import signal

    sos = np.ascontiguousarray(signal.butter(4, 40 / (fs / 2.0), btype='highpass', output='sos'), dtype=np.float64)
    pm = np.zeros((2, 4))
    g = np.zeros((2, block, 2))
    for sec in range(2):
        a1, a2 = sos[sec, 4] / sos[sec, 3], sos[sec, 5] / sos[sec, 3]
        amat = np.array([[-a1, 1.0], [-a2, 0.0]])
        row = np.array([1.0, 0.0])
        for n in range(block):
            g[sec, n] = row
            row = row @ amat
        pm[sec] = np.linalg.matrix_power(amat, block).reshape(-1)
    return sos, pm, g


# ======================================================================================================
# Merlin-style mel-cepstral post-filter (magphase.py:3375-3465)
# ======================================================================================================
def _f32(x):
    """One SPTK pipe boundary: every tool reads and writes float32, computes in double."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def rceps_compact(m_log):
    """
    la.rceps(in_type='log', out_type='compact') (libaudio.py:252-270): real cepstrum of the even extension of
    [F x n] log spectra (length 2(n-1)), first n coefficients, coefficients 1..n-3 doubled (the reference's slice
    1:(ncoeffs-2) leaves n-2 undoubled, Q6).
    """
    m_log = np.asarray(m_log, dtype=np.float64)
    n = m_log.shape[1]
    m_ext = np.hstack((m_log, m_log[:, -2:0:-1]))
    m_c = np.fft.ifft(m_ext).real
    m_c[:, 1:(n - 2)] *= 2
    return m_c[:, :n]


def sptk_mc2b(m_mc, alpha):
    """SPTK ``mc2b``: MLSA filter coefficients b[m] = mc[m] - alpha b[m+1], from the top coefficient down."""
    m_b = np.array(m_mc, dtype=np.float64)
    for m in range(m_b.shape[1] - 2, -1, -1):
        m_b[:, m] -= alpha * m_b[:, m + 1]
    return m_b


def sptk_b2mc(m_b, alpha):
    """SPTK ``b2mc`` (inverse of mc2b): mc[m] = b[m] + alpha b[m+1]."""
    m_mc = np.array(m_b, dtype=np.float64)
    m_mc[:, :-1] += alpha * np.asarray(m_b, dtype=np.float64)[:, 1:]
    return m_mc


_FREQT_CACHE = {}


def sptk_freqt(m_c, order_out, alpha_in, alpha_out=0.0):
    """SPTK ``freqt -m m1 -a alpha_in -M order_out -A alpha_out`` on [F x (m1+1)] cepstra, as one matrix product."""
    a = (alpha_out - alpha_in) / (1.0 - alpha_in * alpha_out)
    key = (m_c.shape[1], int(order_out), float(a))
    if key not in _FREQT_CACHE:
        _FREQT_CACHE[key] = freqt_matrix(m_c.shape[1], order_out, a).T.copy()   # [n_in x (order_out+1)]
    return np.asarray(m_c, dtype=np.float64) @ _FREQT_CACHE[key]


def sptk_c2acr_r0(m_c, fft_len):
    """SPTK ``c2acr -M 0 -l fft_len``: r[0] = mean over the fft_len bins of exp(2 Re FFT(c)) -- the frame energy."""
    m_x = np.zeros((m_c.shape[0], fft_len))
    m_x[:, :m_c.shape[1]] = m_c
    m_re = np.fft.fft(m_x, axis=1).real
    return np.exp(2.0 * m_re).sum(axis=1) / fft_len


def cos_matrix_log_spectrum(m_mcep, n_spbins):
    """la.mcep_to_sp_cosmat(alpha=0.0, out_type='log') (libaudio.py:605-631): out[k] = sum_n mc[n] cos(n pi k/(n_spbins-1))."""
    v_w = np.linspace(0, np.pi, num=n_spbins)
    m_trans = np.cos(np.outer(np.arange(m_mcep.shape[1]), v_w))
    return np.asarray(m_mcep, dtype=np.float64) @ m_trans


_MERLIN_CACHE = {}


def merlin_tables(dim, fs, pf_coef=1.4, fft_len=4096):
    """
    Constant tables of the device form of the Merlin post-filter (mpx_post_filter_merlin), float64:
      c1 [dim x dim]   rceps_compact as a matrix (mcep = x @ c1)
      lifter [dim]     (1, 1, pf, pf, ...) with pf printed with two decimals like the reference's command line
      g [dim x H]      freqt(alpha -> 0, order fft_len/2 - 1) followed by the real part of the fft_len-point DFT, on the
                       H = fft_len/2 + 1 distinct bins;  wk [H] = (1, 2, ..., 2, 1) / fft_len  (c2acr -M 0 -l fft_len)
      cf [dim x dim]   cosine matrix of cos_matrix_log_spectrum (alpha = 0);  alpha = define_alpha(fs)
    """
    key = (int(dim), int(fs), float("%1.2f" % pf_coef), int(fft_len))
    if key not in _MERLIN_CACHE:
        dim, alpha, half = key[0], define_alpha(fs), fft_len // 2
        c1 = rceps_compact(np.eye(dim))
        lifter = np.concatenate(([1.0, 1.0], np.full(dim - 2, key[2])))
        a = (0.0 - alpha) / (1.0 - alpha * 0.0)
        fq = freqt_matrix(dim, half - 1, a).T                     # [dim x half] cepstrum on the linear axis
        dcos = np.cos(2.0 * np.pi * np.outer(np.arange(half), np.arange(half + 1)) / fft_len)   # [half x H]
        wk = np.full(half + 1, 2.0 / fft_len)
        wk[0] = wk[-1] = 1.0 / fft_len
        cf = np.cos(np.outer(np.arange(dim), np.linspace(0, np.pi, num=dim)))
        _MERLIN_CACHE[key] = dict(c1=c1, lifter=lifter, g=fq @ dcos, wk=wk, cf=cf, alpha=alpha)
    return _MERLIN_CACHE[key]
