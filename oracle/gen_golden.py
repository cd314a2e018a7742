"""
TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REAL reference
(/root/reference, imported in memory through oracle/ref_shim.py).  Run in the build container:

    python -m oracle.gen_golden

The fixtures are data only (inputs + the reference's outputs).  Vectors whose computation passes
through SPTK ``mcep`` (absent external binary, restated in oracle/magphase_oracle.py:sptk_mcep)
carry ``pinned = 0`` ("oracle-with-our-mcep"); everything else is the reference's own arithmetic
(``pinned = 1``).

Inputs copied as data: demos/data_48k/params_predicted/hvd_704.{mag,real,imag,lf0}
(float32 feature files the reference bundles as inputs for its generation demo).
"""
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from magphase_amd import synthetic as syn  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PROJ_SEED = 4242


def proj(m, seed=PROJ_SEED):
    """float64 random projections along both axes: a linear checksum sensitive to every element."""
    m = np.asarray(m, dtype=np.float64)
    r = np.random.RandomState(seed)
    return m @ r.standard_normal(m.shape[1]), r.standard_normal(m.shape[0]) @ m


def gen_index_cases(mp):
    """G1: epoch rounding / frame bounds (exact integers) straight from mp.windowing."""
    out = {}
    rng = np.random.RandomState(11)
    cases = []
    for fs, n in ((48000, 30000), (16000, 12000)):
        pm = np.cumsum(rng.uniform(0.002, 0.011, 60)) * fs
        cases.append((pm[pm < n - 2], n))
    cases.append((np.array([0.0, 100.5, 101.5, 102.5, 300.49999, 5000.0]), 9001))          # L=0, half-even ties
    cases.append((np.array([2.0, 2.4, 2.6, 700.0, 5600.0]), 5700))                           # equal after rounding; frame > 4096
    cases.append((np.array([4500.0, 4600.0]), 4700))                                         # first left_len > fft_len
    for i, (pm, n) in enumerate(cases):
        sig = np.zeros(n)
        _, v_lens, v_pm_plus, v_shift, v_rights = mp.windowing(sig, pm)
        out["c%d_pm" % i] = pm
        out["c%d_n" % i] = np.int64(n)
        out["c%d_lens" % i] = v_lens.astype(np.int64)
        out["c%d_pm_plus" % i] = v_pm_plus.astype(np.int64)
        out["c%d_shift" % i] = v_shift.astype(np.int64)
        out["c%d_rights" % i] = v_rights.astype(np.int64)
    out["ncases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "g1_index.npz"), pinned=np.int64(1), **out)


def gen_lossless(mp, tag, u, fs, dur):
    """G2/G3: lossless analysis and synthesis of a synthetic utterance."""
    pcm, pm_sec, voi = syn.make_utterance(u, dur_s=dur, fs=fs)
    wav = os.path.abspath("g2_%s.wav" % tag)
    ref_shim._wav_write(wav, pcm / 32768.0, fs)
    ref_shim.set_epochs(wav, pm_sec, voi)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        m_mag, m_real, m_imag, v_f0, fs_out, v_shift = mp.analysis_lossless(wav)
        n_trunc_warn = sum("fft_len" in str(w.message) for w in wlist)
    v_syn = mp.synthesis_from_lossless(m_mag, m_real, m_imag, v_f0, fs_out)
    sel = np.unique(np.r_[0, 1, len(v_shift) // 3, len(v_shift) // 2, len(v_shift) - 2, len(v_shift) - 1])
    d = dict(pinned=np.int64(1), fs=np.int64(fs), pcm=pcm, pm_sec=pm_sec, voi=voi,
             v_shift=v_shift.astype(np.int64), v_f0=v_f0, n_trunc_warn=np.int64(n_trunc_warn),
             mag32=m_mag.astype(np.float32), real32=m_real.astype(np.float32), imag32=m_imag.astype(np.float32),
             sel=sel, mag_sel=m_mag[sel], real_sel=m_real[sel], imag_sel=m_imag[sel], v_syn=v_syn)
    for nm, m in (("mag", m_mag), ("real", m_real), ("imag", m_imag)):
        d[nm + "_projc"], d[nm + "_projr"] = proj(m)
    np.savez_compressed(os.path.join(OUT, "g2_lossless_%s.npz" % tag), **d)
    os.remove(wav)


def gen_unwarp(mp, la):
    """G4: the mel-unwarp linear maps (pure reference arithmetic: libaudio.py:667-684, magphase.py:1219-1235)."""
    d = dict(pinned=np.int64(1))
    for tag, n, nb, alpha in (("mag48", 60, 2049, 0.77), ("mag16", 60, 1025, 0.58), ("q7", 44, 2049, 0.0)):
        U = la.sp_mel_unwarp(np.eye(n), nb, alpha=alpha, in_type="log")
        d[tag + "_cols"] = U[:, ::16].copy()
        d[tag + "_projc"], d[tag + "_projr"] = proj(U)
    for tag, pd, fft_len, fs, alpha in (("ph48", 45, 4096, 48000, 0.77), ("ph16", 45, 2048, 16000, 0.58), ("ph48_10", 10, 4096, 48000, 0.77)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            R, I = mp.phase_uncompress_type1_mcep(np.eye(pd), np.eye(pd)[::-1].copy(), alpha, fft_len, fs)
        d[tag + "_R_cols"] = R[:, ::16].copy()
        d[tag + "_R_projc"], d[tag + "_R_projr"] = proj(R)
        d[tag + "_I_projc"], d[tag + "_I_projr"] = proj(I)
    np.savez_compressed(os.path.join(OUT, "g4_unwarp.npz"), **d)


def gen_compressed_synthesis(mp, lu):
    """G5/G6: generation from the bundled predicted features hvd_704 (post-filter, seeded noise)."""
    pdir = os.path.join(ref_shim.REF_ROOT, "demos", "data_48k", "params_predicted")
    raw = {}
    for ext, dim in (("mag", 60), ("real", 45), ("imag", 45), ("lf0", 1)):
        raw[ext] = np.fromfile(os.path.join(pdir, "hvd_704." + ext), dtype=np.float32)
    m_mag = lu.read_binfile(os.path.join(pdir, "hvd_704.mag"), dim=60)
    m_real = lu.read_binfile(os.path.join(pdir, "hvd_704.real"), dim=45)
    m_imag = lu.read_binfile(os.path.join(pdir, "hvd_704.imag"), dim=45)
    v_lf0 = lu.read_binfile(os.path.join(pdir, "hvd_704.lf0"), dim=1)
    pf48 = mp.post_filter(m_mag, 48000)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pf16 = mp.post_filter(m_mag, 16000)
    d = dict(pinned=np.int64(1), in_mag=raw["mag"], in_real=raw["real"], in_imag=raw["imag"], in_lf0=raw["lf0"],
             pf48=pf48, pf16=pf16, seed=np.int64(20260928))
    for hpf in (True, False):
        np.random.seed(int(d["seed"]))
        v = mp.synthesis_from_compressed(pf48, m_real, m_imag, v_lf0, 48000, b_out_hpf=hpf)
        d["syn_pf_hpf%d" % int(hpf)] = v
    np.random.seed(int(d["seed"]))
    d["syn_nopf_minphase"] = mp.synthesis_from_compressed(m_mag, m_real, m_imag, v_lf0, 48000, per_phase_type="min_phase")
    np.random.seed(int(d["seed"]))
    d["syn_nopf_novoiwin"] = mp.synthesis_from_compressed(m_mag, m_real, m_imag, v_lf0, 48000, b_voi_ap_win=False)
    # index intermediates (pure fp64 host math of magphase.py:846-848,879-882)
    v_f0 = np.exp(v_lf0)
    v_shift = mp.f0_to_shift(v_f0, 48000).astype(int)
    d["v_shift"] = v_shift.astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "g5_generation_hvd704.npz"), **d)


def gen_const_rate(mp, la):
    """
    G7/G8: constant-rate tables and compressed analysis.  G7 (interpolation + backward scan) is
    pure reference arithmetic; G8 passes through our mcep restatement -> pinned = 0.
    """
    fs = 48000
    pcm, pm_sec, voi = syn.make_utterance(5, dur_s=0.8, fs=fs)
    wav = os.path.abspath("g7.wav")
    ref_shim._wav_write(wav, pcm / 32768.0, fs)
    ref_shim.set_epochs(wav, pm_sec, voi)
    m_mag, m_real, m_imag, v_f0, _, v_shift = mp.analysis_lossless(wav)
    v_pm = la.shift_to_pm(v_shift)
    m_mag_c = mp.interp_from_variable_to_const_frm_rate(m_mag, v_pm, 5.0, fs)
    v_voi = v_f0 > 1.0
    v_f0_c = mp.interp_from_variable_to_const_frm_rate(
        np.r_[v_f0[v_voi][0], v_f0[v_voi], v_f0[v_voi][-1]], np.r_[0, v_pm[v_voi], v_pm[-1]], 5.0, fs).squeeze()
    v_voi_c = mp.interp_from_variable_to_const_frm_rate(v_voi, v_pm, 5.0, fs) > 0.5
    v_f0_c = v_f0_c * v_voi_c
    shift_c = mp.f0_to_shift(v_f0_c, fs)
    v_shift_vr, v_locs = mp.get_shifts_and_frm_locs_from_const_shifts(shift_c, 5.0, fs, interp_type="linear")
    m_back = mp.interp_from_const_to_variable_rate(m_mag_c, v_locs, 5.0, fs)
    g7 = dict(pinned=np.int64(1), fs=np.int64(fs), pcm=pcm, pm_sec=pm_sec, voi=voi, v_shift=v_shift.astype(np.int64),
              v_f0_c=v_f0_c, v_shift_vr=v_shift_vr, v_locs=v_locs, mag_c_cols=m_mag_c[:, ::32].copy(),
              back_cols=m_back[:, ::32].copy())
    g7["mag_c_projc"], g7["mag_c_projr"] = proj(m_mag_c)
    g7["back_projc"], g7["back_projr"] = proj(m_back)
    np.savez_compressed(os.path.join(OUT, "g7_const_rate.npz"), **g7)

    g8 = dict(pinned=np.int64(0), note="oracle-with-our-mcep", fs=np.int64(fs), pcm=pcm, pm_sec=pm_sec, voi=voi)
    for tag, kw in (("vr45", dict(phase_dim=45)), ("cr45", dict(phase_dim=45, b_const_rate=True)),
                    ("q7", dict(phase_dim=10, alpha_phase=False))):
        r = mp.analysis_compressed(wav, mag_dim=60, **kw)
        g8[tag + "_mag"], g8[tag + "_real"], g8[tag + "_imag"], g8[tag + "_lf0"] = r[0], r[1], r[2], r[3]
        g8[tag + "_shift"] = r[4].astype(np.int64)
        if tag == "cr45":
            np.random.seed(99)
            g8["cr45_syn"] = mp.synthesis_from_compressed(r[0], r[1], r[2], r[3], fs, b_const_rate=True, b_out_hpf=False)
            g8["cr45_seed"] = np.int64(99)
    np.savez_compressed(os.path.join(OUT, "g8_compressed_analysis.npz"), **g8)
    os.remove(wav)


def _py3_loadtxt(orig):
    """numpy-2 / python-3 stand-in for the reference's np.loadtxt(dtype='string') (python-2 numpy's name of str)."""
    def f(*a, **k):
        if k.get("dtype") == "string":
            k["dtype"] = str
        return orig(*a, **k)
    return f


def gen_fbank(mp, la, lu):
    """G10: filter-bank mel unwarp (la.sp_mel_unwarp_fbank) and synthesis_from_compressed(b_fbank_mel=True)."""
    rng = np.random.RandomState(123)
    out = {}
    for nb, nbins, alpha in ((60, 2049, 0.77), (60, 1025, 0.58), (40, 2049, 0.77)):
        x = rng.randn(5, nb) * 0.5 - 3.0
        y = la.sp_mel_unwarp_fbank(x, nbins, alpha=alpha)
        out["x_%d_%d" % (nb, nbins)] = x
        out["y_%d_%d" % (nb, nbins)] = y
    d = os.path.join(ref_shim.REF_ROOT if hasattr(ref_shim, "REF_ROOT") else "/root/reference", "demos", "data_48k",
                     "params_predicted")
    m_mag = lu.read_binfile(os.path.join(d, "hvd_704.mag"), dim=60)
    m_real = lu.read_binfile(os.path.join(d, "hvd_704.real"), dim=45)
    m_imag = lu.read_binfile(os.path.join(d, "hvd_704.imag"), dim=45)
    v_lf0 = lu.read_binfile(os.path.join(d, "hvd_704.lf0"), dim=1)
    np.random.seed(77)
    out["seed"] = 77
    out["syn_fbank"] = mp.synthesis_from_compressed(m_mag, m_real, m_imag, v_lf0, 48000, b_fbank_mel=True)
    np.savez_compressed(os.path.join(OUT, "g10_fbank.npz"), **out)


def gen_fbank_warp(mp, la):
    """G11: analysis-side filter bank (la.sp_mel_warp_fbank, libaudio.py:721-769 -- pure numpy, so this vector is
    PINNED) and format_for_modelling(b_mag_fbank_mel=True)'s magnitude stream on the G8 lossless features."""
    rng = np.random.RandomState(321)
    out = {}
    for nb, nbins, alpha in ((60, 2049, 0.77), (60, 1025, 0.58), (40, 2049, 0.77)):
        x = np.exp(rng.randn(6, nbins) * 1.5 - 2.0)
        x[4, 100:140] = 0.0                      # exact zeros: la.log's MAGIC floor, exp underflow, MAGIC again
        x[5, :] = 0.0                            # a silent frame
        y = la.log(la.sp_mel_warp_fbank(x, nb, alpha=alpha))
        out["x_%d_%d" % (nb, nbins)] = x
        out["y_%d_%d" % (nb, nbins)] = y
    g = np.load(os.path.join(OUT, "g2_lossless_48k.npz"))
    m_mag, m_real, m_imag = (g[k].astype(np.float64) for k in ("mag32", "real32", "imag32"))   # inputs = the fixture's
    r = mp.format_for_modelling(m_mag, m_real, m_imag, g["v_f0"], 48000, mag_dim=60, phase_dim=45, b_mag_fbank_mel=True)
    out["ffm_mag_mel_log"] = r[0]
    out["ffm_lf0"] = r[3]
    np.savez_compressed(os.path.join(OUT, "g11_fbank_warp.npz"), **out)


def gen_labels(mp, la):
    """G9: HTS state-aligned labels -> frames per state -> variable-frame-rate labels (magphase.py:2111-2150,
    libaudio.py:687-708).  The label text is synthetic (5 states per phone, 5 ms grid)."""
    rng = np.random.RandomState(77)
    fs = 48000
    v_shift = np.concatenate([rng.randint(180, 520, size=150), np.full(60, 240), rng.randint(200, 400, size=90)]).astype(int)
    ep_ms = np.cumsum(v_shift) * 1000.0 / fs
    n_ph, n_st = 14, 5
    total_ms = float(np.ceil(ep_ms[-1] / 5.0) * 5.0 + 5.0)

    def make_lab(end_ms, zero_state=None):
        cuts = np.sort(rng.choice(np.arange(1, int(end_ms / 5.0)), size=n_ph * n_st - 1, replace=False)) * 5.0
        edges = np.concatenate(([0.0], cuts, [end_ms]))
        if zero_state is not None:   # a state shorter than any epoch interval: gets no frame
            k = zero_state
            edges[k + 1] = edges[k] + 0.0001 * 0 + 5.0
        lines = []
        for i in range(n_ph * n_st):
            ph, st = divmod(i, n_st)
            lines.append("%d %d x^p%d-p%d+p%d=y@%d_%d[%d]" % (int(round(edges[i] * 10000)), int(round(edges[i + 1] * 10000)),
                                                           ph, ph + 1, ph + 2, ph % 3, ph % 4, st + 2))
        return "\n".join(lines) + "\n"

    out = {"fs": fs, "v_shift": v_shift}
    cases = {"exact": make_lab(total_ms), "short_end": make_lab(float(np.floor(ep_ms[-4] / 5.0) * 5.0)),
             "too_short": make_lab(float(np.floor(ep_ms[-40] / 5.0) * 5.0))}
    la.np.loadtxt = _py3_loadtxt(np.loadtxt) if not hasattr(la.np.loadtxt, "__wrapped_py3__") else la.np.loadtxt
    for name, text in cases.items():
        with open(name + ".lab", "w") as f:
            f.write(text)
        out["lab_" + name] = np.array(text)
        for pz in (False, True):
            key = "%s_pz%d" % (name, int(pz))
            try:
                v_n = mp.get_num_of_frms_per_state(v_shift, name + ".lab", fs, b_prevent_zeros=pz)
                out["nfrms_" + key] = np.asarray(v_n, dtype=np.float64)
                la.convert_label_state_align_to_var_frame_rate(name + ".lab", v_n, name + "_out.lab")
                out["outlab_" + key] = np.array(open(name + "_out.lab").read())
            except ValueError as e:
                out["error_" + key] = np.array(str(e))
    np.savez_compressed(os.path.join(OUT, "g9_labels.npz"), **out)


def gen_merlin_legs(la):
    """G12: the two legs of post_filter_merlin (magphase.py:3398, :3451) that are the reference's own Python --
    la.rceps(log, compact) and la.mcep_to_sp_cosmat(alpha=0, 'log') -- on seeded log mel magnitudes (60 and 24 bins) and
    on the bundled predicted magnitudes.  The SPTK legs in between stay unpinned."""
    rng = np.random.RandomState(1204)
    out = {"pinned": np.int64(1)}
    pred = np.fromfile(os.path.join(ref_shim.REF_ROOT, "demos", "data_48k", "params_predicted", "hvd_704.mag"),
                       dtype=np.float32).reshape(-1, 60).astype(np.float64)
    for tag, x in (("a60", rng.uniform(-9, 1, (12, 60))), ("b24", rng.uniform(-9, 1, (7, 24))), ("pred", pred[:40])):
        out[tag + "_in"] = x
        c = la.rceps(x.copy(), in_type="log", out_type="compact")
        out[tag + "_rceps"] = c
        out[tag + "_cos"] = la.mcep_to_sp_cosmat(c.astype(np.float32).astype(np.float64), x.shape[1], alpha=0.0, out_type="log")
    np.savez_compressed(os.path.join(OUT, "g12_merlin_legs.npz"), **out)


def main():
    if not ref_shim.reference_available():
        raise SystemExit("reference not present; golden vectors can only be generated in the build container")
    os.makedirs(OUT, exist_ok=True)
    mp, la, lu = ref_shim.load_reference()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)  # the reference drops temp_<host>_<pid>.est files in the CWD
        try:
            gen_index_cases(mp)
            gen_lossless(mp, "48k", 3, 48000, 0.35)
            gen_lossless(mp, "16k", 4, 16000, 0.45)
            gen_unwarp(mp, la)
            gen_compressed_synthesis(mp, lu)
            gen_const_rate(mp, la)
            gen_labels(mp, la)
            gen_fbank(mp, la, lu)
            gen_fbank_warp(mp, la)
            gen_merlin_legs(la)
        finally:
            os.chdir(cwd)
    for f in sorted(os.listdir(OUT)):
        print("%-36s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024.0))


if __name__ == "__main__":
    main()
