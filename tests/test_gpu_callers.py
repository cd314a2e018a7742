"""-m gpu: the python-3 counterparts of the reference's demos / batch scripts run end to end on the device path."""
import os
import subprocess
import sys
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, cwd):
    out = subprocess.run([sys.executable] + args, cwd=cwd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    return out.stdout


def _wav_len(path):
    with wave.open(path, "rb") as w:
        return w.getnframes(), w.getframerate()


def test_demos_and_batch_scripts(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "demos"))
    import make_demo_data
    wav_dir = tmp_path / "wavs_nat"
    toks = make_demo_data.main(n=2, out_dir=str(wav_dir))
    scp = tmp_path / "file_id.scp"
    assert scp.read_text().split() == toks
    syn_dir = tmp_path / "syn"
    # lossless copy synthesis
    _run([os.path.join(ROOT, "demos", "demo_copy_synthesis_lossless.py"), "--wav", str(wav_dir / "syn_000.wav"),
          "--out-dir", str(syn_dir)], ROOT)
    n_in, fs = _wav_len(str(wav_dir / "syn_000.wav"))
    n_out, fs2 = _wav_len(str(syn_dir / "syn_000_copy_syn_lossless.wav"))
    assert fs == fs2 == 48000 and abs(n_out - n_in) < 2000
    # low-dim copy synthesis
    _run([os.path.join(ROOT, "demos", "demo_copy_synthesis_low_dim.py"), "--wav", str(wav_dir / "syn_000.wav"),
          "--out-dir", str(syn_dir)], ROOT)
    assert os.path.isfile(str(syn_dir / "syn_000_copy_syn_low_dim_mag_dim_60_ph_dim_45_const_rate_0.wav"))
    # batch feature extraction (Q7: phase_dim 10, alpha_phase False -> 44 linear-cepstral bins cut to 10)
    feats = tmp_path / "params_nat"
    _run([os.path.join(ROOT, "scripts", "batch_feature_extraction_for_tts.py"), "--scp", str(scp), "--wav-dir",
          str(wav_dir), "--out-dir", str(feats)], ROOT)
    for tok in toks:
        mag = np.fromfile(str(feats / (tok + ".mag")), dtype=np.float32)
        real = np.fromfile(str(feats / (tok + ".real")), dtype=np.float32)
        lf0 = np.fromfile(str(feats / (tok + ".lf0")), dtype=np.float32)
        shift = np.fromfile(str(feats / (tok + ".shift")), dtype=np.float32)
        assert mag.size == 60 * lf0.size and real.size == 10 * lf0.size and shift.size == lf0.size
    # batch generation from the bundled predicted features
    gen = tmp_path / "gen"
    _run([os.path.join(ROOT, "scripts", "batch_waveform_generation.py"), "--out-dir", str(gen)], ROOT)
    for tok in ("hvd_704", "hvd_705", "hvd_706", "hvd_708"):
        n, fsw = _wav_len(str(gen / (tok + ".wav")))
        assert fsw == 48000 and n > 40000
