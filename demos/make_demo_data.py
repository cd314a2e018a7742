#!/usr/bin/env python
"""
Writes synthetic 48 kHz demo inputs: demos/data_48k/wavs_nat/syn_00N.wav + syn_00N.est (REAPER text format) and
file_id.scp.  The reference's bundled natural wavs need REAPER (absent here) for their epochs; these synthetic
utterances (magphase_amd/synthetic.py, SURVEY section 8d generator) come with exact epochs.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from magphase_amd import libaudio as la  # noqa: E402
from magphase_amd import synthetic as syn  # noqa: E402


def write_est(path, v_pm_sec, v_voi):
    with open(path, "w") as f:
        f.write("EST_File Track\nDataType ascii\nNumFrames %d\nNumChannels 1\n" % len(v_pm_sec))
        f.write("FrameShift 0.0\nVoicingEnabled true\nEST_Header_End\n")
        for t, v in zip(v_pm_sec, v_voi):
            f.write("%.6f %d 0.0\n" % (t, int(v)))


def main(n=3, out_dir=None, dur_s=None):
    out_dir = out_dir or os.path.join(HERE, "data_48k", "wavs_nat")
    os.makedirs(out_dir, exist_ok=True)
    toks = []
    for u in range(n):
        pcm, pm, voi = syn.make_utterance(500 + u, dur_s=(2.0 + 0.5 * u) if dur_s is None else dur_s + 0.05 * u, fs=48000)
        tok = "syn_%03d" % u
        la.write_audio_file(os.path.join(out_dir, tok + ".wav"), pcm / 32768.0, 48000, norm=None)
        write_est(os.path.join(out_dir, tok + ".est"), pm, voi)
        toks.append(tok)
    with open(os.path.join(os.path.dirname(out_dir), "file_id.scp"), "w") as f:
        f.write("\n".join(toks) + "\n")
    return toks


if __name__ == "__main__":
    print(main())
