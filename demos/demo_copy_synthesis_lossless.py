#!/usr/bin/env python
"""
Lossless copy synthesis on the MI355X path: analysis_lossless -> synthesis_from_lossless -> wav.

Counterpart (python 3) of the reference demo of the same name: same three library calls in the same order and the same
output file name (<token>_copy_syn_lossless.wav).  The input wav needs epochs next to it (<stem>.est, REAPER text
format); without arguments a synthetic utterance with exact epochs is generated first (demos/make_demo_data.py).

    python demos/demo_copy_synthesis_lossless.py [--wav FILE] [--out-dir DIR]
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.realpath(__file__))
sys.path.append(os.path.realpath(os.path.join(HERE, "..", "src")))

import libaudio as la  # noqa: E402
import libutils as lu  # noqa: E402
import magphase as mp  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--wav", default=os.path.join(HERE, "data_48k", "wavs_nat", "syn_000.wav"))
    ap.add_argument("--out-dir", default=os.path.join(HERE, "data_48k", "wavs_syn"))
    args = ap.parse_args()
    if not os.path.isfile(args.wav):
        sys.path.insert(0, HERE)
        import make_demo_data
        make_demo_data.main(out_dir=os.path.dirname(args.wav))
    lu.mkdir(args.out_dir)

    features = mp.analysis_lossless(args.wav)                    # (m_mag, m_real, m_imag, v_f0, fs, v_shift)
    m_mag, m_real, m_imag, v_f0, fs = features[:5]
    print("analysed %d pitch-synchronous frames x %d bins at %d Hz" % (m_mag.shape[0], m_mag.shape[1], fs))
    v_syn = mp.synthesis_from_lossless(m_mag, m_real, m_imag, v_f0, fs)
    target = os.path.join(args.out_dir, lu.get_filename(args.wav) + "_copy_syn_lossless.wav")
    la.write_audio_file(target, v_syn, fs)
    print("wrote", target)


if __name__ == "__main__":
    main()
