#!/usr/bin/env python
"""
Lossless copy synthesis on the MI355X path: analysis_lossless -> synthesis_from_lossless -> wav.

Counterpart (python 3) of the reference demo of the same name: same three library calls in the same order and the same
output file name (<token>_copy_syn_lossless.wav).  Default input: the reference's own demo recording
demos/data_48k/wavs_nat/hvd_593.wav (bundled as data).  The reference gets its epochs from the REAPER binary; here they
come from <stem>.est next to the wav if there is one, else -- stated on the command line, never silently -- from the
built-in tracker (--epochs builtin, the default of this demo; not REAPER: parity unpinned) or from a REAPER binary
(--epochs reaper).

    python demos/demo_copy_synthesis_lossless.py [--wav FILE] [--out-dir DIR] [--epochs builtin|reaper]
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
    ap.add_argument("--wav", default=os.path.join(HERE, "data_48k", "wavs_nat", "hvd_593.wav"))
    ap.add_argument("--out-dir", default=os.path.join(HERE, "data_48k", "wavs_syn"))
    ap.add_argument("--epochs", default="builtin", choices=["builtin", "reaper"],
                    help="epoch source when there is no <stem>.est next to the wav")
    ap.add_argument("--one-launch", action="store_true",
                    help="analysis + synthesis as one device launch (magphase.copy_synthesis_lossless)")
    args = ap.parse_args()
    if not os.path.isfile(args.wav):
        sys.path.insert(0, HERE)
        import make_demo_data
        make_demo_data.main(out_dir=os.path.dirname(args.wav))
    lu.mkdir(args.out_dir)
    if args.epochs == "builtin" and not os.path.isfile(os.path.splitext(args.wav)[0] + ".est"):
        print("epochs: built-in zero-frequency-filtering tracker (not REAPER)")
        mp.use_builtin_epoch_tracker()

    if args.one_launch:   # the two calls below as one device launch (mpx_roundtrip_lossless_ola): same outputs
        features, v_syn = mp.copy_synthesis_lossless(args.wav)
        m_mag, m_real, m_imag, v_f0, fs = features[:5]
        print("analysed and resynthesised %d pitch-synchronous frames x %d bins at %d Hz" % (m_mag.shape[0], m_mag.shape[1], fs))
    else:
        features = mp.analysis_lossless(args.wav)                    # (m_mag, m_real, m_imag, v_f0, fs, v_shift)
        m_mag, m_real, m_imag, v_f0, fs = features[:5]
        print("analysed %d pitch-synchronous frames x %d bins at %d Hz" % (m_mag.shape[0], m_mag.shape[1], fs))
        v_syn = mp.synthesis_from_lossless(m_mag, m_real, m_imag, v_f0, fs)
    target = os.path.join(args.out_dir, lu.get_filename(args.wav) + "_copy_syn_lossless.wav")
    la.write_audio_file(target, v_syn, fs)
    print("wrote", target)


if __name__ == "__main__":
    main()
